// convert.hip -- K8/K9: the PixelType -> PixelType scanline conversion matrix on gfx950.
//
// Replaces the row kernels of source/gamut/scanline.d (:139-803) and the
// drivers scanlinesCopy (:37-55) / scanlinesConvert (:70-121).  The reference
// converts through an intermediate scanline (rgba8 when both ends are plain
// 8-bit, else rgbaf32, scanline.d:25-31); here the intermediate pixel lives in
// registers, so each pixel is read once and written once: the kernel is a pure
// HBM stream (algorithmic bytes = size(src)+size(dst) per pixel).
//
// Arithmetic contract (DESIGN.md "f32 parity"): every binary32 operation is
// rounded on its own -- built with -ffp-contract=off and written with
// __fadd_rn/__fmul_rn/__fdiv_rn so no FMA can form; division is the IEEE
// correctly rounded one (v_div_scale/v_div_fmas/v_div_fixup, or for the
// constant divisors 255 / 65535 an FMA sequence proven to give the same result
// for every input, div_by_max below), never a bare reciprocal multiply; float->int is x86 cvttss2si (truncate; NaN / out of
// int32 range -> 0x80000000) followed by taking the low 8/16 bits, which is
// what `cast(ubyte)(float)` compiles to in the reference's x86-64 build.
//
// Work decomposition: one thread converts a "unit" of G pixels, G chosen per
// type pair so the WIDER side of the unit is whole dwords and >= 16 B (one
// dwordx4 per lane, lane-contiguous => fully coalesced 1 KiB per wave
// instruction); the narrower side moves as exact dword / halfword / byte pieces
// (unit_pixels below).  The grid covers the units once, two per thread (see CONVERT_UNROLL).
#include "common.hpp"
#include <utility>

namespace gamut {
namespace {

typedef uint32_t u32;

template <int T> struct PT {
    static constexpr int ch     = kPixelChannels[T];
    static constexpr int bits   = (T % 3 == 0) ? 8 : (T % 3 == 1) ? 16 : 32;
    static constexpr int size   = kPixelSize[T];
    static constexpr bool premul = (T >= GAMUT_PIXEL_lap8 && T <= GAMUT_PIXEL_lapf32) || (T >= GAMUT_PIXEL_rgbap8);
    // internals/types.d:99-111: only l8, la8, rgb8, rgba8 are "8-bit" for the intermediate choice
    static constexpr bool plain8 = (T == GAMUT_PIXEL_l8 || T == GAMUT_PIXEL_la8 || T == GAMUT_PIXEL_rgb8 || T == GAMUT_PIXEL_rgba8);
};

constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }
constexpr int clcm(int a, int b) { return a / cgcd(a, b) * b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }
// Pixels per thread ("unit"): the WIDER side of the pair must be whole dwords and at least 16 bytes, so that it moves as
// lane-contiguous dwordx4 / dwordx2 / dword accesses.  The narrower side gets whatever that leaves -- 1, 2, 3, 6 ... bytes are
// fine: it is read / written with exact dword + halfword + byte pieces (gfx950 takes dwords at any byte address).  Forcing whole
// dwords on BOTH sides (the first version) made expanding pairs such as l16 -> rgbaf32 handle 2-4 pixels per thread, i.e. 32-64
// destination bytes per lane written as 16-byte pieces at a 32-64-byte lane stride: 47 % of peak instead of ~60 %.
constexpr int unit_pixels(int s, int d)
{
    const int wide = cmax(s, d);
    int g = 4 / cgcd(wide, 4);
    while (g * wide < 16) g *= 2;
    return g;
}
constexpr int vec_bytes(int unit_bytes) { return unit_bytes % 16 == 0 ? 16 : unit_bytes % 8 == 0 ? 8 : unit_bytes % 4 == 0 ? 4 : unit_bytes % 2 == 0 ? 2 : 1; }
struct __attribute__((packed)) AnyU32 { u32 v; };
struct __attribute__((packed)) AnyU16 { uint16_t v; };

// ---- unit load / store -----------------------------------------------------
#ifndef CONVERT_NT            // 1: nontemporal stores, 2: nontemporal loads too (tuning knob, tools/variant.sh)
#define CONVERT_NT 2
#endif
typedef u32 u32x4v __attribute__((ext_vector_type(4)));
typedef u32 u32x2v __attribute__((ext_vector_type(2)));
template <int BYTES>
__device__ __forceinline__ void load_unit(const uint8_t* p, u32 (&w)[(BYTES + 3) / 4])
{
    constexpr int V = vec_bytes(BYTES);
    if constexpr (V == 16) {
        #pragma unroll
        for (int i = 0; i < BYTES / 16; ++i) {
            u32x4v v;
            if (CONVERT_NT >= 2) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(p) + i); else v = reinterpret_cast<const u32x4v*>(p)[i];
            w[4*i] = v.x; w[4*i+1] = v.y; w[4*i+2] = v.z; w[4*i+3] = v.w;
        }
    } else if constexpr (V == 8) {
        #pragma unroll
        for (int i = 0; i < BYTES / 8; ++i) {
            u32x2v v;
            if (CONVERT_NT >= 2) v = __builtin_nontemporal_load(reinterpret_cast<const u32x2v*>(p) + i); else v = reinterpret_cast<const u32x2v*>(p)[i];
            w[2*i] = v.x; w[2*i+1] = v.y;
        }
    } else if constexpr (V == 4) {
        #pragma unroll
        for (int i = 0; i < BYTES / 4; ++i) w[i] = (CONVERT_NT >= 2) ? __builtin_nontemporal_load(reinterpret_cast<const u32*>(p) + i) : reinterpret_cast<const u32*>(p)[i];
    } else {                                                   // the narrow side of an expanding / shrinking pair: exact pieces
        #pragma unroll
        for (int i = 0; i < BYTES / 4; ++i) w[i] = reinterpret_cast<const AnyU32*>(p)[i].v;
        constexpr int R = BYTES % 4, O = BYTES - R;
        if constexpr (R == 1) w[BYTES / 4] = p[O];
        else if constexpr (R == 2) w[BYTES / 4] = reinterpret_cast<const AnyU16*>(p + O)->v;
        else if constexpr (R == 3) w[BYTES / 4] = (u32)reinterpret_cast<const AnyU16*>(p + O)->v | ((u32)p[O + 2] << 16);
    }
}
template <int BYTES>
__device__ __forceinline__ void store_unit(uint8_t* p, const u32 (&w)[(BYTES + 3) / 4])
{
    constexpr int V = vec_bytes(BYTES);
    if constexpr (V == 16) {
        #pragma unroll
        for (int i = 0; i < BYTES / 16; ++i) {
            const u32x4v v = { w[4*i], w[4*i+1], w[4*i+2], w[4*i+3] };
            if (CONVERT_NT >= 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4v*>(p) + i); else reinterpret_cast<u32x4v*>(p)[i] = v;
        }
    } else if constexpr (V == 8) {
        #pragma unroll
        for (int i = 0; i < BYTES / 8; ++i) {
            const u32x2v v = { w[2*i], w[2*i+1] };
            if (CONVERT_NT >= 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x2v*>(p) + i); else reinterpret_cast<u32x2v*>(p)[i] = v;
        }
    } else if constexpr (V == 4) {
        #pragma unroll
        for (int i = 0; i < BYTES / 4; ++i) { if (CONVERT_NT >= 1) __builtin_nontemporal_store(w[i], reinterpret_cast<u32*>(p) + i); else reinterpret_cast<u32*>(p)[i] = w[i]; }
    } else {
        #pragma unroll
        for (int i = 0; i < BYTES / 4; ++i) reinterpret_cast<AnyU32*>(p)[i].v = w[i];
        constexpr int R = BYTES % 4, O = BYTES - R;
        const u32 last = w[BYTES / 4];
        if constexpr (R == 1) p[O] = (uint8_t)last;
        else if constexpr (R == 2) reinterpret_cast<AnyU16*>(p + O)->v = (uint16_t)last;
        else if constexpr (R == 3) { reinterpret_cast<AnyU16*>(p + O)->v = (uint16_t)last; p[O + 2] = (uint8_t)(last >> 16); }
    }
}

// ---- component access inside a unit (all indices fold at compile time) ------
template <int BITS> __device__ __forceinline__ u32 get_comp(const u32* w, int idx)
{
    if constexpr (BITS == 8)  return (w[idx >> 2] >> ((idx & 3) * 8)) & 0xFFu;
    if constexpr (BITS == 16) return (w[idx >> 1] >> ((idx & 1) * 16)) & 0xFFFFu;
    return w[idx];
}
template <int BITS> __device__ __forceinline__ void put_comp(u32* w, int idx, u32 v)
{
    if constexpr (BITS == 8)       w[idx >> 2] |= (v & 0xFFu) << ((idx & 3) * 8);
    else if constexpr (BITS == 16) w[idx >> 1] |= (v & 0xFFFFu) << ((idx & 1) * 16);
    else                           w[idx] = v;
}

// x86 cvttss2si
__device__ __forceinline__ int cvtt(float x)
{
    const bool in_range = (x >= -2147483648.0f) && (x < 2147483648.0f);
    return in_range ? (int)x : (int)0x80000000;
}

// x / M, correctly rounded (== the IEEE division scanline.d performs), for integer x in [0, M], M = 255 or 65535:
// q = RN(x * RN(1/M)), one FMA residual, one FMA correction (Markstein).  Exhaustively checked against exact
// rational arithmetic for all 256 / 65536 inputs (tests/test_oracle_pinning.py::test_division_by_max_identity) and
// against the oracle on the GPU for every 16-bit value; 3 instructions instead of the ~10 of a generic division.
template <int M> __device__ __forceinline__ float div_by_max(float x)
{
    constexpr float y = 1.0f / (float)M;
    const float q = __fmul_rn(x, y);
    const float r = __fmaf_rn(-q, (float)M, x);
    return __fmaf_rn(r, y, q);
}

// c / a, correctly rounded, for the quotients 8-bit premultiplied sources produce: c = RN(i / 255), a = RN(j / 255), 0 <= i <= 255,
// 1 <= j <= 255.  One reciprocal per PIXEL (v_rcp_f32, 1 ulp) and per channel a product, an FMA residual and an FMA correction
// (Markstein) instead of a full IEEE division per channel (v_div_scale x2, v_rcp, 4 FMAs, v_div_fmas, v_div_fixup).  That the
// three instructions give RN(c / a) is not a theorem for a 1-ulp reciprocal; it is CHECKED: tests/test_convert_gpu.py runs every
// (i, j) pair of every 8-bit premultiplied source against the oracle's IEEE division (65 280 quotients, all bit-equal).
__device__ __forceinline__ float div_by_alpha8(float c, float a, float ra)
{
    const float q = __fmul_rn(c, ra);
    const float r = __fmaf_rn(-q, a, c);
    return __fmaf_rn(r, ra, q);
}

struct RGBAf { float r, g, b, a; };
struct RGBA8 { u32 r, g, b, a; };

// scanline.d:240-529 ("xxx_to_rgbaf32")
template <int T> __device__ __forceinline__ RGBAf decode_f32(const u32* w, int p)
{
    constexpr int CH = PT<T>::ch, BITS = PT<T>::bits;
    float c[4];
    #pragma unroll
    for (int k = 0; k < CH; ++k) {
        const u32 raw = get_comp<BITS>(w, p * CH + k);
        if constexpr (BITS == 8)       c[k] = div_by_max<255>((float)(int)raw);
        else if constexpr (BITS == 16) c[k] = div_by_max<65535>((float)(int)raw);
        else                           c[k] = __uint_as_float(raw);
    }
    RGBAf o;
    if constexpr (CH == 1)      { o.r = o.g = o.b = c[0]; o.a = 1.0f; }
    else if constexpr (CH == 2) {
        o.r = c[0]; o.a = c[1];
        if constexpr (PT<T>::premul) {
            if constexpr (BITS == 8) { const float q = div_by_alpha8(o.r, o.a, __builtin_amdgcn_rcpf(o.a)); o.r = o.a != 0.0f ? q : o.r; }
            else { if (o.a != 0.0f) o.r = __fdiv_rn(o.r, o.a); }
        }
        o.g = o.b = o.r;
    }
    else if constexpr (CH == 3) { o.r = c[0]; o.g = c[1]; o.b = c[2]; o.a = 1.0f; }
    else {
        o.r = c[0]; o.g = c[1]; o.b = c[2]; o.a = c[3];
        if constexpr (PT<T>::premul) {
            if constexpr (BITS == 8) {
                const float ra = __builtin_amdgcn_rcpf(o.a);
                const float qr = div_by_alpha8(o.r, o.a, ra), qg = div_by_alpha8(o.g, o.a, ra), qb = div_by_alpha8(o.b, o.a, ra);
                const bool nz = o.a != 0.0f;
                o.r = nz ? qr : o.r; o.g = nz ? qg : o.g; o.b = nz ? qb : o.b;
            } else { if (o.a != 0.0f) { o.r = __fdiv_rn(o.r, o.a); o.g = __fdiv_rn(o.g, o.a); o.b = __fdiv_rn(o.b, o.a); } }
        }
    }
    return o;
}

// scanline.d:539-803 ("rgbaf32_to_xxx"); evaluation order as written there
template <int BITS> __device__ __forceinline__ u32 quant(float v)   // cast(T)(0.5f + v)
{
    const int i = cvtt(__fadd_rn(0.5f, v));
    return BITS == 8 ? ((u32)i & 0xFFu) : ((u32)i & 0xFFFFu);
}
template <int T> __device__ __forceinline__ void encode_f32(u32* w, int p, const RGBAf& v)
{
    constexpr int CH = PT<T>::ch, BITS = PT<T>::bits;
    constexpr float M = BITS == 8 ? 255.0f : 65535.0f;
    constexpr bool PRE = PT<T>::premul;
    if constexpr (CH <= 2) {
        float s = __fadd_rn(__fadd_rn(v.r, v.g), v.b);            // (r + g + b)
        if constexpr (PRE) s = __fmul_rn(s, v.a);                 // * a
        if constexpr (BITS == 32) {
            put_comp<32>(w, p * CH, __float_as_uint(__fdiv_rn(s, 3.0f)));
            if constexpr (CH == 2) put_comp<32>(w, p * CH + 1, __float_as_uint(v.a));
        } else {
            put_comp<BITS>(w, p * CH, quant<BITS>(__fdiv_rn(__fmul_rn(s, M), 3.0f)));
            if constexpr (CH == 2) put_comp<BITS>(w, p * CH + 1, quant<BITS>(__fmul_rn(v.a, M)));
        }
    } else {
        float c[4] = { v.r, v.g, v.b, v.a };
        #pragma unroll
        for (int k = 0; k < CH; ++k) {
            float x = c[k];
            if constexpr (PRE) { if (k < 3) x = __fmul_rn(x, v.a); }
            if constexpr (BITS == 32) put_comp<32>(w, p * CH + k, __float_as_uint(x));
            else                      put_comp<BITS>(w, p * CH + k, quant<BITS>(__fmul_rn(x, M)));
        }
    }
}

// scanline.d:160-194 / :201-234 (plain 8-bit types through rgba8; l8 <- R only)
template <int T> __device__ __forceinline__ RGBA8 decode_u8(const u32* w, int p)
{
    constexpr int CH = PT<T>::ch;
    RGBA8 o;
    if constexpr (CH == 1)      { o.r = o.g = o.b = get_comp<8>(w, p); o.a = 255; }
    else if constexpr (CH == 2) { o.r = o.g = o.b = get_comp<8>(w, 2*p); o.a = get_comp<8>(w, 2*p + 1); }
    else if constexpr (CH == 3) { o.r = get_comp<8>(w, 3*p); o.g = get_comp<8>(w, 3*p+1); o.b = get_comp<8>(w, 3*p+2); o.a = 255; }
    else                        { o.r = get_comp<8>(w, 4*p); o.g = get_comp<8>(w, 4*p+1); o.b = get_comp<8>(w, 4*p+2); o.a = get_comp<8>(w, 4*p+3); }
    return o;
}
template <int T> __device__ __forceinline__ void encode_u8(u32* w, int p, const RGBA8& v)
{
    constexpr int CH = PT<T>::ch;
    if constexpr (CH == 1)      { put_comp<8>(w, p, v.r); }
    else if constexpr (CH == 2) { put_comp<8>(w, 2*p, v.r); put_comp<8>(w, 2*p+1, v.a); }
    else if constexpr (CH == 3) { put_comp<8>(w, 3*p, v.r); put_comp<8>(w, 3*p+1, v.g); put_comp<8>(w, 3*p+2, v.b); }
    else                        { put_comp<8>(w, 4*p, v.r); put_comp<8>(w, 4*p+1, v.g); put_comp<8>(w, 4*p+2, v.b); put_comp<8>(w, 4*p+3, v.a); }
}

template <int S, int D, int G>
__device__ __forceinline__ void convert_unit(const u32* in, u32* out)
{
    constexpr int DW = (G * PT<D>::size + 3) / 4;
    #pragma unroll
    for (int i = 0; i < DW; ++i) out[i] = 0;
    #pragma unroll
    for (int p = 0; p < G; ++p) {
        if constexpr (PT<S>::plain8 && PT<D>::plain8) encode_u8<D>(out, p, decode_u8<S>(in, p));
        else                                            encode_f32<D>(out, p, decode_f32<S>(in, p));
    }
}

struct ConvArgs {
    const uint8_t* src; uint8_t* dst;
    int64_t srcPitch, srcLayer, dstPitch, dstLayer;
    int64_t total;          // rows * upr
    u32 upr;                // units per row incl. the tail unit
    u32 full_units;         // width / G
    u32 tail_px;            // width % G
    u32 height;
    u32 rows;
};

// byte-granular single-pixel path (row tails, and everything when alignment forbids vectors)
template <int S, int D>
__device__ __forceinline__ void convert_pixel_bytes(const uint8_t* s, uint8_t* d)
{
    constexpr int SS = PT<S>::size, DS = PT<D>::size;
    u32 in[(SS + 3) / 4] = {}, out[(DS + 3) / 4] = {};
    #pragma unroll
    for (int i = 0; i < SS; ++i) in[i >> 2] |= (u32)s[i] << ((i & 3) * 8);
    if constexpr (PT<S>::plain8 && PT<D>::plain8) encode_u8<D>(out, 0, decode_u8<S>(in, 0));
    else                                            encode_f32<D>(out, 0, decode_f32<S>(in, 0));
    #pragma unroll
    for (int i = 0; i < DS; ++i) d[i] = (uint8_t)(out[i >> 2] >> ((i & 3) * 8));
}

// Launch shape (tools/copy_probe.hip, tools/conv_var.sh): on MI355X a streaming kernel whose grid covers the data once -- every
// thread a fixed, small number of units, blocks retired in address order -- runs a plain copy at 6.3 TB/s, the same loop in a
// persistent grid-stride grid at 5.0-5.7; the conversions follow (4.8 -> 5.3-6.5 TB/s).  Two units per thread and nontemporal
// loads and stores were the best of {1, 2, 4, 8} x {plain, nt stores, nt both} over nine pairs.
#ifndef CONVERT_UNROLL
#define CONVERT_UNROLL 2
#endif
constexpr int kThreads = 256;
constexpr int kUnroll  = CONVERT_UNROLL;

// A destination unit of 3, 6, 12, 24 ... bytes per lane (rgb8 / rgb16 / rgbf32 pixels) leaves every store instruction of the wave
// with holes: dwordx2 pieces at a 24-byte lane stride, three instructions that each touch every line of the wave's 1.5 KiB
// (l8 -> rgbf32: 2.9 TB/s).  Such pairs hand their units through LDS inside the wave, so that every store instruction writes
// 64 x 16 contiguous bytes.
constexpr bool staged_store(int db) { return db % 16 != 0 && db != 1 && db != 2 && db != 4 && db != 8; }
#ifndef CONVERT_STAGE
#define CONVERT_STAGE 1
#endif
__device__ __forceinline__ void convert_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int S, int D>
__global__ __launch_bounds__(kThreads) void k_convert_vec(ConvArgs a)
{
    constexpr int G  = unit_pixels(PT<S>::size, PT<D>::size);
    constexpr int SB = G * PT<S>::size, DB = G * PT<D>::size;
    constexpr int64_t stride = kThreads;
    constexpr bool STAGE = CONVERT_STAGE && staged_store(DB);
    __shared__ __attribute__((aligned(16))) uint8_t stage[STAGE ? (kThreads / 64) * 64 * DB : 16];
    {
        const int64_t base = (int64_t)blockIdx.x * (kThreads * kUnroll) + threadIdx.x;      // a block owns kUnroll * 256 consecutive units
        u32 in[kUnroll][(SB + 3) / 4];
        const uint8_t* sp[kUnroll]; uint8_t* dp[kUnroll];
        u32 uidx[kUnroll]; bool live[kUnroll];
        #pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            const int64_t idx = base + j * stride;
            live[j] = idx < a.total;
            u32 row = 0, u = (u32)idx;
            if (a.rows > 1) {
                row = (a.total <= 0xFFFFFFFFLL) ? (u32)idx / a.upr : (u32)(idx / a.upr);
                u = (u32)(idx - (int64_t)row * a.upr);
            }
            u32 layer = 0, y = row;
            if (row >= a.height) { layer = row / a.height; y = row - layer * a.height; }
            sp[j] = a.src + layer * a.srcLayer + (int64_t)y * a.srcPitch + (int64_t)u * SB;
            dp[j] = a.dst + layer * a.dstLayer + (int64_t)y * a.dstPitch + (int64_t)u * DB;
            uidx[j] = u;
            if (live[j] && u < a.full_units) load_unit<SB>(sp[j], in[j]);
        }
        #pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            if (!live[j]) continue;
            if (uidx[j] < a.full_units) {
                u32 out[(DB + 3) / 4];
                convert_unit<S, D, G>(in[j], out);
                if constexpr (STAGE) {
                    // (wave-uniform) 64 whole units, one after the other in one row: the wave's bytes are one contiguous run
                    const u32 u0 = (u32)__builtin_amdgcn_readfirstlane((int)uidx[j]), lane = threadIdx.x & 63u;
                    if (__ballot(uidx[j] == u0 + lane && uidx[j] < a.full_units) == ~0ull) {          // (lanes that are not here count as 0)
                        uint8_t* mine = stage + (threadIdx.x >> 6) * (64 * DB);
                        if constexpr (DB % 4 == 0) { _Pragma("unroll") for (int i = 0; i < DB / 4; ++i) reinterpret_cast<u32*>(mine + lane * DB)[i] = out[i]; }
                        else if constexpr (DB % 2 == 0) { _Pragma("unroll") for (int i = 0; i < DB / 2; ++i) reinterpret_cast<uint16_t*>(mine + lane * DB)[i] = (uint16_t)(out[i >> 1] >> ((i & 1) * 16)); }
                        else { _Pragma("unroll") for (int i = 0; i < DB; ++i) mine[lane * DB + i] = (uint8_t)(out[i >> 2] >> ((i & 3) * 8)); }
                        convert_wave_sync();
                        const uint64_t wd = (uint64_t)dp[j];
                        uint8_t* run = reinterpret_cast<uint8_t*>((uint64_t)(u32)__builtin_amdgcn_readfirstlane((int)(u32)wd) |
                                                                  (uint64_t)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(wd >> 32)) << 32);
                        struct __attribute__((packed, aligned(1))) Any16 { u32x4v v; };
                        _Pragma("unroll")
                        for (int off = 0; off < 64 * DB; off += 1024) {
                            const int o = off + (int)lane * 16;
                            if (o < 64 * DB) reinterpret_cast<Any16*>(run + o)->v = *reinterpret_cast<const u32x4v*>(mine + o);
                        }
                        convert_wave_sync();
                        continue;
                    }
                }
                store_unit<DB>(dp[j], out);
            } else {
                for (u32 p = 0; p < a.tail_px; ++p)
                    convert_pixel_bytes<S, D>(sp[j] + p * PT<S>::size, dp[j] + p * PT<D>::size);
            }
        }
    }
}

// alignment-free fallback: one pixel per thread, byte accesses
template <int S, int D>
__global__ __launch_bounds__(kThreads) void k_convert_bytes(ConvArgs a, u32 width)
{
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    const int64_t total = (int64_t)a.rows * width;
    for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += stride) {
        const u32 row = (u32)(idx / width), x = (u32)(idx - (int64_t)row * width);
        const u32 layer = row / a.height, y = row - layer * a.height;
        convert_pixel_bytes<S, D>(a.src + layer * a.srcLayer + (int64_t)y * a.srcPitch + (int64_t)x * PT<S>::size,
                                  a.dst + layer * a.dstLayer + (int64_t)y * a.dstPitch + (int64_t)x * PT<D>::size);
    }
}

// K9 scanlinesCopy: same type, re-layout only (pitch / v-flip / border)
__global__ __launch_bounds__(kThreads) void k_copy_rows(ConvArgs a, u32 row_bytes, int vec)
{
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    const u32 upr = (row_bytes + vec - 1) / vec;
    const int64_t total = (int64_t)a.rows * upr;
    for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += stride) {
        const u32 row = (u32)(idx / upr), u = (u32)(idx - (int64_t)row * upr);
        const u32 layer = row / a.height, y = row - layer * a.height;
        const uint8_t* s = a.src + layer * a.srcLayer + (int64_t)y * a.srcPitch + (int64_t)u * vec;
        uint8_t* d = a.dst + layer * a.dstLayer + (int64_t)y * a.dstPitch + (int64_t)u * vec;
        if (vec == 16 && (u + 1) * 16u <= row_bytes) *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
        else {
            const u32 n = min((u32)vec, row_bytes - u * vec);
            for (u32 i = 0; i < n; ++i) d[i] = s[i];
        }
    }
}

inline int grid_for(int64_t work_items)                  // one 256-thread block per 256 items: the grid covers the data once
{
    int64_t blocks = (work_items + kThreads - 1) / kThreads;
    if (blocks > 0x7FFFFFFFLL) blocks = 0x7FFFFFFFLL;    // (the kernels with a loop then go round again; 32 GiB images stay below)
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

inline bool aligned(const void* p, int64_t pitch, int64_t layer, int rows, int layers, int a)
{
    if (((uintptr_t)p) % a) return false;
    if (rows > 1 && (pitch % a)) return false;
    if (layers > 1 && (layer % a)) return false;
    return true;
}

template <int S, int D>
int launch_pair(ConvArgs a, int width, int height, int layers, hipStream_t stream)
{
    constexpr int G  = unit_pixels(PT<S>::size, PT<D>::size);
    constexpr int SB = G * PT<S>::size, DB = G * PT<D>::size;
    // collapse gapless top-down storage into one long row (the common case: decoder outputs,
    // LAYOUT_GAPLESS images) so only the base pointers need alignment
    int64_t w = width; int rows = height * layers; int h = height;
    const bool gapless = a.srcPitch == (int64_t)width * PT<S>::size && a.dstPitch == (int64_t)width * PT<D>::size &&
                         (layers == 1 || (a.srcLayer == a.srcPitch * height && a.dstLayer == a.dstPitch * height));
    // More gapless layers than one launch can index with 32 bits (158 layers of 8192 x 8192 rgba8 -> rgba16 are 5.3 G units): as one launch they take the
    // rows-and-layers form, whose unit index needs a 64-bit division per unit -- measured 5-7 % on an HBM-bound kernel (256 layers in chunks of 158: 0.71-0.74 of
    // peak, in chunks of 79: 0.77-0.78; profiles/r06_convert_chunk_size.txt).  So such a batch goes as several launches of as many whole layers as fit.
    if (gapless && layers > 1 && (int64_t)width * rows / G >= 0xFFFFFFF0LL) {
        const int64_t per_layer = (int64_t)width * height / G + 1;
        const int per = (int)std::max<int64_t>(1, 0xFFFFFFF0LL / per_layer - 1);
        if (per < layers) {
            for (int l0 = 0; l0 < layers; l0 += per) {
                ConvArgs b = a;
                b.src = a.src + (int64_t)l0 * a.srcLayer; b.dst = a.dst + (int64_t)l0 * a.dstLayer;
                if (const int rc = launch_pair<S, D>(b, width, height, std::min(per, layers - l0), stream)) return rc;
            }
            return GAMUT_HIP_OK;
        }
    }
    if (gapless && (int64_t)width * rows / G < 0xFFFFFFF0LL) { w = (int64_t)width * rows; rows = 1; h = 1; }
    a.rows = (u32)rows; a.height = (u32)h;
    const bool vec_ok = aligned(a.src, a.srcPitch, a.srcLayer, rows, rows > h ? 2 : 1, vec_bytes(SB)) &&
                        aligned(a.dst, a.dstPitch, a.dstLayer, rows, rows > h ? 2 : 1, vec_bytes(DB));
    if (vec_ok) {
        a.full_units = (u32)(w / G); a.tail_px = (u32)(w % G);
        a.upr = a.full_units + (a.tail_px ? 1 : 0);
        a.total = (int64_t)rows * a.upr;
        if (a.total == 0) return GAMUT_HIP_OK;
        const int grid = grid_for((a.total + kUnroll - 1) / kUnroll);
        hipLaunchKernelGGL((k_convert_vec<S, D>), dim3(grid), dim3(kThreads), 0, stream, a);
    } else {
        const int64_t total = (int64_t)rows * w;
        if (total == 0) return GAMUT_HIP_OK;
        hipLaunchKernelGGL((k_convert_bytes<S, D>), dim3(grid_for(total)), dim3(kThreads), 0, stream, a, (u32)w);
    }
    return launch_status("scanlines_convert");
}

template <int S, int Dd>
bool try_pair(int dstType, const ConvArgs& a, int w, int h, int l, hipStream_t st, int& rc)
{
    if constexpr (S != Dd) {          // same-type pairs take the copy path and never reach here
        if (dstType == Dd) { rc = launch_pair<S, Dd>(a, w, h, l, st); return true; }
    }
    return false;
}
template <int S, int... Ds>
int dispatch_dst(int dstType, std::integer_sequence<int, Ds...>, const ConvArgs& a, int w, int h, int l, hipStream_t st)
{
    int rc = GAMUT_HIP_ERR_INVALID_ARG;
    (void)(try_pair<S, Ds>(dstType, a, w, h, l, st, rc) || ...);
    return rc;
}
template <int... Ss>
int dispatch_src(int srcType, int dstType, std::integer_sequence<int, Ss...>, const ConvArgs& a, int w, int h, int l, hipStream_t st)
{
    int rc = GAMUT_HIP_ERR_INVALID_ARG;
    (void)((srcType == Ss ? (rc = dispatch_dst<Ss>(dstType, std::make_integer_sequence<int, GAMUT_PIXEL_COUNT>{}, a, w, h, l, st), true) : false) || ...);
    return rc;
}

} // namespace

int convert_device(int srcType, const void* src, int64_t srcPitch, int64_t srcLayerOffset,
                   int dstType, void* dst, int64_t dstPitch, int64_t dstLayerOffset,
                   int width, int height, int layers, hipStream_t stream)
{
    if (!valid_type(srcType) || !valid_type(dstType))
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "scanlines_convert: invalid PixelType %d -> %d", srcType, dstType);
    if (width < 0 || height < 0 || layers < 0)
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "scanlines_convert: negative dimension");
    if (width == 0 || height == 0 || layers == 0) return GAMUT_HIP_OK;
    if (!src || !dst) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "scanlines_convert: null pointer");

    ConvArgs a{};
    a.src = static_cast<const uint8_t*>(src); a.dst = static_cast<uint8_t*>(dst);
    a.srcPitch = srcPitch; a.srcLayer = srcLayerOffset; a.dstPitch = dstPitch; a.dstLayer = dstLayerOffset;

    if (srcType == dstType) {                                  // scanlinesCopy, scanline.d:75-78
        const u32 row_bytes = (u32)width * kPixelSize[srcType];
        int64_t rows = (int64_t)height * layers; u32 h = (u32)height; u32 rb = row_bytes;
        const bool gapless = srcPitch == row_bytes && dstPitch == row_bytes &&
                             (layers == 1 || (srcLayerOffset == srcPitch * height && dstLayerOffset == dstPitch * height));
        if (gapless && (int64_t)row_bytes * rows < 0xFFFFFFF0LL) { rb = (u32)(row_bytes * rows); rows = 1; h = 1; }
        a.rows = (u32)rows; a.height = h;
        const int layers_eff = rows > h ? 2 : 1;
        const int vec = (aligned(src, srcPitch, srcLayerOffset, (int)rows, layers_eff, 16) &&
                         aligned(dst, dstPitch, dstLayerOffset, (int)rows, layers_eff, 16)) ? 16 : 1;
        const int64_t total = rows * ((rb + vec - 1) / vec);
        hipLaunchKernelGGL(k_copy_rows, dim3(grid_for(total)), dim3(kThreads), 0, stream, a, rb, vec);
        return launch_status("scanlines_copy");
    }
    return dispatch_src(srcType, dstType, std::make_integer_sequence<int, GAMUT_PIXEL_COUNT>{}, a, width, height, layers, stream);
}

} // namespace gamut
