// inflate.hip -- DEFLATE (RFC 1951) on the GPU, for batches of PNG files.
//
// Replaces, for a batch, stbi_zlib_decode_malloc_guesssize_headerflag (stbdec.d:1267-1321 -> the `miniz` inflate): IDAT
// streams resident in HBM are inflated straight into the arena the de-filter kernels read (png.hip), so a file batch no
// longer waits for sixteen host threads running zlib.  One workgroup of kT = 512 threads per stream; inside a stream:
//
//   * block headers and the code-length alphabet of a dynamic block are read by one thread (a few hundred bits); the two
//     canonical Huffman codes are turned into lookup tables by all threads (counting by LDS atomics, ranks, one table entry
//     per thread and pass);
//   * the block's symbols are decoded 8 KiB at a time (kT lanes x 128 bits), SPECULATIVELY: lane k starts at bit 128 k of the chunk as if a
//     token began there, lanes then restart from their predecessor's exit position until nothing moves -- Huffman streams
//     re-synchronise within a few tokens, so a few sweeps settle all the lanes (the chain is exact from lane 0 on,
//     and a lane that meets the end-of-block code, an invalid code or its output cap ends the chunk there);
//   * a prefix sum of the lanes' output bytes places everything: a last sweep writes the literals into a 64 KiB ring in LDS
//     (the 32 KiB window plus the chunk's output) and, for every byte a match produces, the position it copies from;
//   * the copies are resolved for all bytes at once by pointer doubling: a byte whose source is known takes its value, any
//     other byte adopts its source's source -- a chain of n dependent copies (runs, distance-1 matches, the short distances
//     of filtered scanlines) is done after log2 n rounds, at most 15, whatever the match structure;
//   * the chunk's bytes leave the ring for HBM in dwords.
// No data-dependent branch is taken on a whole stream's behalf by a single lane except the block headers.
#include "common.hpp"

namespace gamut {
namespace {

#ifndef INFLATE_T                 // tuning knobs (tools/variant.sh)
#define INFLATE_T 512
#endif
#ifndef INFLATE_SUB_BITS
#define INFLATE_SUB_BITS 128
#endif
#ifndef INFLATE_NEW_MAX
#define INFLATE_NEW_MAX 28672
#endif
constexpr int kT = INFLATE_T;                        // threads per stream
constexpr int kSubBits = INFLATE_SUB_BITS;           // compressed bits a lane owns per chunk
constexpr int kChunkBytes = kT * kSubBits / 8;       // 8192
constexpr int kWinDwords = kChunkBytes / 4 + 8;      // + 32 bytes: the last lane runs up to 47 bits past its end and peeks 64 bits from there
constexpr int kRing = 65536, kRingMask = kRing - 1;
constexpr int kNewMax = INFLATE_NEW_MAX;              // bytes a chunk may add to the ring: the 32 KiB history must survive them
constexpr int kLaneOutMax = 8192;                    // a lane stops early beyond this (ends the chunk: pathological match runs)
constexpr int kHist = 32768;                         // DEFLATE's window
#ifndef INFLATE_LONG_CAP
#define INFLATE_LONG_CAP 1024
#endif
#ifndef INFLATE_LIT_BITS
#define INFLATE_LIT_BITS 11
#endif
constexpr int kLongMin = 32, kLongCap = INFLATE_LONG_CAP;        // matches at least this long are expanded by a whole wave, not by their lane
constexpr int kLitBits = INFLATE_LIT_BITS, kDistBits = 10;         // primary lookup widths; longer codes take the canonical search

enum : uint32_t { F_EOB = 1, F_BAD = 2, F_EARLY = 4 };
// status word per stream (0 = ok)
enum : uint32_t { E_BLOCK_TYPE = 1, E_STORED = 2, E_LENGTHS = 3, E_CODE = 4, E_DISTANCE = 5, E_INPUT = 6 };
enum { C_FIRST_BAD = 0, C_FIRST_STOP, C_CUT, C_OPEN, C_NLONG, C_ERR, C_BTYPE, C_FINAL, C_HDR_END_LO, C_HDR_END_HI, C_HLIT, C_HDIST, C_N };

struct InfItem { const uint8_t* src; uint8_t* dst; uint32_t src_len, dst_cap; };

#ifndef INFLATE_PROFILE           // measurement only (tools/variant.sh inflate:prof:-DINFLATE_PROFILE=1): cycles per phase, summed over streams
#define INFLATE_PROFILE 0
#endif
enum { P_HEADER = 0, P_TABLES, P_WINDOW, P_SWEEP0, P_SWEEPS, P_SCAN, P_WRITE, P_MATCH, P_FLUSH, P_STORED, P_N_BLOCKS, P_N_CHUNKS, P_N_SWEEPS, P_N_ROUNDS, P_N_MATCHES, P_N };
#if INFLATE_PROFILE
__device__ unsigned long long g_inflate_prof[P_N];
#define PROF_DECL unsigned long long prof_t0 = clock64(), prof_acc[P_N] = {}
#define PROF(slot) do { const unsigned long long now_ = clock64(); prof_acc[slot] += now_ - prof_t0; prof_t0 = now_; } while (0)
#define PROF_COUNT(slot, n) (prof_acc[slot] += (n))
#define PROF_FLUSH do { if (threadIdx.x == 0) for (int k_ = 0; k_ < P_N; ++k_) atomicAdd(&g_inflate_prof[k_], prof_acc[k_]); } while (0)
#else
#define PROF_DECL
#define PROF(slot)
#define PROF_COUNT(slot, n)
#define PROF_FLUSH
#endif

struct Canon { uint32_t first[16], count[16], offs[16]; };

struct Shared {
    __attribute__((aligned(16))) uint8_t ring[kRing];
    __attribute__((aligned(16))) uint32_t win[kWinDwords];
    uint32_t lit_lut[1 << kLitBits];
    uint32_t dist_lut[1 << kDistBits];
    __attribute__((aligned(16))) uint16_t from[kNewMax + 8];                          // per byte of the chunk: where its value comes from, as a position in [chunk start - 32768, ...); itself = known
    uint32_t exit_bit[kT], flags[kT];
    uint32_t scan_a[kT];
    uint2    longm[kLongCap];                        // long matches of the chunk: x = first byte (chunk-relative), y = length | distance << 16
    uint32_t lit_sorted[288], dist_sorted[32];       // symbol payloads in canonical order
    Canon    lit, dist;
    uint32_t cl_lut[128];
    uint32_t ctrl[C_N];
    uint8_t  lens[320];
};

// 64 stream bits from bit `bit` of the window (LSB first, as DEFLATE packs them)
__device__ __forceinline__ uint64_t peek64(const uint32_t* win, uint32_t bit)
{
    const uint32_t w = bit >> 5, sh = bit & 31u;
    const uint32_t a = win[w], b = win[w + 1], c = win[w + 2];
    return (uint64_t)__builtin_amdgcn_alignbit(c, b, sh) << 32 | __builtin_amdgcn_alignbit(b, a, sh);
}

// symbol -> table payload: bits 4-6 kind (0 literal, 1 length / distance, 2 end of block, 3 invalid), 8-12 extra bits, 16-31 base
__device__ __forceinline__ uint32_t lit_payload(uint32_t s)
{
    if (s < 256u) return s << 16;
    if (s == 256u) return 2u << 4;
    if (s > 285u) return 3u << 4;
    const uint32_t i = s - 257u;
    if (i < 8u) return (3u + i) << 16 | 1u << 4;
    if (i == 28u) return 258u << 16 | 1u << 4;
    const uint32_t x = (i - 4u) >> 2;
    return (3u + ((4u + (i & 3u)) << x)) << 16 | x << 8 | 1u << 4;
}
__device__ __forceinline__ uint32_t dist_payload(uint32_t d)
{
    if (d > 29u) return 3u << 4;
    if (d < 4u) return (1u + d) << 16 | 1u << 4;
    const uint32_t x = (d - 2u) >> 1;
    return (1u + ((2u + (d & 1u)) << x)) << 16 | x << 8 | 1u << 4;
}

// canonical search over code lengths [from, to]: `bits` holds the stream bits LSB first; 0 = no code of these lengths matches
__device__ __forceinline__ uint32_t canon_decode(uint32_t bits, const Canon& c, const uint32_t* sorted, int from, int to)
{
    const uint32_t rev = __brev(bits);
    for (int l = from; l <= to; ++l) {
        const uint32_t d = (rev >> (32 - l)) - c.first[l];
        if (d < c.count[l]) return sorted[c.offs[l] + d] | (uint32_t)l;
    }
    return 0;
}

// canonical code of `n` symbols with lengths S.lens[base .. base + n): Canon, payloads in canonical order, primary table.
// Validity as zlib's inflate_table: over-subscribed sets fail; incomplete ones too, unless the set is a single 1-bit code.
template <bool DIST>
__device__ void build_table(Shared& S, int base, int n)
{
    const int t = threadIdx.x;
    Canon& c = DIST ? S.dist : S.lit;
    uint32_t* sorted = DIST ? S.dist_sorted : S.lit_sorted;
    uint32_t* lut = DIST ? S.dist_lut : S.lit_lut;
    constexpr int P = DIST ? kDistBits : kLitBits;
    if (t < 16) c.count[t] = 0;
    __syncthreads();
    for (int s = t; s < n; s += kT) { const uint32_t L = S.lens[base + s]; if (L) atomicAdd(&c.count[L], 1u); }
    __syncthreads();
    if (t == 0) {
        int left = 1, maxlen = 0; bool over = false;
        for (int l = 1; l <= 15; ++l) { left = left * 2 - (int)c.count[l]; if (left < 0) over = true; if (c.count[l]) maxlen = l; }
        if (over || (left > 0 && maxlen > 1)) atomicMax(&S.ctrl[C_ERR], (uint32_t)E_LENGTHS);
        c.first[0] = 0; c.offs[0] = 0; c.first[1] = 0; c.offs[1] = 0;
        for (int l = 2; l <= 15; ++l) { c.first[l] = (c.first[l - 1] + c.count[l - 1]) << 1; c.offs[l] = c.offs[l - 1] + c.count[l - 1]; }
    }
    __syncthreads();
    for (int s = t; s < n; s += kT) {
        const uint32_t L = S.lens[base + s];
        if (!L) continue;
        uint32_t rank = 0;
        for (int k = 0; k < s; ++k) rank += S.lens[base + k] == L;
        sorted[c.offs[L] + rank] = DIST ? dist_payload((uint32_t)s) : lit_payload((uint32_t)s);
    }
    __syncthreads();
    for (int e = t; e < (1 << P); e += kT) {
        uint32_t r = canon_decode((uint32_t)e, c, sorted, 1, P);
        if constexpr (!DIST) {
            // two literals behind one look-up when both codes fit the index (kind 4: base = first | second << 8): residual data
            // of photographs is mostly literals of 3-6 bits
            const uint32_t l1 = r & 15u;
            if (r && ((r >> 4) & 7u) == 0u && l1 < (uint32_t)P) {
                const uint32_t r2 = canon_decode((uint32_t)e >> l1, c, sorted, 1, P - (int)l1);
                if (r2 && ((r2 >> 4) & 7u) == 0u) r = ((r >> 16) | (r2 >> 16) << 8) << 16 | l1 << 8 | 4u << 4 | (l1 + (r2 & 15u));
            }
        }
        lut[e] = r;
    }
    __syncthreads();
}

struct LaneResult { uint32_t exit, flags, out; };

// the code lengths the primary table does not cover, in registers (the canonical search then costs no LDS round trip but the last)
template <int FROM> struct LongCodes {
    uint32_t first[16 - FROM], count[16 - FROM], offs[16 - FROM];
    __device__ __forceinline__ void load(const Canon& c)
    {
        #pragma unroll
        for (int l = FROM; l <= 15; ++l) { first[l - FROM] = c.first[l]; count[l - FROM] = c.count[l]; offs[l - FROM] = c.offs[l]; }
    }
    __device__ __forceinline__ uint32_t decode(uint32_t bits, const uint32_t* sorted) const
    {
        const uint32_t rev = __brev(bits);
        uint32_t at = 0xFFFFFFFFu, len = 0;
        #pragma unroll
        for (int l = 15; l >= FROM; --l) {                         // (at most one length matches: the code is prefix-free)
            const uint32_t d = (rev >> (32 - l)) - first[l - FROM];
            const bool hit = d < count[l - FROM];
            at = hit ? offs[l - FROM] + d : at; len = hit ? (uint32_t)l : len;
        }
        return len ? sorted[at] | len : 0u;
    }
};

// tokens of one lane: from bit `start` until a token begins at or beyond `end` (window-relative bits).  WRITE: literals into the
// ring at absolute output offset obase.., and S.from[] for every byte (orel = obase - chunk start)
// MODE 0: count only (the sweeps); 1: write (see above); 2: nothing is written, but distances are still checked against the
// bytes produced so far (the part of a stream beyond the caller's capacity is decoded to its end all the same: a stream
// damaged anywhere is a corrupt stream, as it is for the reference, which inflates all of it)
template <int MODE>
__device__ __forceinline__ LaneResult lane_decode(Shared& S, uint32_t start, uint32_t end, uint32_t obase, uint32_t orel)
{
    constexpr bool WRITE = MODE == 1;
    LongCodes<kLitBits + 1> long_lit; long_lit.load(S.lit);
    LongCodes<kDistBits + 1> long_dist; long_dist.load(S.dist);
    uint32_t pos = start, o = 0, fl = 0;
    while (pos < end && !fl) {
        uint64_t bits = peek64(S.win, pos);
        uint32_t used = 0;
        do {
            uint32_t e = S.lit_lut[(uint32_t)bits & ((1u << kLitBits) - 1u)];
            if ((e & 15u) == 0u) { e = long_lit.decode((uint32_t)bits, S.lit_sorted); if (!e) { fl = F_BAD; break; } }
            uint32_t nb = e & 15u;
            uint32_t kind = (e >> 4) & 7u;
            if (kind == 4u) {
                // a pair whose second literal would begin at or beyond `end` is taken as its first literal alone: a lane must
                // leave at the FIRST token boundary past its end whatever way it came in, or the lanes never fall into step
                const uint32_t l1 = (e >> 8) & 15u;
                if (pos + used + l1 >= end) { kind = 0u; nb = l1; e &= 0x00FFFFFFu; }
            }
            bits >>= nb; used += nb;
            if (kind == 0u) {
                if (WRITE) { S.ring[(obase + o) & kRingMask] = (uint8_t)(e >> 16); S.from[orel + o] = (uint16_t)(orel + o + kHist); }
                ++o;
            } else if (kind == 4u) {                              // two literals
                if (WRITE) {
                    S.ring[(obase + o) & kRingMask] = (uint8_t)(e >> 16); S.from[orel + o] = (uint16_t)(orel + o + kHist);
                    S.ring[(obase + o + 1u) & kRingMask] = (uint8_t)(e >> 24); S.from[orel + o + 1u] = (uint16_t)(orel + o + 1u + kHist);
                }
                o += 2u;
            } else if (kind == 1u) {
                uint32_t xb = (e >> 8) & 31u;
                const uint32_t len = (e >> 16) + ((uint32_t)bits & ((1u << xb) - 1u));
                bits >>= xb; used += xb;
                uint32_t de = S.dist_lut[(uint32_t)bits & ((1u << kDistBits) - 1u)];
                if ((de & 15u) == 0u) { de = long_dist.decode((uint32_t)bits, S.dist_sorted); if (!de) { fl = F_BAD; break; } }
                if (((de >> 4) & 7u) != 1u) { fl = F_BAD; break; }                 // distance symbols 30 / 31
                nb = de & 15u;
                bits >>= nb; used += nb;
                xb = (de >> 8) & 31u;
                const uint32_t dist = (de >> 16) + ((uint32_t)bits & ((1u << xb) - 1u));
                bits >>= xb; used += xb;
                if (MODE != 0 && dist > obase + o) atomicMax(&S.ctrl[C_ERR], (uint32_t)E_DISTANCE);     // reaches before the first output byte
                if (WRITE) {
                    const uint32_t q = orel + o + kHist - dist;                    // position of the first source byte
                    // byte i copies byte i - dist, i.e. byte i mod dist of the dist bytes before the match: pointing there at once
                    // keeps the chains of overlapping copies (runs) one link long.  Long matches are left to a whole wave
                    // (expand_long_matches): a lane with many of them (flat images: 8 KiB from 128 bits) would hold up the chunk.
                    uint32_t slot = kLongCap;
                    if (len >= (uint32_t)kLongMin) slot = atomicAdd(&S.ctrl[C_NLONG], 1u);
                    if (slot < (uint32_t)kLongCap) S.longm[slot] = make_uint2(orel + o, len | dist << 16);
                    else for (uint32_t i = 0, m = 0; i < len; ++i) { S.from[orel + o + i] = (uint16_t)(q + m); if (++m == dist) m = 0; }
                }
                o += len;
                if (o > (uint32_t)kLaneOutMax) fl = F_EARLY;
            } else { fl = kind == 2u ? F_EOB : F_BAD; break; }
        } while (!fl && used <= 16u && pos + used < end);
        pos += used;
    }
    return LaneResult{ pos, fl, o };
}

// inclusive prefix sum over the workgroup
__device__ __forceinline__ void block_scan(Shared& S, uint32_t& a)
{
    const int t = threadIdx.x;
    S.scan_a[t] = a;
    __syncthreads();
    for (int d = 1; d < kT; d <<= 1) {
        const uint32_t pa = t >= d ? S.scan_a[t - d] : 0u;
        __syncthreads();
        a += pa;
        S.scan_a[t] = a;
        __syncthreads();
    }
}

// S.from[] for the matches the lanes left in S.longm: one match per wave and pass, 64 bytes per step
__device__ __forceinline__ void expand_long_matches(Shared& S)
{
    const uint32_t n = S.ctrl[C_NLONG] < (uint32_t)kLongCap ? S.ctrl[C_NLONG] : (uint32_t)kLongCap;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t m = wave; m < n; m += kT / 64) {
        const uint2 e = S.longm[m];
        const uint32_t at = e.x, len = e.y & 0xFFFFu, dist = e.y >> 16, q = at + kHist - dist;
        for (uint32_t i = lane; i < len; i += 64u) S.from[at + i] = (uint16_t)(q + (dist >= len ? i : i % dist));
    }
}

// Every byte of the chunk [cs, cs + total) takes its value: S.from[j] names the position byte j copies from (a position p
// counts from cs - 32768: p < 32768 is window history, always known; p == j + 32768 is the byte itself = known).  Round by
// round a byte whose source is known copies it, and any other byte adopts its source's source.  A thread looks at 8
// neighbouring bytes at a time (one 16-byte read of their entries) and passes over groups that are known already.
__device__ __forceinline__ int resolve_copies(Shared& S, uint32_t cs, uint32_t total)      // -> rounds taken
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int t = threadIdx.x;
    const uint32_t units = (total + 7u) >> 3;
    int round = 0;
    for (; round < 20; ++round) {
        if (t == 0) S.ctrl[C_OPEN] = 0;
        __syncthreads();
        bool open = false;
        for (uint32_t u = (uint32_t)t; u < units; u += kT) {
            const uint32_t j0 = u * 8u, self0 = j0 + kHist;
            const u32x4 f = *reinterpret_cast<const u32x4*>(&S.from[j0]);
            const uint32_t s0 = self0 | (self0 + 1u) << 16;
            if (f.x == s0 && f.y == s0 + 0x00020002u && f.z == s0 + 0x00040004u && f.w == s0 + 0x00060006u) continue;      // all eight known
            uint32_t p[8] = { f.x & 0xFFFFu, f.x >> 16, f.y & 0xFFFFu, f.y >> 16, f.z & 0xFFFFu, f.z >> 16, f.w & 0xFFFFu, f.w >> 16 };
            uint32_t pp[8]; uint8_t v[8];
            #pragma unroll
            for (int k = 0; k < 8; ++k) if (j0 + k >= total) p[k] = self0 + k;                 // past the chunk: nothing to do
            #pragma unroll
            for (int k = 0; k < 8; ++k) pp[k] = (p[k] != self0 + k && p[k] >= (uint32_t)kHist) ? S.from[p[k] - kHist] : p[k];
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");           // values are read after the marks that vouch for them
            #pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = S.ring[(cs + p[k] - kHist) & kRingMask];
            #pragma unroll
            for (int k = 0; k < 8; ++k)
                if (p[k] != self0 + k && pp[k] == p[k]) S.ring[(cs + j0 + k) & kRingMask] = v[k];      // the source is known: history, or a byte that points at itself
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");           // a byte is marked known only after its value is in the ring
            #pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (p[k] == self0 + k) continue;
                if (pp[k] == p[k]) S.from[j0 + k] = (uint16_t)(self0 + k);
                else { S.from[j0 + k] = (uint16_t)pp[k]; open = true; }
            }
        }
        if (open) S.ctrl[C_OPEN] = 1;
        __syncthreads();
        const bool any_open = S.ctrl[C_OPEN] != 0;
        __syncthreads();
        if (!any_open) break;
    }
    return round + 1;
}

__global__ __launch_bounds__(kT) void k_inflate(const InfItem* items, int n_items, uint32_t* out_len, uint32_t* status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Shared& S = *reinterpret_cast<Shared*>(smem);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    struct __attribute__((packed, aligned(1))) AnyVec { u32x4 v; };
    struct __attribute__((packed, aligned(1))) AnyU32 { uint32_t v; };
    const int t = threadIdx.x;
    if ((int)blockIdx.x >= n_items) return;
    const InfItem it = items[blockIdx.x];
    const uint8_t* src = it.src;
    const uint64_t src_bits = (uint64_t)it.src_len * 8u;

    uint64_t pos = 0;                                 // next unread bit of the stream
    uint32_t produced = 0;                            // bytes inflated so far (ring index = produced & kRingMask)
    uint32_t err = 0;
    bool done = false;
    if (t < C_N) S.ctrl[t] = 0;
    __syncthreads();
    PROF_DECL;

    // the window: kWinDwords dwords from the dword that holds bit `pos` (relative to the stream start); zeros past the end
    auto load_window = [&](uint64_t at_bit) -> uint64_t {
        const uint64_t base_byte = (at_bit >> 3) & ~(uint64_t)3;
        for (int q = t; q < kWinDwords / 4; q += kT) {
            const uint64_t b = base_byte + (uint64_t)q * 16u;
            u32x4 v = {0, 0, 0, 0};
            if (b + 16u <= it.src_len) v = reinterpret_cast<const AnyVec*>(src + b)->v;
            else if (b < it.src_len) {
                uint32_t w[4] = {0, 0, 0, 0};
                for (uint32_t k = 0; k < 16u && b + k < it.src_len; ++k) w[k >> 2] |= (uint32_t)src[b + k] << (8u * (k & 3u));
                v = u32x4{w[0], w[1], w[2], w[3]};
            }
            *reinterpret_cast<u32x4*>(&S.win[q * 4]) = v;
        }
        __syncthreads();
        return base_byte;
    };
    // ring bytes [from, from + n) -> dst (clamped to the caller's capacity)
    auto flush = [&](uint32_t from, uint32_t n) {
        if (from >= it.dst_cap) return;
        if (n > it.dst_cap - from) n = it.dst_cap - from;
        uint8_t* d = it.dst + from;
        const uint32_t head = (uint32_t)((4u - ((uintptr_t)d & 3u)) & 3u) < n ? (uint32_t)((4u - ((uintptr_t)d & 3u)) & 3u) : n;
        if ((uint32_t)t < head) d[t] = S.ring[(from + t) & kRingMask];
        const uint32_t body = (n - head) >> 2;
        const uint32_t* ring32 = reinterpret_cast<const uint32_t*>(S.ring);
        for (uint32_t k = t; k < body; k += kT) {
            const uint32_t i = from + head + 4u * k, r = i & kRingMask;
            const uint32_t w0 = ring32[r >> 2], w1 = ring32[((r >> 2) + 1u) & (kRing / 4 - 1)];
            reinterpret_cast<uint32_t*>(d + head)[k] = __builtin_amdgcn_alignbit(w1, w0, 8u * (r & 3u));
        }
        const uint32_t tail0 = head + 4u * body;
        if ((uint32_t)t < n - tail0) d[tail0 + t] = S.ring[(from + tail0 + t) & kRingMask];
    };

    // A stream is untrusted input, and a block costs ~40 000 cycles whatever it holds: an "image" made of empty blocks (10 bits
    // each) would keep a workgroup busy for minutes.  No encoder emits more than a block per scanline or per few KiB; streams
    // with more than one block per 8 compressed bytes (beyond the first 4096) are turned away as corrupt.  The tables of the
    // fixed code are built once per stream.
    const uint32_t block_budget = it.src_len / 8u + 4096u;
    uint32_t blocks = 0;
    bool fixed_tables = false;                        // the LDS tables hold the fixed code
    while (!done && !err) {
        if (++blocks > block_budget) { err = E_INPUT; break; }
        // ------------------------------------------------------------------ block header
        PROF(P_FLUSH);
        const uint64_t base_byte = load_window(pos);
        PROF(P_WINDOW); PROF_COUNT(P_N_BLOCKS, 1);
        if (t == 0) {
            uint32_t p = (uint32_t)(pos - base_byte * 8u);                       // window-relative bit
            uint64_t bb = peek64(S.win, p);
            const uint32_t bfinal = (uint32_t)bb & 1u, btype = ((uint32_t)bb >> 1) & 3u;
            p += 3;
            S.ctrl[C_FINAL] = bfinal; S.ctrl[C_BTYPE] = btype;
            if (btype == 2u) {
                bb = peek64(S.win, p);
                const uint32_t hlit = ((uint32_t)bb & 31u) + 257u, hdist = ((uint32_t)(bb >> 5) & 31u) + 1u, hclen = ((uint32_t)(bb >> 10) & 15u) + 4u;
                p += 14;
                S.ctrl[C_HLIT] = hlit; S.ctrl[C_HDIST] = hdist;
                uint32_t bad = (hlit > 286u || hdist > 30u) ? 1u : 0u;
                // the code-length code: 19 symbols of at most 7 bits, into a 128-entry table (entry = symbol << 4 | length)
                const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
                uint32_t cl[19]; uint32_t count[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
                for (int i = 0; i < 19; ++i) cl[i] = 0;
                bb = peek64(S.win, p);
                for (uint32_t i = 0; i < hclen; ++i) { const uint32_t v = (uint32_t)(bb >> (3u * i)) & 7u; cl[order[i]] = v; }
                p += 3u * hclen;
                for (int i = 0; i < 19; ++i) ++count[cl[i]];
                count[0] = 0;
                int left = 1;
                for (int l = 1; l <= 7; ++l) { left = left * 2 - (int)count[l]; if (left < 0) bad = 1; }
                if (left > 0) bad = 1;                                               // zlib: an incomplete code-length code is invalid
                for (int e = 0; e < 128; ++e) S.cl_lut[e] = 0;
                if (!bad) {
                    uint32_t next[8]; next[1] = 0;
                    for (int l = 2; l <= 7; ++l) next[l] = (next[l - 1] + count[l - 1]) << 1;
                    for (uint32_t s = 0; s < 19u; ++s) {
                        const uint32_t L = cl[s];
                        if (!L) continue;
                        const uint32_t code = next[L]++, rev = __brev(code) >> (32u - L);
                        for (uint32_t e = rev; e < 128u; e += 1u << L) S.cl_lut[e] = s << 4 | L;
                    }
                    // the hlit + hdist code lengths
                    const uint32_t total = hlit + hdist;
                    uint32_t i = 0, prev = 0, have = 0; uint64_t buf = 0;
                    while (i < total && !bad) {
                        if (have < 16u) { buf = peek64(S.win, p); have = 64; }
                        const uint32_t e = S.cl_lut[(uint32_t)buf & 127u];
                        const uint32_t L = e & 15u, s = e >> 4;
                        if (!L) { bad = 1; break; }
                        buf >>= L; have -= L; p += L;
                        if (s < 16u) { S.lens[i++] = (uint8_t)s; prev = s; continue; }
                        uint32_t rep, val = 0;
                        if (s == 16u) { if (i == 0) { bad = 1; break; } val = prev; rep = 3u + ((uint32_t)buf & 3u); buf >>= 2; have -= 2; p += 2; }
                        else if (s == 17u) { rep = 3u + ((uint32_t)buf & 7u); buf >>= 3; have -= 3; p += 3; prev = 0; }
                        else { rep = 11u + ((uint32_t)buf & 127u); buf >>= 7; have -= 7; p += 7; prev = 0; }
                        if (i + rep > total) { bad = 1; break; }
                        for (uint32_t k = 0; k < rep; ++k) S.lens[i++] = (uint8_t)val;
                    }
                    if (!bad && S.lens[256] == 0) bad = 1;                           // no end-of-block code
                }
                if (bad) atomicMax(&S.ctrl[C_ERR], (uint32_t)E_LENGTHS);
            } else if (btype == 1u) {
                if (!fixed_tables) {
                    for (int s = 0; s < 288; ++s) S.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
                    for (int s = 0; s < 32; ++s) S.lens[288 + s] = 5;
                }
                S.ctrl[C_HLIT] = 288; S.ctrl[C_HDIST] = 32;
            } else if (btype == 3u) atomicMax(&S.ctrl[C_ERR], (uint32_t)E_BLOCK_TYPE);
            const uint64_t end = base_byte * 8u + p;
            S.ctrl[C_HDR_END_LO] = (uint32_t)end; S.ctrl[C_HDR_END_HI] = (uint32_t)(end >> 32);
        }
        __syncthreads();
        const uint32_t btype = S.ctrl[C_BTYPE];
        const bool bfinal = S.ctrl[C_FINAL] != 0;
        pos = (uint64_t)S.ctrl[C_HDR_END_HI] << 32 | S.ctrl[C_HDR_END_LO];
        err = S.ctrl[C_ERR];
        if (!err && pos > src_bits) err = E_INPUT;
        PROF(P_HEADER);
        if (err) break;

        if (btype == 0u) {
            // -------------------------------------------------------------- stored block: LEN, ~LEN, bytes
            const uint64_t p = (pos + 7u) >> 3;
            if (p + 4u > it.src_len) { err = E_INPUT; break; }
            const uint32_t len = src[p] | (uint32_t)src[p + 1] << 8, nlen = src[p + 2] | (uint32_t)src[p + 3] << 8;
            if (len != (nlen ^ 0xFFFFu)) { err = E_STORED; break; }
            if (p + 4u + len > it.src_len) { err = E_INPUT; break; }
            for (uint32_t o = 0; o < len; o += (uint32_t)kNewMax) {
                const uint32_t n = len - o < (uint32_t)kNewMax ? len - o : (uint32_t)kNewMax;
                if (produced < it.dst_cap || o + (uint32_t)kHist >= len) {           // (beyond the caller's capacity only the window matters)
                    for (uint32_t i = t; i < n; i += kT) S.ring[(produced + i) & kRingMask] = src[p + 4u + o + i];
                    __syncthreads();
                    flush(produced, n);
                    __syncthreads();
                }
                produced = produced + n < produced ? 0xFFFFFFFFu : produced + n;
            }
            pos = (p + 4u + len) * 8u;
            PROF(P_STORED);
        } else {
            // -------------------------------------------------------------- Huffman block: tables, then chunks
            const int hlit = (int)S.ctrl[C_HLIT], hdist = (int)S.ctrl[C_HDIST];
            if (!(btype == 1u && fixed_tables)) {
                build_table<false>(S, 0, hlit);
                build_table<true>(S, btype == 1u ? 288 : hlit, hdist);
            }
            fixed_tables = btype == 1u;
            err = S.ctrl[C_ERR];
            PROF(P_TABLES);
            if (err) break;
            bool in_block = true;
            while (in_block && !err) {
                const uint64_t wbase = load_window(pos);
                PROF(P_WINDOW); PROF_COUNT(P_N_CHUNKS, 1);
                const uint32_t rel0 = (uint32_t)(pos - wbase * 8u);
                // speculative sweep, then sweeps from the predecessors' exits until the chain is consistent up to its end
                uint32_t my_start = t == 0 ? rel0 : (uint32_t)t * kSubBits;
                const uint32_t my_end = (uint32_t)(t + 1) * kSubBits;
                LaneResult r = lane_decode<0>(S, my_start, my_end, 0, 0);
                S.exit_bit[t] = r.exit; S.flags[t] = r.flags;
                if (t == 0) { S.ctrl[C_FIRST_BAD] = kT; S.ctrl[C_FIRST_STOP] = kT; }
                __syncthreads();
                PROF(P_SWEEP0);
                uint32_t first_stop = kT;
                for (int sweep = 0; sweep <= kT; ++sweep) {
                    PROF_COUNT(P_N_SWEEPS, 1);
                    const uint32_t prev = t ? S.exit_bit[t - 1] : rel0;
                    const bool moved = t > 0 && prev != my_start;
                    if (moved) atomicMin(&S.ctrl[C_FIRST_BAD], (uint32_t)t);
                    if (r.flags) atomicMin(&S.ctrl[C_FIRST_STOP], (uint32_t)t);
                    __syncthreads();
                    const uint32_t first_bad = S.ctrl[C_FIRST_BAD];
                    first_stop = S.ctrl[C_FIRST_STOP];
                    __syncthreads();
                    if (first_bad == (uint32_t)kT || first_stop < first_bad) break;
                    if (t == 0) { S.ctrl[C_FIRST_BAD] = kT; S.ctrl[C_FIRST_STOP] = kT; }
                    if (moved) { my_start = prev; r = lane_decode<0>(S, my_start, my_end, 0, 0); }
                    __syncthreads();
                    S.exit_bit[t] = r.exit; S.flags[t] = r.flags;
                    __syncthreads();
                }
                __syncthreads();
                PROF(P_SWEEPS);
                if (t == 0) S.ctrl[C_CUT] = kT;
                uint32_t nvalid = first_stop < (uint32_t)kT ? first_stop + 1u : (uint32_t)kT;      // lanes 0 .. nvalid - 1 form the chain
                // where everything goes; lanes whose output would overflow the ring wait for the next chunk
                uint32_t inc_o = (uint32_t)t < nvalid ? r.out : 0u;
                block_scan(S, inc_o);
                if ((uint32_t)t < nvalid && inc_o > (uint32_t)kNewMax) atomicMin(&S.ctrl[C_CUT], (uint32_t)t);
                __syncthreads();
                const uint32_t cut = S.ctrl[C_CUT];
                const bool was_cut = cut < nvalid;
                if (was_cut) nvalid = cut;                                            // >= 1: one lane alone always fits
                const uint32_t total = S.scan_a[nvalid - 1];
                const uint32_t last_exit = S.exit_bit[nvalid - 1], last_flags = was_cut ? 0u : S.flags[nvalid - 1];
                if (t == 0) S.ctrl[C_NLONG] = 0;
                __syncthreads();
                PROF(P_SCAN);
                const bool sink = produced >= it.dst_cap;                             // the caller's buffer is full: decode on, write nothing
                if ((uint32_t)t < nvalid) {
                    if (sink) lane_decode<2>(S, my_start, my_end, produced + (inc_o - r.out), inc_o - r.out);
                    else      lane_decode<1>(S, my_start, my_end, produced + (inc_o - r.out), inc_o - r.out);
                }
                __syncthreads();
                err = S.ctrl[C_ERR];
                if (!err && (last_flags & F_BAD)) err = E_CODE;
                if (!err && wbase * 8u + last_exit > src_bits) err = E_INPUT;
                PROF(P_WRITE);
                if (err) break;
                if (!sink) {
                    expand_long_matches(S);
                    __syncthreads();
                    { const int rounds = resolve_copies(S, produced, total); (void)rounds; PROF_COUNT(P_N_ROUNDS, rounds); }
                }
                __syncthreads();
                PROF(P_MATCH);
                if (!sink) flush(produced, total);
                __syncthreads();
                produced = produced + total < produced ? 0xFFFFFFFFu : produced + total;           // (saturating: only its size matters beyond 32 KiB)
                pos = wbase * 8u + last_exit;
                if (last_flags & F_EOB) in_block = false;
            }
        }
        if (bfinal) done = true;
    }
    PROF_FLUSH;
    if (t == 0) {
        out_len[blockIdx.x] = produced < it.dst_cap ? produced : it.dst_cap;
        status[blockIdx.x] = err;
    }
}

} // namespace

int inflate_launch(const gamut_hip_inflate_desc* descs, int count, uint32_t* out_len_dev, uint32_t* status_dev, hipStream_t stream)
{
    static_assert(sizeof(InfItem) == sizeof(gamut_hip_inflate_desc), "descriptor layout");
    std::vector<InfItem> items((size_t)count);                 // pageable: the upload below has read it when hipMemcpyAsync returns
    for (int i = 0; i < count; ++i) {
        if (!descs[i].src || (!descs[i].dst && descs[i].dst_cap)) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "inflate: stream %d: null pointer", i);
        items[(size_t)i] = InfItem{ descs[i].src, descs[i].dst, descs[i].src_len, descs[i].dst_cap };
    }
    // per call: the attribute belongs to the current device's copy of the kernel, and a process may drive several devices
    if (hipFuncSetAttribute((const void*)k_inflate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Shared)) != hipSuccess) return set_error(GAMUT_HIP_ERR_HIP, "inflate: %zu bytes of LDS are not available", sizeof(Shared));
    // the descriptor table in HBM: one buffer per (thread, stream) -- calls on one stream are ordered, calls on different
    // streams never share it; growing it waits for its own stream only
    struct StreamTable { int device; hipStream_t stream; void* p; size_t cap; };
    static thread_local std::vector<StreamTable> tables;
    StreamTable* e = nullptr;
    const int device = current_device();                    // (the null stream is one handle for every device)
    for (StreamTable& c : tables) if (c.stream == stream && c.device == device) { e = &c; break; }
    if (!e) { tables.push_back(StreamTable{ device, stream, nullptr, 0 }); e = &tables.back(); }
    const size_t bytes = items.size() * sizeof(InfItem);
    if (bytes > e->cap) {
        if (e->p) { (void)hipStreamSynchronize(stream); (void)hipFree(e->p); e->p = nullptr; e->cap = 0; }
        if (hipMalloc(&e->p, bytes * 2 + 4096) != hipSuccess) { (void)hipGetLastError(); e->p = nullptr; return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "inflate: descriptor table allocation failed"); }
        e->cap = bytes * 2 + 4096;
    }
    GAMUT_HIP_CHECK(hipMemcpyAsync(e->p, items.data(), bytes, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_inflate, dim3((unsigned)count), dim3(kT), sizeof(Shared), stream, (const InfItem*)e->p, count, out_len_dev, status_dev);
    return launch_status("inflate");
}

} // namespace gamut

using namespace gamut;

#if INFLATE_PROFILE
extern "C" int gamut_hip_inflate_profile(unsigned long long* out, int reset)      // measurement builds only
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_inflate_prof), sizeof(unsigned long long) * P_N) != hipSuccess) return 1;
    if (reset) { unsigned long long z[P_N] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_inflate_prof), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

extern "C" int gamut_hip_inflate_batch_device(const gamut_hip_inflate_desc* descs, int count, uint32_t* out_len_dev, uint32_t* status_dev, void* stream)
{
    clear_error();
    if (count < 0 || (count > 0 && (!descs || !out_len_dev || !status_dev))) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "inflate_batch_device: bad arguments");
    if (count == 0) return GAMUT_HIP_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_error(GAMUT_HIP_ERR_NO_DEVICE, "no HIP device available (libgamut_hip has no CPU fallback)");
    try { return inflate_launch(descs, count, out_len_dev, status_dev, pick_stream(stream)); }
    catch (...) { return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "inflate_batch_device: out of host memory"); }
}
