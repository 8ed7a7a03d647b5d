// jpeg_math.hpp -- device arithmetic of JPEG block reconstruction, bit-exact with the
// reference's jpgd port (source/gamut/codecs/jpegload.d).
//
//   1-D butterfly            :178-202 / :240-265   (Row!N / Col!N, dense form)
//   pass 1 descale           :204-211              DESCALE(x, CONST_BITS-PASS1_BITS)
//   pass 2 descale + clamp   :267-289              DESCALE_ZEROSHIFT(x, 18), CLAMP
//   Col!1 shortcut           :222-232              only reachable for max_zag <= 2
//   chroma upsample maps     :914-1072             F(x) = (int)(x*1024+0.5f), D(i) = (i+512)>>10
//   YCbCr -> RGB             :2080-2094, :2769-2794
//
// All arithmetic is 32-bit wrap-around.  Multiplies are written with __mul24 /
// mad24 where both operands provably fit 24 signed bits (see DESIGN.md "integer
// ranges"): v_mul_i32_i24 returns the low 32 bits of the exact product, identical
// to a wrapping 32-bit multiply for such operands, and is full rate on gfx950
// where v_mul_lo_u32 is quarter rate.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace gamut {
namespace jpg {

typedef int32_t i32;
typedef uint32_t u32;

constexpr int FIX_0_298631336 = 2446,  FIX_0_390180644 = 3196,  FIX_0_541196100 = 4433,
              FIX_0_765366865 = 6270,  FIX_0_899976223 = 7373,  FIX_1_175875602 = 9633,
              FIX_1_501321110 = 12299, FIX_1_847759065 = 15137, FIX_1_961570560 = 16069,
              FIX_2_053119869 = 16819, FIX_2_562915447 = 20995, FIX_3_072711026 = 25172;

// wrap-around ops on unsigned to stay defined in C++
__device__ __forceinline__ i32 wadd(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
__device__ __forceinline__ i32 wsub(i32 a, i32 b) { return (i32)((u32)a - (u32)b); }
// a*c for |a| < 2^23, |c| < 2^23 : low 32 bits of the exact product
__device__ __forceinline__ i32 mul24(i32 a, i32 c) { return __mul24(a, c); }
// a*c + b (wrapping)
__device__ __forceinline__ i32 mad24(i32 a, i32 c, i32 b) { return (i32)((u32)__mul24(a, c) + (u32)b); }

// Un-descaled 1-D IDCT butterfly; `round` is added to every output (it rides on tmp0/tmp1).
// NZ = number of leading non-zero inputs known at compile time (8 = dense, 4 = idct_4x4 rows/cols).
template <int NZ>
__device__ __forceinline__ void butterfly(const i32 (&x)[8], i32 (&y)[8], i32 round)
{
    const i32 x0 = x[0], x1 = NZ > 1 ? x[1] : 0, x2 = NZ > 2 ? x[2] : 0, x3 = NZ > 3 ? x[3] : 0;
    const i32 x4 = NZ > 4 ? x[4] : 0, x5 = NZ > 5 ? x[5] : 0, x6 = NZ > 6 ? x[6] : 0, x7 = NZ > 7 ? x[7] : 0;

    const i32 z1   = mul24(wadd(x2, x6), FIX_0_541196100);
    const i32 tmp2 = mad24(x6, -FIX_1_847759065, z1);
    const i32 tmp3 = mad24(x2, FIX_0_765366865, z1);
    const i32 tmp0 = wadd((i32)((u32)wadd(x0, x4) << 13), round);
    const i32 tmp1 = wadd((i32)((u32)wsub(x0, x4) << 13), round);
    const i32 tmp10 = wadd(tmp0, tmp3), tmp13 = wsub(tmp0, tmp3), tmp11 = wadd(tmp1, tmp2), tmp12 = wsub(tmp1, tmp2);

    const i32 bz1 = wadd(x7, x1), bz2 = wadd(x5, x3), bz3 = wadd(x7, x3), bz4 = wadd(x5, x1);
    const i32 bz5 = mul24(wadd(bz3, bz4), FIX_1_175875602);
    const i32 az1 = mul24(bz1, -FIX_0_899976223);
    const i32 az2 = mul24(bz2, -FIX_2_562915447);
    const i32 az3 = mad24(bz3, -FIX_1_961570560, bz5);
    const i32 az4 = mad24(bz4, -FIX_0_390180644, bz5);
    const i32 btmp0 = wadd(mad24(x7, FIX_0_298631336, az1), az3);
    const i32 btmp1 = wadd(mad24(x5, FIX_2_053119869, az2), az4);
    const i32 btmp2 = wadd(mad24(x3, FIX_3_072711026, az2), az3);
    const i32 btmp3 = wadd(mad24(x1, FIX_1_501321110, az1), az4);

    y[0] = wadd(tmp10, btmp3); y[7] = wsub(tmp10, btmp3);
    y[1] = wadd(tmp11, btmp2); y[6] = wsub(tmp11, btmp2);
    y[2] = wadd(tmp12, btmp1); y[5] = wsub(tmp12, btmp1);
    y[3] = wadd(tmp13, btmp0); y[4] = wsub(tmp13, btmp0);
}

// pass 1 on one row of coefficients -> 8 ints (DESCALE by 11)
template <int NZ>
__device__ __forceinline__ void row_pass(const i32 (&x)[8], i32 (&t)[8])
{
    i32 y[8];
    butterfly<NZ>(x, y, 1 << 10);
    #pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = y[i] >> 11;
}

__device__ __forceinline__ i32 clamp255(i32 v) { return min(max(v, 0), 255); }

// pass 2 on one column of pass-1 values -> 8 samples 0..255 (DESCALE_ZEROSHIFT by 18, CLAMP)
template <int NZ>
__device__ __forceinline__ void col_pass(const i32 (&t)[8], i32 (&s)[8])
{
    i32 y[8];
    butterfly<NZ>(t, y, (128 << 18) + (1 << 17));
    #pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = clamp255(y[i] >> 18);
}
// Col!(1) (:222-232): reachable only when m_mcu_block_max_zag <= 2
__device__ __forceinline__ i32 col1_sample(i32 t0) { return clamp255(wadd(t0, (128 << 5) + 16) >> 5); }

// unpack one 16-byte row of 8 int16 coefficients
__device__ __forceinline__ void unpack_row(const uint4& v, i32 (&x)[8])
{
    x[0] = (i32)(short)(v.x & 0xFFFF); x[1] = (i32)v.x >> 16;
    x[2] = (i32)(short)(v.y & 0xFFFF); x[3] = (i32)v.y >> 16;
    x[4] = (i32)(short)(v.z & 0xFFFF); x[5] = (i32)v.z >> 16;
    x[6] = (i32)(short)(v.w & 0xFFFF); x[7] = (i32)v.w >> 16;
}

// ---- frequency-domain 2x upsample -------------------------------------------
// F!(x) of :911 for the sixteen constants that appear in P_Q / R_S
constexpr int F(float x) { return (int)(x * 1024 + 0.5f); }
constexpr int E1a = F(0.415735f), E1b = F(0.791065f), E1c = F(-0.352443f), E1d = F(0.277785f);
constexpr int E3a = F(0.022887f), E3b = F(-0.097545f), E3c = F(0.490393f), E3d = F(0.865723f);
constexpr int O0a = F(0.906127f), O0b = F(-0.318190f), O0c = F(0.212608f), O0d = F(-0.180240f);
constexpr int O2a = F(-0.074658f), O2b = F(0.513280f), O2c = F(0.768178f), O2d = F(-0.375330f);
static_assert(E1a == 426 && E1b == 810 && E1c == -360 && E1d == 284, "F() constants");
static_assert(E3a == 23 && E3b == -99 && E3c == 502 && E3d == 887, "F() constants");
static_assert(O0a == 928 && O0b == -325 && O0c == 218 && O0d == -184, "F() constants");
static_assert(O2a == -75 && O2b == 526 && O2c == 787 && O2d == -383, "F() constants");

__device__ __forceinline__ i32 D4(i32 a, i32 u1, i32 b, i32 u3, i32 c, i32 u5, i32 d, i32 u7)
{   // D(a*u1 + b*u3 + c*u5 + d*u7) = (sum + 512) >> 10
    return mad24(d, u7, mad24(c, u5, mad24(b, u3, mad24(a, u1, 512)))) >> 10;
}
// "E" map: (u0, D(426 u1 + 810 u3 - 360 u5 + 284 u7), u4, D(23 u1 - 99 u3 + 502 u5 + 887 u7))
__device__ __forceinline__ void map_E(const i32 (&u)[8], i32 (&e)[4])
{
    e[0] = u[0]; e[1] = D4(E1a, u[1], E1b, u[3], E1c, u[5], E1d, u[7]);
    e[2] = u[4]; e[3] = D4(E3a, u[1], E3b, u[3], E3c, u[5], E3d, u[7]);
}
// "O" map: (D(928 u1 - 325 u3 + 218 u5 - 184 u7), u2, D(-75 u1 + 526 u3 + 787 u5 - 383 u7), u6)
__device__ __forceinline__ void map_O(const i32 (&u)[8], i32 (&o)[4])
{
    o[0] = D4(O0a, u[1], O0b, u[3], O0c, u[5], O0d, u[7]); o[1] = u[2];
    o[2] = D4(O2a, u[1], O2b, u[3], O2c, u[5], O2d, u[7]); o[3] = u[6];
}

// ---- colour (:2080-2094 / expanded_convert :2769-2794) -----------------------
constexpr int CFIX(float x) { return (int)(x * 65536.0f + 0.5f); }
constexpr int C_CRR = CFIX(1.40200f), C_CBB = CFIX(1.77200f), C_CRG = -CFIX(0.71414f), C_CBG = -CFIX(0.34414f);
static_assert(C_CRR == 91881 && C_CBB == 116130 && C_CRG == -46802 && C_CBG == -22554, "FIX() constants");

// The two addends of the single multiply-adds below, pinned in vector registers: a multiply-add (VOP3) can name one scalar
// constant only, and left to itself the compiler re-materialises the other one with a v_mov in front of every pixel.
struct ColourConsts { i32 kr, kb; };
__device__ __forceinline__ ColourConsts colour_consts()
{
    ColourConsts c;
    asm volatile("v_mov_b32 %0, %1" : "=v"(c.kr) : "s"(32768 - 128 * C_CRR));
    asm volatile("v_mov_b32 %0, %1" : "=v"(c.kb) : "s"(32768 - 128 * C_CBB));
    return c;
}
// returns packed RGBA8 (A = 255), little-endian byte order R,G,B,A
__device__ __forceinline__ u32 ycc_to_rgba(i32 y, i32 cb, i32 cr, i32 kr = 32768 - 128 * C_CRR, i32 kb = 32768 - 128 * C_CBB)
{
    // (c - 128) * K + 32768 == c * K + (32768 - 128 * K): the level shift rides in the addend
    const i32 r = y + (mad24(cr, C_CRR, kr) >> 16);
    const i32 g = y + (mad24(cr, C_CRG, mad24(cb, C_CBG, 32768 - 128 * C_CRG - 128 * C_CBG)) >> 16);
    const i32 b = y + (mad24(cb, C_CBB, kb) >> 16);
    // clamp + pack: r, g, b are within +-2^10, so they can be saturated as 16-bit lanes (v_sat_pk_u8_i16 does two at once)
    // and gathered with byte permutes: 4 instructions instead of 3 clamps + 3 shift/ors
    const u32 rg = __builtin_amdgcn_perm((u32)g, (u32)r, 0x05040100u);     // r.lo16 | g.lo16 << 16
    u32 rg8, b8;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(rg8) : "v"(rg));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b8) : "v"(b));                     // byte 0 = sat(b); byte 1 unused
    return __builtin_amdgcn_perm(b8, rg8, 0x0d040100u);                    // R, G, B, 0xFF
}
// RGB -> grey of decompress_jpeg_image_from_stream (:3786-3792)
__device__ __forceinline__ u32 rgb_to_luma(u32 rgba)
{
    const u32 r = rgba & 0xFF, g = (rgba >> 8) & 0xFF, b = (rgba >> 16) & 0xFF;
    return (r * 19595u + g * 38470u + b * 7471u + 32768u) >> 16;
}

} // namespace jpg
} // namespace gamut
