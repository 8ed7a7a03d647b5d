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
// the same as exactly one v_mad_i32_i24, for a run-time multiplier of +-1: the optimiser otherwise "strength-reduces" such a
// product into sign-extend / negate / select chains (six instructions, one of them a 64-bit multiply-add)
__device__ __forceinline__ i32 mad24_1(i32 a, i32 c, i32 b)
{
    i32 d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(c), "v"(b));
    return d;
}

// a * K + b as exactly one v_mad_i32_i24 with the constant K in a scalar register and the addend in a vector register.  Left to
// itself the compiler rewrites chains of multiply-adds by literals into v_mul (literal) + v_mul (literal) + v_add3, or into
// difference chains (tmp13 = tmp10 - 2 K t2 ...): one instruction more per chain, 10 per wave in the colour stage alone.
__device__ __forceinline__ i32 mad24_k(i32 a, i32 k, i32 b)
{
    i32 d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(b));
    return d;
}

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

// ---- packed int16 forms (v_dot2_i32_i16) ---------------------------------------
// The 1-D butterfly (:178-202) is a LINEAR map with integer coefficients, and the reference evaluates it in wrap-around
// 32-bit arithmetic: any regrouping of its products is bit-identical mod 2^32.  Written out per input,
//   tmp0/1 = 8192 x0 +- 8192 x4                 tmp3 = (c541 + c765) x2 + c541 x6        tmp2 = c541 x2 + (c541 - c1847) x6
//   btmp0  = (c1175-c899) x1 + (c1175-c1961) x3 + c1175 x5 + (c298-c899-c1961+c1175) x7      (and the three siblings)
// every coefficient still fits int16, so when the INPUTS are int16 too (the de-quantised coefficients of pass 1; the
// cast(short) outputs of the upsample in idct_4x4's pass 1) two products and their sum are one v_dot2_i32_i16 on a packed
// pair: exact 31-bit products, 32-bit wrapping accumulate.  Measured on gfx950 (tools/microbench.hip) a dot2 issues in the
// same 4.3 cycles as one v_mad_i32_i24.  Pass 2 cannot use it: its inputs are the >>11 values (up to 2^20 for wild input).
typedef short short2v __attribute__((ext_vector_type(2)));
constexpr u32 pk16(int lo, int hi) { return ((u32)lo & 0xFFFFu) | (((u32)hi & 0xFFFFu) << 16); }
// Written as the three-operand VOP3P form by hand: left to itself the compiler picks the two-operand accumulate form
// (v_dot2c, the only one that takes a literal) and pays a v_mov per accumulator it has to preserve or initialise.
// The constant pair sits in a scalar register (VOP3P may name one), a zero accumulator is the inline constant.
__device__ __forceinline__ i32 dot2(u32 pair, u32 cpair, i32 acc)
{
    i32 d;
    if (__builtin_constant_p(cpair)) {
        if (__builtin_constant_p(acc) && acc == 0) asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(pair), "s"(cpair));
        else asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(pair), "s"(cpair), "v"(acc));
    } else {
        asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(pair), "v"(cpair), "v"(acc));
    }
    return d;
}
constexpr int K_E2 = FIX_0_541196100 + FIX_0_765366865;      // tmp3: x2
constexpr int K_E6 = FIX_0_541196100 - FIX_1_847759065;      // tmp2: x6
// odd part, per output (x1, x3, x5, x7)
constexpr int K_B0[4] = { FIX_1_175875602 - FIX_0_899976223, FIX_1_175875602 - FIX_1_961570560, FIX_1_175875602,
                          FIX_0_298631336 - FIX_0_899976223 - FIX_1_961570560 + FIX_1_175875602 };
constexpr int K_B1[4] = { FIX_1_175875602 - FIX_0_390180644, FIX_1_175875602 - FIX_2_562915447,
                          FIX_2_053119869 - FIX_2_562915447 - FIX_0_390180644 + FIX_1_175875602, FIX_1_175875602 };
constexpr int K_B2[4] = { FIX_1_175875602, FIX_3_072711026 - FIX_2_562915447 - FIX_1_961570560 + FIX_1_175875602,
                          FIX_1_175875602 - FIX_2_562915447, FIX_1_175875602 - FIX_1_961570560 };
constexpr int K_B3[4] = { FIX_1_501321110 - FIX_0_899976223 - FIX_0_390180644 + FIX_1_175875602, FIX_1_175875602,
                          FIX_1_175875602 - FIX_0_390180644, FIX_1_175875602 - FIX_0_899976223 };
static_assert(K_E2 == 10703 && K_E6 == -10704, "direct-form even constants");
static_assert(K_B0[0] == 2260 && K_B0[1] == -6436 && K_B0[2] == 9633 && K_B0[3] == -11363, "direct-form odd constants");
static_assert(K_B1[0] == 6437 && K_B1[1] == -11362 && K_B1[2] == 2261 && K_B1[3] == 9633, "direct-form odd constants");
static_assert(K_B2[0] == 9633 && K_B2[1] == -2259 && K_B2[2] == -11362 && K_B2[3] == -6436, "direct-form odd constants");
static_assert(K_B3[0] == 11363 && K_B3[1] == 9633 && K_B3[2] == 6437 && K_B3[3] == 2260, "direct-form odd constants");

// pass 1 on one 16-byte row of 8 int16 coefficients (x0|x1, x2|x3, x4|x5, x6|x7) -> 8 ints, == row_pass<8>(unpack_row(v))
__device__ __forceinline__ void row_pass_packed(const uint4& v, i32 (&t)[8])
{
    const u32 p04 = __builtin_amdgcn_perm(v.z, v.x, 0x05040100u);       // x0 | x4 << 16
    const u32 p26 = __builtin_amdgcn_perm(v.w, v.y, 0x05040100u);
    const u32 p13 = __builtin_amdgcn_perm(v.y, v.x, 0x07060302u);
    const u32 p57 = __builtin_amdgcn_perm(v.w, v.z, 0x07060302u);
    constexpr i32 R = 1 << 10;
    const i32 tmp0 = dot2(p04, pk16(8192, 8192), R), tmp1 = dot2(p04, pk16(8192, -8192), R);
    const i32 tmp10 = dot2(p26, pk16(K_E2, FIX_0_541196100), tmp0), tmp13 = dot2(p26, pk16(-K_E2, -FIX_0_541196100), tmp0);
    const i32 tmp11 = dot2(p26, pk16(FIX_0_541196100, K_E6), tmp1), tmp12 = dot2(p26, pk16(-FIX_0_541196100, -K_E6), tmp1);
    const i32 b0 = dot2(p13, pk16(K_B0[0], K_B0[1]), dot2(p57, pk16(K_B0[2], K_B0[3]), 0));
    const i32 b1 = dot2(p13, pk16(K_B1[0], K_B1[1]), dot2(p57, pk16(K_B1[2], K_B1[3]), 0));
    const i32 b2 = dot2(p13, pk16(K_B2[0], K_B2[1]), dot2(p57, pk16(K_B2[2], K_B2[3]), 0));
    const i32 b3 = dot2(p13, pk16(K_B3[0], K_B3[1]), dot2(p57, pk16(K_B3[2], K_B3[3]), 0));
    t[0] = wadd(tmp10, b3) >> 11; t[7] = wsub(tmp10, b3) >> 11;
    t[1] = wadd(tmp11, b2) >> 11; t[6] = wsub(tmp11, b2) >> 11;
    t[2] = wadd(tmp12, b1) >> 11; t[5] = wsub(tmp12, b1) >> 11;
    t[3] = wadd(tmp13, b0) >> 11; t[4] = wsub(tmp13, b0) >> 11;
}
// idct_4x4's pass 1 (Row!4, :378-397) on cast(short)(v0..v3): the byte permutes take the low halves, which IS the cast
__device__ __forceinline__ void row_pass4_pairs(u32 p02, u32 p13, i32 (&t)[8]);
__device__ __forceinline__ void row_pass4_packed(i32 v0, i32 v1, i32 v2, i32 v3, i32 (&t)[8])
{
    row_pass4_pairs(__builtin_amdgcn_perm((u32)v2, (u32)v0, 0x05040100u), __builtin_amdgcn_perm((u32)v3, (u32)v1, 0x05040100u), t);
}
// the same on ready-made pairs p02 = x0 | x2 << 16, p13 = x1 | x3 << 16
__device__ __forceinline__ void row_pass4_pairs(u32 p02, u32 p13, i32 (&t)[8])
{
    constexpr i32 R = 1 << 10;
    const i32 tmp10 = dot2(p02, pk16(8192, K_E2), R), tmp13 = dot2(p02, pk16(8192, -K_E2), R);
    const i32 tmp11 = dot2(p02, pk16(8192, FIX_0_541196100), R), tmp12 = dot2(p02, pk16(8192, -FIX_0_541196100), R);
    t[0] = dot2(p13, pk16(K_B3[0], K_B3[1]), tmp10) >> 11; t[7] = dot2(p13, pk16(-K_B3[0], -K_B3[1]), tmp10) >> 11;
    t[1] = dot2(p13, pk16(K_B2[0], K_B2[1]), tmp11) >> 11; t[6] = dot2(p13, pk16(-K_B2[0], -K_B2[1]), tmp11) >> 11;
    t[2] = dot2(p13, pk16(K_B1[0], K_B1[1]), tmp12) >> 11; t[5] = dot2(p13, pk16(-K_B1[0], -K_B1[1]), tmp12) >> 11;
    t[3] = dot2(p13, pk16(K_B0[0], K_B0[1]), tmp13) >> 11; t[4] = dot2(p13, pk16(-K_B0[0], -K_B0[1]), tmp13) >> 11;
}
// pass 2 of idct_4x4 (Col!4) on 32-bit inputs t0..t3 (t4..t7 = 0), direct form: 13 multiply-adds + 8 adds instead of
// the butterfly's 10 + 17 (an add costs 2.5 issue cycles, a multiply-add 4.3)
__device__ __forceinline__ void col_pass4_direct(i32 t0, i32 t1, i32 t2, i32 t3, i32 (&s)[8], i32 round = (128 << 18) + (1 << 17))
{
    // `round` from a register (col_round()): one v_lshl_add_u32 and four multiply-adds; with a literal the compiler chains
    // shift, multiply, add3 and three multiply-adds (a VOP3 instruction reads one scalar / literal operand at most)
    const i32 tmp0 = wadd((i32)((u32)t0 << 13), round);
    const i32 tmp10 = mad24_k(t2, K_E2, tmp0), tmp13 = mad24_k(t2, -K_E2, tmp0);
    const i32 tmp11 = mad24_k(t2, FIX_0_541196100, tmp0), tmp12 = mad24_k(t2, -FIX_0_541196100, tmp0);
    const i32 b0 = mad24_k(t3, K_B0[1], mul24(t1, K_B0[0])), b1 = mad24_k(t3, K_B1[1], mul24(t1, K_B1[0]));
    const i32 b2 = mad24_k(t3, K_B2[1], mul24(t1, K_B2[0])), b3 = mad24_k(t3, K_B3[1], mul24(t1, K_B3[0]));
    s[0] = clamp255(wadd(tmp10, b3) >> 18); s[7] = clamp255(wsub(tmp10, b3) >> 18);
    s[1] = clamp255(wadd(tmp11, b2) >> 18); s[6] = clamp255(wsub(tmp11, b2) >> 18);
    s[2] = clamp255(wadd(tmp12, b1) >> 18); s[5] = clamp255(wsub(tmp12, b1) >> 18);
    s[3] = clamp255(wadd(tmp13, b0) >> 18); s[4] = clamp255(wsub(tmp13, b0) >> 18);
}

__device__ __forceinline__ i32 col_round()
{
    i32 r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"((128 << 18) + (1 << 17)));
    return r;
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
struct ColourConsts { i32 kr, kg, kb; };
__device__ __forceinline__ ColourConsts colour_consts()
{
    ColourConsts c;
    asm volatile("v_mov_b32 %0, %1" : "=v"(c.kr) : "s"(32768 - 128 * C_CRR));
    asm volatile("v_mov_b32 %0, %1" : "=v"(c.kg) : "s"(32768 - 128 * C_CRG - 128 * C_CBG));
    asm volatile("v_mov_b32 %0, %1" : "=v"(c.kb) : "s"(32768 - 128 * C_CBB));
    return c;
}
// returns packed RGBA8 (A = 255), little-endian byte order R,G,B,A
__device__ __forceinline__ u32 ycc_to_rgba(i32 y, i32 cb, i32 cr, i32 kr = 32768 - 128 * C_CRR, i32 kb = 32768 - 128 * C_CBB,
                                           i32 kg = 32768 - 128 * C_CRG - 128 * C_CBG)
{
    // (c - 128) * K + 32768 == c * K + (32768 - 128 * K): the level shift rides in the addend
    const i32 r = y + (mad24_k(cr, C_CRR, kr) >> 16);
    const i32 g = y + (mad24_k(cr, C_CRG, mad24_k(cb, C_CBG, kg)) >> 16);           // with kg in a register: two multiply-adds (a literal addend made it mul + mul + add3)
    const i32 b = y + (mad24_k(cb, C_CBB, kb) >> 16);
    // clamp + pack: r, g, b are within +-2^10, so they can be saturated as 16-bit lanes (v_sat_pk_u8_i16 does two at once)
    // and gathered with byte permutes: 4 instructions instead of 3 clamps + 3 shift/ors
    const u32 rg = __builtin_amdgcn_perm((u32)g, (u32)r, 0x05040100u);     // r.lo16 | g.lo16 << 16
    u32 rg8, b8;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(rg8) : "v"(rg));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b8) : "v"(b));                     // byte 0 = sat(b); byte 1 unused
    return __builtin_amdgcn_perm(b8, rg8, 0x0d040100u);                    // R, G, B, 0xFF
}
// (Tried in round 3 and dropped: chroma samples centred on zero for free -- clamp the column pass's value WITHOUT its 128 << 18
// level shift to [-128, 127] -- with Y riding in the multiply-adds' addend as (Y << 16) + 32768, 9 instructions per pixel instead
// of 11 and 2.66 -> 2.51 ms per 1024 x 1080p.  It is not the reference's arithmetic on every input: the level shift takes part in
// the 32-bit wrap-around, and a column sum in [2^31 - 2^25, 2^31) -- reachable with full-range int16 coefficients, which a
// decodable file can contain -- clamps to 0 there and to 255 without it.  test_random_coefficients[wild-4] caught it.)
// RGB -> grey of decompress_jpeg_image_from_stream (:3786-3792)
__device__ __forceinline__ u32 rgb_to_luma(u32 rgba)
{
    const u32 r = rgba & 0xFF, g = (rgba >> 8) & 0xFF, b = (rgba >> 16) & 0xFF;
    return (r * 19595u + g * 38470u + b * 7471u + 32768u) >> 16;
}

} // namespace jpg
} // namespace gamut
