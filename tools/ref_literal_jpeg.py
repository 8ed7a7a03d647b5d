#!/usr/bin/env python3
"""ref_literal_jpeg.py -- a SECOND, structurally independent restatement of the reference's H2V2 JPEG block path, written
literally from the D text of /root/reference/source/gamut/codecs/jpegload.d (TEST INFRASTRUCTURE ONLY, like oracle/).

Why it exists: the reference cannot be built here (D, no compiler) and its tests hold no JPEG pixel vectors, so
oracle/oracle_jpeg.c -- the checker every HIP kernel is compared with -- was pinned for 4:2:0 only by a float model with a
6-LSB budget.  oracle_jpeg.c restates the upsample COMPACTLY (two 4-output maps applied twice, loops, a dense butterfly).
This file restates the same lines the LONG way, statement for statement as the D source has them:

  * Row!(N).idct / Col!(N).idct           :156-214 / :218-292   with ACCESS_COL / ACCESS_ROW substituting literal zeros
  * idct (DC shortcut, row/col tables)    :295-376              s_idct_row_table / s_idct_col_table copied as data
  * idct_4x4                              :378-397
  * DCT_Upsample.Matrix44                 :829-903              at(r, c), +, -, +=, -=, add_and_store, sub_and_store
  * DCT_Upsample.P_Q!(R,C) / R_S!(R,C)    :914-1072             every X0ij / X1ij / P.at / Q.at / R.at / S.at statement,
                                                                 produced MECHANICALLY from the D lines by tools/make_ref_literal.py
                                                                 (regex: `mixin(AT!(c, r))` -> AT(c, r), `F!(xf)` -> F(x), `P.at(r, c)` -> P.v[r][c])
  * s_max_rc + transform_mcu_expand       :2132-2255            the 15-case switch, a/b/c/d, the four store + idct_4x4 calls
  * create_look_ups, expanded_convert     :2080-2094, :2731-2823  the SSE sequence (mullo, srai, packs_epi32, packus_epi16)

tests/test_oracle_pinning.py requires BIT-EQUALITY between this file and oracle_jpeg.c on >= 10^5 random blocks (natural,
dense and wild int16), for every s_max_rc entry, with zeros and with garbage beyond max_zag -- two independently written
readings of the same source that agree everywhere.  What remains unpinnable: that both readings match what an actual D
compiler emits for this source (no D toolchain in the image).

All arithmetic is vectorised over N blocks with numpy int32 (wrap-around like D's int; `>>` is arithmetic).
"""
import numpy as np

np.seterr(over="ignore")
I32 = np.int32

# ---------------------------------------------------------------------------------------------- D semantics (hand-written glue)
def i32(x):
    return np.asarray(x).astype(I32)


def to_int(x):                     # a value assigned to a D `int` variable
    return np.asarray(x).astype(I32)


def to_short(x):                   # cast(jpgd_block_t): the low 16 bits
    return i32(x).astype(np.int16)


def FIX(x):                        # enum FIX(float x) = (cast(int)((x) * (1L<<SCALEBITS) + 0.5f)); float arithmetic, truncation toward zero
    return I32(int(np.float32(np.float32(x) * np.float32(1 << 16)) + np.float32(0.5)))


NUM_ROWS = NUM_COLS = 4            # DCT_Upsample.Matrix44: enum { NUM_ROWS = 4, NUM_COLS = 4 } (:832)


# __m128i / __m128 of intel-intrinsics as N registers at once: (N, 16) bytes; `+`, `-`, `+=` are D's int4 vector operators.
# (What an intrinsic does is Intel's definition, not the reference's arithmetic; WHICH intrinsics run on WHAT, in which order,
# is the generated text below.)
class M128:
    def __init__(self, b):
        self.b = np.ascontiguousarray(b, np.uint8)

    def i(self):
        return self.b.view(I32)

    @staticmethod
    def from_i32(v):
        return M128(np.ascontiguousarray(v, I32).view(np.uint8))

    def __add__(self, o):
        return M128.from_i32(self.i() + o.i())

    def __sub__(self, o):
        return M128.from_i32(self.i() - o.i())

    __iadd__ = __add__              # a NEW register: `A = mm_crr` followed by `mm_crr += x` must not change A


class BytesAt:
    """N buffers at once, with a D pointer into them: p[k] / &p[k] address byte base + k of every buffer; `p += n` moves it"""
    def __init__(self, arr, base=0):
        self.arr, self.base = arr, base

    def __iadd__(self, n):
        return BytesAt(self.arr, self.base + n)


def _lanes(ref):
    return ref.arr.shape[0]


_N = [1]                           # lanes of the registers made by _mm_set1_epi32 / _mm_setzero_si128 (set by the loaders)


def _mm_loadu_si32(p, k):          # 4 bytes into the low dword, the rest zero
    _N[0] = p.arr.shape[0]
    r = np.zeros((_N[0], 16), np.uint8)
    r[:, :4] = p.arr[:, p.base + k: p.base + k + 4]
    return M128(r)


def _mm_setzero_si128():
    return M128(np.zeros((_N[0], 16), np.uint8))


def _mm_set1_epi32(v):
    return M128.from_i32(np.full((_N[0], 4), int(v), I32))


def _mm_unpacklo_epi8(a, b):       # a0 b0 a1 b1 ... a7 b7
    r = np.empty_like(a.b); r[:, 0::2] = a.b[:, :8]; r[:, 1::2] = b.b[:, :8]
    return M128(r)


def _mm_unpacklo_epi16(a, b):
    a16, b16 = a.b.view(np.uint16), b.b.view(np.uint16)
    r = np.empty_like(a16); r[:, 0::2] = a16[:, :4]; r[:, 1::2] = b16[:, :4]
    return M128(r.view(np.uint8))


def _mm_mullo_epi32(a, b):
    return M128.from_i32(a.i() * b.i())


def _mm_srai_epi32(a, n):
    return M128.from_i32(a.i() >> I32(n))


def _MM_TRANSPOSE4_PS(A, B, C, D):
    m = np.stack([A.i(), B.i(), C.i(), D.i()], axis=1)        # (N, 4 registers, 4 lanes)
    t = m.transpose(0, 2, 1)
    return tuple(M128.from_i32(np.ascontiguousarray(t[:, k])) for k in range(4))


def _mm_packs_epi32(a, b):         # signed saturation int32 -> int16, a's four then b's four
    return M128(np.clip(np.concatenate([a.i(), b.i()], axis=1), -32768, 32767).astype(np.int16).view(np.uint8))


def _mm_packus_epi16(a, b):        # unsigned saturation int16 -> uint8
    return M128(np.clip(np.concatenate([a.b.view(np.int16), b.b.view(np.int16)], axis=1), 0, 255).astype(np.uint8))


def _mm_storeu_si128(p, k, v):
    p.arr[:, p.base + k: p.base + k + 16] = v.b


# BEGIN GENERATED MISC (tools/make_ref_literal.py from jpegload.d)
# jpegload.d:120-135
CONST_BITS = 13
PASS1_BITS = 2
SCALEDONE = 1
FIX_0_298631336 = 2446
FIX_0_390180644 = 3196
FIX_0_541196100 = 4433
FIX_0_765366865 = 6270
FIX_0_899976223 = 7373
FIX_1_175875602 = 9633
FIX_1_501321110 = 12299
FIX_1_847759065 = 15137
FIX_1_961570560 = 16069
FIX_2_053119869 = 16819
FIX_2_562915447 = 20995
FIX_3_072711026 = 25172
g_ZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]          # jpegload.d:106-106
s_idct_row_table = [1, 0, 0, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 2, 1, 0, 0, 0, 0, 0, 0, 2, 1, 1, 0, 0, 0, 0, 0, 2, 2, 1, 0, 0, 0, 0, 0, 3, 2, 1, 0, 0, 0, 0, 0, 4, 2, 1, 0, 0, 0, 0, 0, 4, 3, 1, 0, 0, 0, 0, 0, 4, 3, 2, 0, 0, 0, 0, 0, 4, 3, 2, 1, 0, 0, 0, 0, 4, 3, 2, 1, 1, 0, 0, 0, 4, 3, 2, 2, 1, 0, 0, 0, 4, 3, 3, 2, 1, 0, 0, 0, 4, 4, 3, 2, 1, 0, 0, 0, 5, 4, 3, 2, 1, 0, 0, 0, 6, 4, 3, 2, 1, 0, 0, 0, 6, 5, 3, 2, 1, 0, 0, 0, 6, 5, 4, 2, 1, 0, 0, 0, 6, 5, 4, 3, 1, 0, 0, 0, 6, 5, 4, 3, 2, 0, 0, 0, 6, 5, 4, 3, 2, 1, 0, 0, 6, 5, 4, 3, 2, 1, 1, 0, 6, 5, 4, 3, 2, 2, 1, 0, 6, 5, 4, 3, 3, 2, 1, 0, 6, 5, 4, 4, 3, 2, 1, 0, 6, 5, 5, 4, 3, 2, 1, 0, 6, 6, 5, 4, 3, 2, 1, 0, 7, 6, 5, 4, 3, 2, 1, 0, 8, 6, 5, 4, 3, 2, 1, 0, 8, 7, 5, 4, 3, 2, 1, 0, 8, 7, 6, 4, 3, 2, 1, 0, 8, 7, 6, 5, 3, 2, 1, 0, 8, 7, 6, 5, 4, 2, 1, 0, 8, 7, 6, 5, 4, 3, 1, 0, 8, 7, 6, 5, 4, 3, 2, 0, 8, 7, 6, 5, 4, 3, 2, 1, 8, 7, 6, 5, 4, 3, 2, 2, 8, 7, 6, 5, 4, 3, 3, 2, 8, 7, 6, 5, 4, 4, 3, 2, 8, 7, 6, 5, 5, 4, 3, 2, 8, 7, 6, 6, 5, 4, 3, 2, 8, 7, 7, 6, 5, 4, 3, 2, 8, 8, 7, 6, 5, 4, 3, 2, 8, 8, 8, 6, 5, 4, 3, 2, 8, 8, 8, 7, 5, 4, 3, 2, 8, 8, 8, 7, 6, 4, 3, 2, 8, 8, 8, 7, 6, 5, 3, 2, 8, 8, 8, 7, 6, 5, 4, 2, 8, 8, 8, 7, 6, 5, 4, 3, 8, 8, 8, 7, 6, 5, 4, 4, 8, 8, 8, 7, 6, 5, 5, 4, 8, 8, 8, 7, 6, 6, 5, 4, 8, 8, 8, 7, 7, 6, 5, 4, 8, 8, 8, 8, 7, 6, 5, 4, 8, 8, 8, 8, 8, 6, 5, 4, 8, 8, 8, 8, 8, 7, 5, 4, 8, 8, 8, 8, 8, 7, 6, 4, 8, 8, 8, 8, 8, 7, 6, 5, 8, 8, 8, 8, 8, 7, 6, 6, 8, 8, 8, 8, 8, 7, 7, 6, 8, 8, 8, 8, 8, 8, 7, 6, 8, 8, 8, 8, 8, 8, 8, 6, 8, 8, 8, 8, 8, 8, 8, 7, 8, 8, 8, 8, 8, 8, 8, 8]          # jpegload.d:295-304
s_idct_col_table = [1, 1, 2, 3, 3, 3, 3, 3, 3, 4, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 6, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8]          # jpegload.d:306-306
s_max_rc = [17, 18, 34, 50, 50, 51, 52, 52, 52, 68, 84, 84, 84, 84, 85, 86, 86, 86, 86, 86, 102, 118, 118, 118, 118, 118, 118, 119, 120, 120, 120, 120, 120, 120, 120, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136]          # jpegload.d:2132-2137
SCALEBITS = 16          # jpegload.d:2080
ONE_HALF = 1 << (SCALEBITS-1)          # jpegload.d:2081

def idct_dc_d(pSrc_ptr):          # jpegload.d:308-319
    k = to_int(((pSrc_ptr[0] + 4) >> 3) + 128)
    k = to_int(CLAMP(k))
    k = to_int(k | (k<<8))
    k = to_int(k | (k<<16))
    return k

def Matrix44_iadd_d(this, a):          # jpegload.d:842-850
    for r in range(NUM_ROWS):
        this.set(r, 0, this.at(r, 0) + a.at(r, 0))
        this.set(r, 1, this.at(r, 1) + a.at(r, 1))
        this.set(r, 2, this.at(r, 2) + a.at(r, 2))
        this.set(r, 3, this.at(r, 3) + a.at(r, 3))

def Matrix44_isub_d(this, a):          # jpegload.d:852-860
    for r in range(NUM_ROWS):
        this.set(r, 0, this.at(r, 0) - a.at(r, 0))
        this.set(r, 1, this.at(r, 1) - a.at(r, 1))
        this.set(r, 2, this.at(r, 2) - a.at(r, 2))
        this.set(r, 3, this.at(r, 3) - a.at(r, 3))

def Matrix44_add_d(this, b, ret):          # jpegload.d:862-872
    a = this
    for r in range(NUM_ROWS):
        ret.set(r, 0, a.at(r, 0) + b.at(r, 0))
        ret.set(r, 1, a.at(r, 1) + b.at(r, 1))
        ret.set(r, 2, a.at(r, 2) + b.at(r, 2))
        ret.set(r, 3, a.at(r, 3) + b.at(r, 3))

def Matrix44_sub_d(this, b, ret):          # jpegload.d:874-884
    a = this
    for r in range(NUM_ROWS):
        ret.set(r, 0, a.at(r, 0) - b.at(r, 0))
        ret.set(r, 1, a.at(r, 1) - b.at(r, 1))
        ret.set(r, 2, a.at(r, 2) - b.at(r, 2))
        ret.set(r, 3, a.at(r, 3) - b.at(r, 3))

def add_and_store_d(pDst, a, b):          # jpegload.d:886-893
    for r in range(4):
        pDst[0*8 + r] = to_short(a.at(r, 0) + b.at(r, 0))
        pDst[1*8 + r] = to_short(a.at(r, 1) + b.at(r, 1))
        pDst[2*8 + r] = to_short(a.at(r, 2) + b.at(r, 2))
        pDst[3*8 + r] = to_short(a.at(r, 3) + b.at(r, 3))

def sub_and_store_d(pDst, a, b):          # jpegload.d:895-902
    for r in range(4):
        pDst[0*8 + r] = to_short(a.at(r, 0) - b.at(r, 0))
        pDst[1*8 + r] = to_short(a.at(r, 1) - b.at(r, 1))
        pDst[2*8 + r] = to_short(a.at(r, 2) - b.at(r, 2))
        pDst[3*8 + r] = to_short(a.at(r, 3) - b.at(r, 3))

def create_look_ups_body_d(m_crr, m_cbb, m_crg, m_cbg, i):          # jpegload.d:2085-2094
    k = i - 128
    m_crr[i] = ( FIX(1.40200)  * k + ONE_HALF) >> SCALEBITS
    m_cbb[i] = ( FIX(1.77200)  * k + ONE_HALF) >> SCALEBITS
    m_crg[i] = (-FIX(0.71414)) * k
    m_cbg[i] = (-FIX(0.34414)) * k + ONE_HALF

def mcu_expand_tail_d(P, Q, R, S, temp_block, pDst_ptr, idct_4x4):          # jpegload.d:2230-2252
    a = Matrix44(P + Q)
    P -= Q
    b = P
    c = Matrix44(R + S)
    R -= S
    d = R
    Matrix44.add_and_store(temp_block, a, c)
    idct_4x4(temp_block, pDst_ptr)
    pDst_ptr += 64
    Matrix44.sub_and_store(temp_block, a, c)
    idct_4x4(temp_block, pDst_ptr)
    pDst_ptr += 64
    Matrix44.add_and_store(temp_block, b, d)
    idct_4x4(temp_block, pDst_ptr)
    pDst_ptr += 64
    Matrix44.sub_and_store(temp_block, b, d)
    idct_4x4(temp_block, pDst_ptr)
    pDst_ptr += 64

def expanded_convert_simd_d(Py, Y_ofs, Cb_ofs, Cr_ofs, j, d):          # jpegload.d:2754-2817
    mm_y = _mm_loadu_si32(Py, Y_ofs + j)
    mm_cb = _mm_loadu_si32(Py, Cb_ofs + j)
    mm_cr = _mm_loadu_si32(Py, Cr_ofs + j)
    zero = _mm_setzero_si128()
    mm_y = _mm_unpacklo_epi8(mm_y, zero)
    mm_cb = _mm_unpacklo_epi8(mm_cb, zero)
    mm_cr = _mm_unpacklo_epi8(mm_cr, zero)
    mm_y = _mm_unpacklo_epi16(mm_y, zero)
    mm_cb = _mm_unpacklo_epi16(mm_cb, zero)
    mm_cr = _mm_unpacklo_epi16(mm_cr, zero)
    mm_128 = _mm_set1_epi32(128)
    mm_crr = _mm_mullo_epi32(mm_cr - mm_128, _mm_set1_epi32( FIX(1.40200) ) )
    mm_crg = _mm_mullo_epi32(mm_cr - mm_128, _mm_set1_epi32(-FIX(0.71414) ) )
    mm_cbg = _mm_mullo_epi32(mm_cb - mm_128, _mm_set1_epi32(-FIX(0.34414) ) )
    mm_cbb = _mm_mullo_epi32(mm_cb - mm_128, _mm_set1_epi32( FIX(1.77200) ) )
    mm_ONE_HALF = _mm_set1_epi32(ONE_HALF)
    mm_crr += mm_ONE_HALF
    mm_cbg += mm_ONE_HALF
    mm_cbb += mm_ONE_HALF
    mm_crr = _mm_srai_epi32(mm_crr, 16)
    mm_cbb = _mm_srai_epi32(mm_cbb, 16)
    mm_crg = _mm_srai_epi32(mm_crg + mm_cbg, 16)
    mm_crr += mm_y
    mm_crg += mm_y
    mm_cbb += mm_y
    A = mm_crr
    B = mm_crg
    C = mm_cbb
    D = _mm_set1_epi32(255)
    A, B, C, D = _MM_TRANSPOSE4_PS(A, B, C, D)
    Ai = _mm_packs_epi32(A, B)
    Ci = _mm_packs_epi32(C, D)
    Ai = _mm_packus_epi16(Ai, Ci)
    _mm_storeu_si128(d, 0, Ai)
    d += 16
    return d
# END GENERATED MISC


def shl(x, n):                     # D `<<` on int: a plain two's-complement shift
    return (i32(x).view(np.uint32) << np.uint32(n)).view(I32)


def DESCALE(x, n):                 # :137-140
    return (i32(x) + I32(SCALEDONE << (n - 1))) >> I32(n)


def DESCALE_ZEROSHIFT(x, n):       # :142-145
    return (i32(x) + I32(128 << n) + I32(SCALEDONE << (n - 1))) >> I32(n)


def CLAMP(i):                      # :147-152
    i = i32(i).copy()
    i[i < 0] = 0
    i[i > 255] = 255
    return i.astype(np.uint8)


# ---------------------------------------------------------------------------------------------- :156-214
def Row_idct(NONZERO_COLS, pTemp, pSrc):
    """pTemp: (N, 8) int32 view of one temp row; pSrc: (N, 8) int16 view of one coefficient row"""
    if NONZERO_COLS == 0:
        return
    if NONZERO_COLS == 1:
        dcval = shl(pSrc[:, 0].astype(I32), PASS1_BITS)
        for k in range(8):
            pTemp[:, k] = dcval
        return

    def ACCESS_COL(x):
        return pSrc[:, x].astype(I32) if x < NONZERO_COLS else I32(0)

    z2 = ACCESS_COL(2); z3 = ACCESS_COL(6)

    z1 = i32((z2 + z3) * I32(FIX_0_541196100))
    tmp2 = i32(z1 + z3 * I32(-FIX_1_847759065))
    tmp3 = i32(z1 + z2 * I32(FIX_0_765366865))

    tmp0 = shl(ACCESS_COL(0) + ACCESS_COL(4), CONST_BITS)
    tmp1 = shl(ACCESS_COL(0) - ACCESS_COL(4), CONST_BITS)

    tmp10 = i32(tmp0 + tmp3); tmp13 = i32(tmp0 - tmp3); tmp11 = i32(tmp1 + tmp2); tmp12 = i32(tmp1 - tmp2)

    atmp0 = ACCESS_COL(7); atmp1 = ACCESS_COL(5); atmp2 = ACCESS_COL(3); atmp3 = ACCESS_COL(1)

    bz1 = i32(atmp0 + atmp3); bz2 = i32(atmp1 + atmp2); bz3 = i32(atmp0 + atmp2); bz4 = i32(atmp1 + atmp3)
    bz5 = i32((bz3 + bz4) * I32(FIX_1_175875602))

    az1 = i32(bz1 * I32(-FIX_0_899976223))
    az2 = i32(bz2 * I32(-FIX_2_562915447))
    az3 = i32(bz3 * I32(-FIX_1_961570560) + bz5)
    az4 = i32(bz4 * I32(-FIX_0_390180644) + bz5)

    btmp0 = i32(atmp0 * I32(FIX_0_298631336) + az1 + az3)
    btmp1 = i32(atmp1 * I32(FIX_2_053119869) + az2 + az4)
    btmp2 = i32(atmp2 * I32(FIX_3_072711026) + az2 + az3)
    btmp3 = i32(atmp3 * I32(FIX_1_501321110) + az1 + az4)

    pTemp[:, 0] = DESCALE(tmp10 + btmp3, CONST_BITS - PASS1_BITS)
    pTemp[:, 7] = DESCALE(tmp10 - btmp3, CONST_BITS - PASS1_BITS)
    pTemp[:, 1] = DESCALE(tmp11 + btmp2, CONST_BITS - PASS1_BITS)
    pTemp[:, 6] = DESCALE(tmp11 - btmp2, CONST_BITS - PASS1_BITS)
    pTemp[:, 2] = DESCALE(tmp12 + btmp1, CONST_BITS - PASS1_BITS)
    pTemp[:, 5] = DESCALE(tmp12 - btmp1, CONST_BITS - PASS1_BITS)
    pTemp[:, 3] = DESCALE(tmp13 + btmp0, CONST_BITS - PASS1_BITS)
    pTemp[:, 4] = DESCALE(tmp13 - btmp0, CONST_BITS - PASS1_BITS)


# ---------------------------------------------------------------------------------------------- :218-292
def Col_idct(NONZERO_ROWS, pDst, pTemp, col):
    """pDst: (N, 64) uint8 block, pTemp: (N, 64) int32 temp, both indexed from column `col` with stride 8"""
    assert NONZERO_ROWS > 0
    if NONZERO_ROWS == 1:
        dcval = DESCALE_ZEROSHIFT(pTemp[:, col + 0], PASS1_BITS + 3)
        dcval_clamped = CLAMP(dcval)
        for k in range(8):
            pDst[:, col + k * 8] = dcval_clamped
        return

    def ACCESS_ROW(x):
        return pTemp[:, col + x * 8] if x < NONZERO_ROWS else I32(0)

    z2 = ACCESS_ROW(2)
    z3 = ACCESS_ROW(6)

    z1 = i32((z2 + z3) * I32(FIX_0_541196100))
    tmp2 = i32(z1 + z3 * I32(-FIX_1_847759065))
    tmp3 = i32(z1 + z2 * I32(FIX_0_765366865))

    tmp0 = shl(ACCESS_ROW(0) + ACCESS_ROW(4), CONST_BITS)
    tmp1 = shl(ACCESS_ROW(0) - ACCESS_ROW(4), CONST_BITS)

    tmp10 = i32(tmp0 + tmp3); tmp13 = i32(tmp0 - tmp3); tmp11 = i32(tmp1 + tmp2); tmp12 = i32(tmp1 - tmp2)

    atmp0 = ACCESS_ROW(7); atmp1 = ACCESS_ROW(5); atmp2 = ACCESS_ROW(3); atmp3 = ACCESS_ROW(1)

    bz1 = i32(atmp0 + atmp3); bz2 = i32(atmp1 + atmp2); bz3 = i32(atmp0 + atmp2); bz4 = i32(atmp1 + atmp3)
    bz5 = i32((bz3 + bz4) * I32(FIX_1_175875602))

    az1 = i32(bz1 * I32(-FIX_0_899976223))
    az2 = i32(bz2 * I32(-FIX_2_562915447))
    az3 = i32(bz3 * I32(-FIX_1_961570560) + bz5)
    az4 = i32(bz4 * I32(-FIX_0_390180644) + bz5)

    btmp0 = i32(atmp0 * I32(FIX_0_298631336) + az1 + az3)
    btmp1 = i32(atmp1 * I32(FIX_2_053119869) + az2 + az4)
    btmp2 = i32(atmp2 * I32(FIX_3_072711026) + az2 + az3)
    btmp3 = i32(atmp3 * I32(FIX_1_501321110) + az1 + az4)

    i = DESCALE_ZEROSHIFT(tmp10 + btmp3, CONST_BITS + PASS1_BITS + 3)
    pDst[:, col + 8 * 0] = CLAMP(i)

    i = DESCALE_ZEROSHIFT(tmp10 - btmp3, CONST_BITS + PASS1_BITS + 3)
    pDst[:, col + 8 * 7] = CLAMP(i)

    i = DESCALE_ZEROSHIFT(tmp11 + btmp2, CONST_BITS + PASS1_BITS + 3)
    pDst[:, col + 8 * 1] = CLAMP(i)

    i = DESCALE_ZEROSHIFT(tmp11 - btmp2, CONST_BITS + PASS1_BITS + 3)
    pDst[:, col + 8 * 6] = CLAMP(i)

    i = DESCALE_ZEROSHIFT(tmp12 + btmp1, CONST_BITS + PASS1_BITS + 3)
    pDst[:, col + 8 * 2] = CLAMP(i)

    i = DESCALE_ZEROSHIFT(tmp12 - btmp1, CONST_BITS + PASS1_BITS + 3)
    pDst[:, col + 8 * 5] = CLAMP(i)

    i = DESCALE_ZEROSHIFT(tmp13 + btmp0, CONST_BITS + PASS1_BITS + 3)
    pDst[:, col + 8 * 3] = CLAMP(i)

    i = DESCALE_ZEROSHIFT(tmp13 - btmp0, CONST_BITS + PASS1_BITS + 3)
    pDst[:, col + 8 * 4] = CLAMP(i)


# ---------------------------------------------------------------------------------------------- :156-292 again, mechanically
class Ptr:
    """a D pointer walking N blocks at once: p[k] is element k (a column of the (N, 64) array) read or written for all blocks;
    reading a `short` promotes to int, as every D expression on it does"""
    def __init__(self, arr, base):
        self.arr, self.base = arr, base

    def __getitem__(self, k):
        v = self.arr[:, self.base + k]
        return v.astype(I32) if v.dtype == np.int16 else v

    def __setitem__(self, k, v):
        self.arr[:, self.base + k] = v


def to_ubyte(x):                   # cast(ubyte): the low 8 bits
    return i32(x).astype(np.uint8)


# BEGIN GENERATED ROWCOL (tools/make_ref_literal.py from jpegload.d)
def Row_idct_d(NONZERO_COLS, pTemp, pSrc):          # jpegload.d:159-212
    ACCESS_COL = lambda x: pSrc[x] if x < NONZERO_COLS else 0      # template ACCESS_COL: "cast(int)pSrc[x]" or "0"
    if NONZERO_COLS == 0:
        pass
        # nothing
    elif NONZERO_COLS == 1:
        pass
        dcval = (pSrc[0] << PASS1_BITS)
        pTemp[0] = dcval
        pTemp[1] = dcval
        pTemp[2] = dcval
        pTemp[3] = dcval
        pTemp[4] = dcval
        pTemp[5] = dcval
        pTemp[6] = dcval
        pTemp[7] = dcval
    else:
        pass
        # ACCESS_COL() will be optimized at compile time to either an array access, or 0.
        ##define ACCESS_COL(x) (((x) < NONZERO_COLS) ? (int)pSrc[x] : 0)
        z2 = ACCESS_COL(2)
        z3 = ACCESS_COL(6)
        z1 = (z2 + z3)*FIX_0_541196100
        tmp2 = z1 + z3*(-FIX_1_847759065)
        tmp3 = z1 + z2*FIX_0_765366865
        tmp0 = (ACCESS_COL(0) + ACCESS_COL(4)) << CONST_BITS
        tmp1 = (ACCESS_COL(0) - ACCESS_COL(4)) << CONST_BITS
        tmp10 = tmp0 + tmp3
        tmp13 = tmp0 - tmp3
        tmp11 = tmp1 + tmp2
        tmp12 = tmp1 - tmp2
        atmp0 = ACCESS_COL(7)
        atmp1 = ACCESS_COL(5)
        atmp2 = ACCESS_COL(3)
        atmp3 = ACCESS_COL(1)
        bz1 = atmp0 + atmp3
        bz2 = atmp1 + atmp2
        bz3 = atmp0 + atmp2
        bz4 = atmp1 + atmp3
        bz5 = (bz3 + bz4)*FIX_1_175875602
        az1 = bz1*(-FIX_0_899976223)
        az2 = bz2*(-FIX_2_562915447)
        az3 = bz3*(-FIX_1_961570560) + bz5
        az4 = bz4*(-FIX_0_390180644) + bz5
        btmp0 = atmp0*FIX_0_298631336 + az1 + az3
        btmp1 = atmp1*FIX_2_053119869 + az2 + az4
        btmp2 = atmp2*FIX_3_072711026 + az2 + az3
        btmp3 = atmp3*FIX_1_501321110 + az1 + az4
        pTemp[0] = DESCALE(tmp10 + btmp3, CONST_BITS-PASS1_BITS)
        pTemp[7] = DESCALE(tmp10 - btmp3, CONST_BITS-PASS1_BITS)
        pTemp[1] = DESCALE(tmp11 + btmp2, CONST_BITS-PASS1_BITS)
        pTemp[6] = DESCALE(tmp11 - btmp2, CONST_BITS-PASS1_BITS)
        pTemp[2] = DESCALE(tmp12 + btmp1, CONST_BITS-PASS1_BITS)
        pTemp[5] = DESCALE(tmp12 - btmp1, CONST_BITS-PASS1_BITS)
        pTemp[3] = DESCALE(tmp13 + btmp0, CONST_BITS-PASS1_BITS)
        pTemp[4] = DESCALE(tmp13 - btmp0, CONST_BITS-PASS1_BITS)


def Col_idct_d(NONZERO_ROWS, pDst_ptr, pTemp):          # jpegload.d:221-290
    ACCESS_ROW = lambda x: pTemp[x * 8] if x < NONZERO_ROWS else 0  # template ACCESS_ROW: "pTemp[x*8]" or "0"
    assert NONZERO_ROWS > 0
    if NONZERO_ROWS == 1:
        pass
        dcval = DESCALE_ZEROSHIFT(pTemp[0], PASS1_BITS+3)
        dcval_clamped = to_ubyte(CLAMP(dcval))
        pDst_ptr[0*8] = dcval_clamped
        pDst_ptr[1*8] = dcval_clamped
        pDst_ptr[2*8] = dcval_clamped
        pDst_ptr[3*8] = dcval_clamped
        pDst_ptr[4*8] = dcval_clamped
        pDst_ptr[5*8] = dcval_clamped
        pDst_ptr[6*8] = dcval_clamped
        pDst_ptr[7*8] = dcval_clamped
    else:
        pass
        # ACCESS_ROW() will be optimized at compile time to either an array access, or 0.
        ##define ACCESS_ROW(x) (((x) < NONZERO_ROWS) ? pTemp[x * 8] : 0)
        z2 = ACCESS_ROW(2)
        z3 = ACCESS_ROW(6)
        z1 = (z2 + z3)*FIX_0_541196100
        tmp2 = z1 + z3*(-FIX_1_847759065)
        tmp3 = z1 + z2*FIX_0_765366865
        tmp0 = (ACCESS_ROW(0) + ACCESS_ROW(4)) << CONST_BITS
        tmp1 = (ACCESS_ROW(0) - ACCESS_ROW(4)) << CONST_BITS
        tmp10 = tmp0 + tmp3
        tmp13 = tmp0 - tmp3
        tmp11 = tmp1 + tmp2
        tmp12 = tmp1 - tmp2
        atmp0 = ACCESS_ROW(7)
        atmp1 = ACCESS_ROW(5)
        atmp2 = ACCESS_ROW(3)
        atmp3 = ACCESS_ROW(1)
        bz1 = atmp0 + atmp3
        bz2 = atmp1 + atmp2
        bz3 = atmp0 + atmp2
        bz4 = atmp1 + atmp3
        bz5 = (bz3 + bz4)*FIX_1_175875602
        az1 = bz1*(-FIX_0_899976223)
        az2 = bz2*(-FIX_2_562915447)
        az3 = bz3*(-FIX_1_961570560) + bz5
        az4 = bz4*(-FIX_0_390180644) + bz5
        btmp0 = atmp0*FIX_0_298631336 + az1 + az3
        btmp1 = atmp1*FIX_2_053119869 + az2 + az4
        btmp2 = atmp2*FIX_3_072711026 + az2 + az3
        btmp3 = atmp3*FIX_1_501321110 + az1 + az4
        i = DESCALE_ZEROSHIFT(tmp10 + btmp3, CONST_BITS+PASS1_BITS+3)
        pDst_ptr[8*0] = to_ubyte(CLAMP(i))
        i = DESCALE_ZEROSHIFT(tmp10 - btmp3, CONST_BITS+PASS1_BITS+3)
        pDst_ptr[8*7] = to_ubyte(CLAMP(i))
        i = DESCALE_ZEROSHIFT(tmp11 + btmp2, CONST_BITS+PASS1_BITS+3)
        pDst_ptr[8*1] = to_ubyte(CLAMP(i))
        i = DESCALE_ZEROSHIFT(tmp11 - btmp2, CONST_BITS+PASS1_BITS+3)
        pDst_ptr[8*6] = to_ubyte(CLAMP(i))
        i = DESCALE_ZEROSHIFT(tmp12 + btmp1, CONST_BITS+PASS1_BITS+3)
        pDst_ptr[8*2] = to_ubyte(CLAMP(i))
        i = DESCALE_ZEROSHIFT(tmp12 - btmp1, CONST_BITS+PASS1_BITS+3)
        pDst_ptr[8*5] = to_ubyte(CLAMP(i))
        i = DESCALE_ZEROSHIFT(tmp13 + btmp0, CONST_BITS+PASS1_BITS+3)
        pDst_ptr[8*3] = to_ubyte(CLAMP(i))
        i = DESCALE_ZEROSHIFT(tmp13 - btmp0, CONST_BITS+PASS1_BITS+3)
        pDst_ptr[8*4] = to_ubyte(CLAMP(i))

# END GENERATED ROWCOL


assert len(s_idct_row_table) == 512 and len(s_idct_col_table) == 64 and len(s_max_rc) == 64 and sorted(g_ZAG) == list(range(64))     # (generated above)


# ---------------------------------------------------------------------------------------------- :308-376
def idct(pSrc_ptr, block_max_zag):
    """pSrc_ptr: (N, 64) int16, ONE block_max_zag for all N blocks -> (N, 64) uint8"""
    assert 1 <= block_max_zag <= 64
    N = pSrc_ptr.shape[0]
    pDst = np.zeros((N, 64), np.uint8)
    if block_max_zag <= 1:
        k = idct_dc_d(Ptr(pSrc_ptr, 0))                       # the four statements on k (:314-317)
        for i in range(8):                                    # *cast(int*)&pDst_ptr[0] = k; *cast(int*)&pDst_ptr[4] = k; pDst_ptr += 8  (little-endian stores)
            pDst[:, i * 8: i * 8 + 4] = k.view(np.uint8).reshape(N, 4)
            pDst[:, i * 8 + 4: i * 8 + 8] = k.view(np.uint8).reshape(N, 4)
        return pDst

    temp = np.zeros((N, 64), I32)                             # int[64] temp (D zero-initialises)
    for row in range(8):                                      # Row!(n).idct(pTemp, pSrc); pSrc += 8; pTemp += 8
        Row_idct_d(s_idct_row_table[(block_max_zag - 1) * 8 + row], Ptr(temp, row * 8), Ptr(pSrc_ptr, row * 8))
    nonzero_rows = s_idct_col_table[block_max_zag - 1]
    for col in range(8):                                      # Col!(n).idct(pDst_ptr, pTemp); pTemp++; pDst_ptr++
        Col_idct_d(nonzero_rows, Ptr(pDst, col), Ptr(temp, col))
    return pDst


# ---------------------------------------------------------------------------------------------- :378-397
def idct_4x4(pSrc_ptr):
    N = pSrc_ptr.shape[0]
    pDst = np.zeros((N, 64), np.uint8)
    temp = np.zeros((N, 64), I32)
    for row in range(4):
        Row_idct_d(4, Ptr(temp, row * 8), Ptr(pSrc_ptr, row * 8))
    for col in range(8):
        Col_idct_d(4, Ptr(pDst, col), Ptr(temp, col))
    return pDst


# ---------------------------------------------------------------------------------------------- :827-903
class Matrix44:
    NUM_ROWS = 4
    NUM_COLS = 4

    def __init__(self, N=None, m=None):
        if isinstance(N, Matrix44):                           # this(in Matrix44 m): a copy (:836-838)
            N, m = None, N
        if m is not None:
            self.v = [[m.v[r][c].copy() for c in range(4)] for r in range(4)]
        else:
            self.v = [[np.zeros(N, I32) for _ in range(4)] for _ in range(4)]

    def at(self, r, c):
        return self.v[r][c]

    def set(self, r, c, x):                                   # `M.at(r, c) = x` (a literal 0 broadcasts over the N blocks)
        self.v[r][c] = np.broadcast_to(i32(x), self.v[r][c].shape).copy()

    # the operator and store bodies are the generated Matrix44_*_d / *_and_store_d functions (:842-902)
    def __iadd__(self, a):
        Matrix44_iadd_d(self, a)
        return self

    def __isub__(self, a):
        Matrix44_isub_d(self, a)
        return self

    def __add__(self, b):
        ret = Matrix44(len(self.v[0][0]))
        Matrix44_add_d(self, b, ret)
        return ret

    def __sub__(self, b):
        ret = Matrix44(len(self.v[0][0]))
        Matrix44_sub_d(self, b, ret)
        return ret

    @staticmethod
    def add_and_store(pDst, a, b):                           # pDst: (N, 64) int16
        add_and_store_d(Ptr(pDst, 0), a, b)

    @staticmethod
    def sub_and_store(pDst, a, b):
        sub_and_store_d(Ptr(pDst, 0), a, b)


# ---------------------------------------------------------------------------------------------- :905-911
FRACT_BITS = 10
SCALE = 1 << FRACT_BITS


def D(i):
    return (i32(i) + I32(SCALE >> 1)) >> I32(FRACT_BITS)


def F(i):                          # enum F(float i) = (cast(int)((i) * SCALE + 0.5f)): float arithmetic, truncation toward zero
    return I32(int(np.float32(np.float32(i) * np.float32(SCALE)) + np.float32(0.5)))


# the sixteen constants, as SURVEY.md 7.2-5 lists them
assert [int(F(x)) for x in (0.415735, 0.791065, -0.352443, 0.277785, 0.022887, -0.097545, 0.490393, 0.865723,
                            0.906127, -0.318190, 0.212608, -0.180240, -0.074658, 0.513280, 0.768178, -0.375330)] == \
       [426, 810, -360, 284, 23, -99, 502, 887, 928, -325, 218, -184, -75, 526, 787, -383]


def _AT(pSrc, NUM_ROWS, NUM_COLS):
    def AT(c, r):                  # :917-919  (c >= NUM_COLS || r >= NUM_ROWS ? 0 : pSrc[c+r*8])
        return I32(0) if (c >= NUM_COLS or r >= NUM_ROWS) else pSrc[:, c + r * 8].astype(I32)
    return AT


# BEGIN GENERATED (tools/make_ref_literal.py from jpegload.d)
def P_Q_calc(NUM_ROWS, NUM_COLS, P, Q, pSrc):          # jpegload.d:916-990
    AT = _AT(pSrc, NUM_ROWS, NUM_COLS)
    #auto AT (int c, int r) nothrow @trusted @nogc { return (c >= NUM_COLS || r >= NUM_ROWS ? 0 : pSrc[c+r*8]); }
    # 4x8 = 4x8 times 8x8, matrix 0 is constant
    X000 = AT(0, 0)
    X001 = AT(0, 1)
    X002 = AT(0, 2)
    X003 = AT(0, 3)
    X004 = AT(0, 4)
    X005 = AT(0, 5)
    X006 = AT(0, 6)
    X007 = AT(0, 7)
    X010 = D(F(0.415735) * AT(1, 0) + F(0.791065) * AT(3, 0) + F(-0.352443) * AT(5, 0) + F(0.277785) * AT(7, 0))
    X011 = D(F(0.415735) * AT(1, 1) + F(0.791065) * AT(3, 1) + F(-0.352443) * AT(5, 1) + F(0.277785) * AT(7, 1))
    X012 = D(F(0.415735) * AT(1, 2) + F(0.791065) * AT(3, 2) + F(-0.352443) * AT(5, 2) + F(0.277785) * AT(7, 2))
    X013 = D(F(0.415735) * AT(1, 3) + F(0.791065) * AT(3, 3) + F(-0.352443) * AT(5, 3) + F(0.277785) * AT(7, 3))
    X014 = D(F(0.415735) * AT(1, 4) + F(0.791065) * AT(3, 4) + F(-0.352443) * AT(5, 4) + F(0.277785) * AT(7, 4))
    X015 = D(F(0.415735) * AT(1, 5) + F(0.791065) * AT(3, 5) + F(-0.352443) * AT(5, 5) + F(0.277785) * AT(7, 5))
    X016 = D(F(0.415735) * AT(1, 6) + F(0.791065) * AT(3, 6) + F(-0.352443) * AT(5, 6) + F(0.277785) * AT(7, 6))
    X017 = D(F(0.415735) * AT(1, 7) + F(0.791065) * AT(3, 7) + F(-0.352443) * AT(5, 7) + F(0.277785) * AT(7, 7))
    X020 = AT(4, 0)
    X021 = AT(4, 1)
    X022 = AT(4, 2)
    X023 = AT(4, 3)
    X024 = AT(4, 4)
    X025 = AT(4, 5)
    X026 = AT(4, 6)
    X027 = AT(4, 7)
    X030 = D(F(0.022887) * AT(1, 0) + F(-0.097545) * AT(3, 0) + F(0.490393) * AT(5, 0) + F(0.865723) * AT(7, 0))
    X031 = D(F(0.022887) * AT(1, 1) + F(-0.097545) * AT(3, 1) + F(0.490393) * AT(5, 1) + F(0.865723) * AT(7, 1))
    X032 = D(F(0.022887) * AT(1, 2) + F(-0.097545) * AT(3, 2) + F(0.490393) * AT(5, 2) + F(0.865723) * AT(7, 2))
    X033 = D(F(0.022887) * AT(1, 3) + F(-0.097545) * AT(3, 3) + F(0.490393) * AT(5, 3) + F(0.865723) * AT(7, 3))
    X034 = D(F(0.022887) * AT(1, 4) + F(-0.097545) * AT(3, 4) + F(0.490393) * AT(5, 4) + F(0.865723) * AT(7, 4))
    X035 = D(F(0.022887) * AT(1, 5) + F(-0.097545) * AT(3, 5) + F(0.490393) * AT(5, 5) + F(0.865723) * AT(7, 5))
    X036 = D(F(0.022887) * AT(1, 6) + F(-0.097545) * AT(3, 6) + F(0.490393) * AT(5, 6) + F(0.865723) * AT(7, 6))
    X037 = D(F(0.022887) * AT(1, 7) + F(-0.097545) * AT(3, 7) + F(0.490393) * AT(5, 7) + F(0.865723) * AT(7, 7))
    # 4x4 = 4x8 times 8x4, matrix 1 is constant
    P.set(0, 0, X000)
    P.set(0, 1, D(X001 * F(0.415735) + X003 * F(0.791065) + X005 * F(-0.352443) + X007 * F(0.277785)))
    P.set(0, 2, X004)
    P.set(0, 3, D(X001 * F(0.022887) + X003 * F(-0.097545) + X005 * F(0.490393) + X007 * F(0.865723)))
    P.set(1, 0, X010)
    P.set(1, 1, D(X011 * F(0.415735) + X013 * F(0.791065) + X015 * F(-0.352443) + X017 * F(0.277785)))
    P.set(1, 2, X014)
    P.set(1, 3, D(X011 * F(0.022887) + X013 * F(-0.097545) + X015 * F(0.490393) + X017 * F(0.865723)))
    P.set(2, 0, X020)
    P.set(2, 1, D(X021 * F(0.415735) + X023 * F(0.791065) + X025 * F(-0.352443) + X027 * F(0.277785)))
    P.set(2, 2, X024)
    P.set(2, 3, D(X021 * F(0.022887) + X023 * F(-0.097545) + X025 * F(0.490393) + X027 * F(0.865723)))
    P.set(3, 0, X030)
    P.set(3, 1, D(X031 * F(0.415735) + X033 * F(0.791065) + X035 * F(-0.352443) + X037 * F(0.277785)))
    P.set(3, 2, X034)
    P.set(3, 3, D(X031 * F(0.022887) + X033 * F(-0.097545) + X035 * F(0.490393) + X037 * F(0.865723)))
    # 40 muls 24 adds
    # 4x4 = 4x8 times 8x4, matrix 1 is constant
    Q.set(0, 0, D(X001 * F(0.906127) + X003 * F(-0.318190) + X005 * F(0.212608) + X007 * F(-0.180240)))
    Q.set(0, 1, X002)
    Q.set(0, 2, D(X001 * F(-0.074658) + X003 * F(0.513280) + X005 * F(0.768178) + X007 * F(-0.375330)))
    Q.set(0, 3, X006)
    Q.set(1, 0, D(X011 * F(0.906127) + X013 * F(-0.318190) + X015 * F(0.212608) + X017 * F(-0.180240)))
    Q.set(1, 1, X012)
    Q.set(1, 2, D(X011 * F(-0.074658) + X013 * F(0.513280) + X015 * F(0.768178) + X017 * F(-0.375330)))
    Q.set(1, 3, X016)
    Q.set(2, 0, D(X021 * F(0.906127) + X023 * F(-0.318190) + X025 * F(0.212608) + X027 * F(-0.180240)))
    Q.set(2, 1, X022)
    Q.set(2, 2, D(X021 * F(-0.074658) + X023 * F(0.513280) + X025 * F(0.768178) + X027 * F(-0.375330)))
    Q.set(2, 3, X026)
    Q.set(3, 0, D(X031 * F(0.906127) + X033 * F(-0.318190) + X035 * F(0.212608) + X037 * F(-0.180240)))
    Q.set(3, 1, X032)
    Q.set(3, 2, D(X031 * F(-0.074658) + X033 * F(0.513280) + X035 * F(0.768178) + X037 * F(-0.375330)))
    Q.set(3, 3, X036)
    # 40 muls 24 adds


def R_S_calc(NUM_ROWS, NUM_COLS, R, S, pSrc):          # jpegload.d:996-1070
    AT = _AT(pSrc, NUM_ROWS, NUM_COLS)
    #auto AT (int c, int r) nothrow @trusted @nogc { return (c >= NUM_COLS || r >= NUM_ROWS ? 0 : pSrc[c+r*8]); }
    # 4x8 = 4x8 times 8x8, matrix 0 is constant
    X100 = D(F(0.906127) * AT(1, 0) + F(-0.318190) * AT(3, 0) + F(0.212608) * AT(5, 0) + F(-0.180240) * AT(7, 0))
    X101 = D(F(0.906127) * AT(1, 1) + F(-0.318190) * AT(3, 1) + F(0.212608) * AT(5, 1) + F(-0.180240) * AT(7, 1))
    X102 = D(F(0.906127) * AT(1, 2) + F(-0.318190) * AT(3, 2) + F(0.212608) * AT(5, 2) + F(-0.180240) * AT(7, 2))
    X103 = D(F(0.906127) * AT(1, 3) + F(-0.318190) * AT(3, 3) + F(0.212608) * AT(5, 3) + F(-0.180240) * AT(7, 3))
    X104 = D(F(0.906127) * AT(1, 4) + F(-0.318190) * AT(3, 4) + F(0.212608) * AT(5, 4) + F(-0.180240) * AT(7, 4))
    X105 = D(F(0.906127) * AT(1, 5) + F(-0.318190) * AT(3, 5) + F(0.212608) * AT(5, 5) + F(-0.180240) * AT(7, 5))
    X106 = D(F(0.906127) * AT(1, 6) + F(-0.318190) * AT(3, 6) + F(0.212608) * AT(5, 6) + F(-0.180240) * AT(7, 6))
    X107 = D(F(0.906127) * AT(1, 7) + F(-0.318190) * AT(3, 7) + F(0.212608) * AT(5, 7) + F(-0.180240) * AT(7, 7))
    X110 = AT(2, 0)
    X111 = AT(2, 1)
    X112 = AT(2, 2)
    X113 = AT(2, 3)
    X114 = AT(2, 4)
    X115 = AT(2, 5)
    X116 = AT(2, 6)
    X117 = AT(2, 7)
    X120 = D(F(-0.074658) * AT(1, 0) + F(0.513280) * AT(3, 0) + F(0.768178) * AT(5, 0) + F(-0.375330) * AT(7, 0))
    X121 = D(F(-0.074658) * AT(1, 1) + F(0.513280) * AT(3, 1) + F(0.768178) * AT(5, 1) + F(-0.375330) * AT(7, 1))
    X122 = D(F(-0.074658) * AT(1, 2) + F(0.513280) * AT(3, 2) + F(0.768178) * AT(5, 2) + F(-0.375330) * AT(7, 2))
    X123 = D(F(-0.074658) * AT(1, 3) + F(0.513280) * AT(3, 3) + F(0.768178) * AT(5, 3) + F(-0.375330) * AT(7, 3))
    X124 = D(F(-0.074658) * AT(1, 4) + F(0.513280) * AT(3, 4) + F(0.768178) * AT(5, 4) + F(-0.375330) * AT(7, 4))
    X125 = D(F(-0.074658) * AT(1, 5) + F(0.513280) * AT(3, 5) + F(0.768178) * AT(5, 5) + F(-0.375330) * AT(7, 5))
    X126 = D(F(-0.074658) * AT(1, 6) + F(0.513280) * AT(3, 6) + F(0.768178) * AT(5, 6) + F(-0.375330) * AT(7, 6))
    X127 = D(F(-0.074658) * AT(1, 7) + F(0.513280) * AT(3, 7) + F(0.768178) * AT(5, 7) + F(-0.375330) * AT(7, 7))
    X130 = AT(6, 0)
    X131 = AT(6, 1)
    X132 = AT(6, 2)
    X133 = AT(6, 3)
    X134 = AT(6, 4)
    X135 = AT(6, 5)
    X136 = AT(6, 6)
    X137 = AT(6, 7)
    # 80 muls 48 adds
    # 4x4 = 4x8 times 8x4, matrix 1 is constant
    R.set(0, 0, X100)
    R.set(0, 1, D(X101 * F(0.415735) + X103 * F(0.791065) + X105 * F(-0.352443) + X107 * F(0.277785)))
    R.set(0, 2, X104)
    R.set(0, 3, D(X101 * F(0.022887) + X103 * F(-0.097545) + X105 * F(0.490393) + X107 * F(0.865723)))
    R.set(1, 0, X110)
    R.set(1, 1, D(X111 * F(0.415735) + X113 * F(0.791065) + X115 * F(-0.352443) + X117 * F(0.277785)))
    R.set(1, 2, X114)
    R.set(1, 3, D(X111 * F(0.022887) + X113 * F(-0.097545) + X115 * F(0.490393) + X117 * F(0.865723)))
    R.set(2, 0, X120)
    R.set(2, 1, D(X121 * F(0.415735) + X123 * F(0.791065) + X125 * F(-0.352443) + X127 * F(0.277785)))
    R.set(2, 2, X124)
    R.set(2, 3, D(X121 * F(0.022887) + X123 * F(-0.097545) + X125 * F(0.490393) + X127 * F(0.865723)))
    R.set(3, 0, X130)
    R.set(3, 1, D(X131 * F(0.415735) + X133 * F(0.791065) + X135 * F(-0.352443) + X137 * F(0.277785)))
    R.set(3, 2, X134)
    R.set(3, 3, D(X131 * F(0.022887) + X133 * F(-0.097545) + X135 * F(0.490393) + X137 * F(0.865723)))
    # 40 muls 24 adds
    # 4x4 = 4x8 times 8x4, matrix 1 is constant
    S.set(0, 0, D(X101 * F(0.906127) + X103 * F(-0.318190) + X105 * F(0.212608) + X107 * F(-0.180240)))
    S.set(0, 1, X102)
    S.set(0, 2, D(X101 * F(-0.074658) + X103 * F(0.513280) + X105 * F(0.768178) + X107 * F(-0.375330)))
    S.set(0, 3, X106)
    S.set(1, 0, D(X111 * F(0.906127) + X113 * F(-0.318190) + X115 * F(0.212608) + X117 * F(-0.180240)))
    S.set(1, 1, X112)
    S.set(1, 2, D(X111 * F(-0.074658) + X113 * F(0.513280) + X115 * F(0.768178) + X117 * F(-0.375330)))
    S.set(1, 3, X116)
    S.set(2, 0, D(X121 * F(0.906127) + X123 * F(-0.318190) + X125 * F(0.212608) + X127 * F(-0.180240)))
    S.set(2, 1, X122)
    S.set(2, 2, D(X121 * F(-0.074658) + X123 * F(0.513280) + X125 * F(0.768178) + X127 * F(-0.375330)))
    S.set(2, 3, X126)
    S.set(3, 0, D(X131 * F(0.906127) + X133 * F(-0.318190) + X135 * F(0.212608) + X137 * F(-0.180240)))
    S.set(3, 1, X132)
    S.set(3, 2, D(X131 * F(-0.074658) + X133 * F(0.513280) + X135 * F(0.768178) + X137 * F(-0.375330)))
    S.set(3, 3, X136)
    # 40 muls 24 adds

# END GENERATED


# ---------------------------------------------------------------------------------------------- :2132-2255
_CASES = [1*16+1, 1*16+2, 2*16+2, 3*16+2, 3*16+3, 3*16+4, 4*16+4, 5*16+4, 5*16+5, 5*16+6, 6*16+6, 7*16+6, 7*16+7, 7*16+8, 8*16+8]


def chroma_expand(pSrc_ptr, block_max_zag):
    """the chroma half of transform_mcu_expand for N chroma blocks sharing one m_mcu_block_max_zag:
    returns (temp_blocks (4, N, 64) int16 as stored by add/sub_and_store, samples (4, N, 64) uint8 after idct_4x4)"""
    N = pSrc_ptr.shape[0]
    P, Q, R, S = Matrix44(N), Matrix44(N), Matrix44(N), Matrix44(N)
    assert 1 <= block_max_zag <= 64
    max_zag = block_max_zag - 1
    if max_zag <= 0:
        max_zag = 0
    code = s_max_rc[max_zag]
    assert code in _CASES                                     # `default: assert(false)`
    NR, NC = code >> 4, code & 15                             # case R*16+C: P_Q!(R, C).calc / R_S!(R, C).calc
    P_Q_calc(NR, NC, P, Q, pSrc_ptr)
    R_S_calc(NR, NC, R, S, pSrc_ptr)

    # `auto a = Matrix44(P + Q); P -= Q; ...` up to the fourth idct_4x4: the generated mcu_expand_tail_d (:2230-2251)
    temps, samples = [], []
    temp_block = np.zeros((N, 64), np.int16)                  # jpgd_block_t[64] temp_block, zero-initialised, reused by all four

    def idct_4x4_here(tb, pDst_ptr):                          # idct_4x4(temp_block.ptr, pDst_ptr)
        temps.append(tb.copy()); samples.append(idct_4x4(tb))

    class _Dst:                                               # pDst_ptr += 64: the four blocks are collected in order
        def __iadd__(self, n):
            return self
    mcu_expand_tail_d(P, Q, R, S, temp_block, _Dst(), idct_4x4_here)
    return np.stack(temps), np.stack(samples)


# ---------------------------------------------------------------------------------------------- :2080-2094
def create_look_ups():
    m_crr = np.zeros(256, I32); m_cbb = np.zeros(256, I32); m_crg = np.zeros(256, I32); m_cbg = np.zeros(256, I32)
    for i in range(256):                                      # for (int i = 0; i <= 255; i++): the generated body (:2087-2091)
        create_look_ups_body_d(m_crr, m_cbb, m_crg, m_cbg, I32(i))
    return m_crr, m_cbb, m_crg, m_cbg


# ---------------------------------------------------------------------------------------------- :2731-2823
def _sse_pixels(mm_y, mm_cb, mm_cr):
    """the loop body of expanded_convert (:2752-2818, generated: expanded_convert_simd_d) on any number of 4-pixel groups at
    once: mm_y / mm_cb / mm_cr (..., 4) sample values -> (..., 4, 4) uint8 R,G,B,255"""
    shape = np.asarray(mm_y).shape
    n = int(np.prod(shape[:-1]))
    Py = np.zeros((n, 12), np.uint8)
    Py[:, 0:4] = np.asarray(mm_y).reshape(n, 4); Py[:, 4:8] = np.asarray(mm_cb).reshape(n, 4); Py[:, 8:12] = np.asarray(mm_cr).reshape(n, 4)
    d = BytesAt(np.zeros((n, 16), np.uint8))
    expanded_convert_simd_d(BytesAt(Py), 0, 4, 8, 0, d)
    return d.arr.reshape(shape[:-1] + (4, 4))


def expanded_convert(sample_buf, m_max_mcus_per_row, row):
    """one output scanline (RGBA8, 16 * m_max_mcus_per_row pixels) from one MCU row's sample buffer
    sample_buf: (m_max_mcus_per_row * 12 * 64,) uint8 -- per MCU: Y0..Y3, Cb0..Cb3, Cr0..Cr3 (m_expanded_blocks_per_mcu = 12)"""
    m_comp_h_samp0, m_max_mcu_x_size, m_expanded_blocks_per_component, m_expanded_blocks_per_mcu = 2, 16, 4, 12
    Py = (row // 8) * 64 * m_comp_h_samp0 + (row & 7) * 8
    out = []
    for i in range(m_max_mcus_per_row):
        for k in range(0, m_max_mcu_x_size, 8):
            Y_ofs = k * 8
            Cb_ofs = Y_ofs + 64 * m_expanded_blocks_per_component
            Cr_ofs = Y_ofs + 64 * m_expanded_blocks_per_component * 2
            for j in range(0, 8 - 3, 4):
                d = BytesAt(np.zeros((1, 16), np.uint8))
                expanded_convert_simd_d(BytesAt(sample_buf[None, :], Py), Y_ofs, Cb_ofs, Cr_ofs, j, d)
                out.append(d.arr[0])
        Py += 64 * m_expanded_blocks_per_mcu
    return np.concatenate(out)


def decode_h2v2_rgba(coeffs, max_zag, width, height):
    """the whole 4:2:0 path the way decompress_jpeg_image_from_stream drives it (:3753-3764, req_comps = 4):
    per MCU row transform_mcu_expand, then 16 x expanded_convert, first `width` pixels of each line kept"""
    mcus_per_row, mcus_per_col = (width + 15) // 16, (height + 15) // 16
    co = np.asarray(coeffs, np.int16).reshape(mcus_per_col, mcus_per_row, 6, 64)
    mz = np.full((mcus_per_col, mcus_per_row, 6), 64, np.int64) if max_zag is None else np.asarray(max_zag).reshape(mcus_per_col, mcus_per_row, 6)
    img = np.zeros((height, width * 4), np.uint8)
    for my in range(mcus_per_col):
        sample_buf = np.zeros(mcus_per_row * 12 * 64, np.uint8)
        for mx in range(mcus_per_row):
            base = mx * 12 * 64
            for blk in range(4):                                                        # Y IDCT
                sample_buf[base + blk * 64: base + blk * 64 + 64] = idct(co[my, mx, blk][None, :], int(mz[my, mx, blk]))[0]
            for i in range(2):                                                          # chroma IDCT, with upsampling
                _, smp = chroma_expand(co[my, mx, 4 + i][None, :], int(mz[my, mx, 4 + i]))
                for q in range(4):
                    o = base + (4 + i * 4 + q) * 64
                    sample_buf[o:o + 64] = smp[q, 0]
        for row in range(16):
            y = my * 16 + row
            if y >= height:
                break
            line = expanded_convert(sample_buf, mcus_per_row, row)
            img[y] = line[:width * 4]
    return img


def decode_h2v2_rgba_fast(coeffs, width, height):
    """decode_h2v2_rgba (dense, max_zag = 64) for frames of whole MCUs, vectorised over all MCUs at once: the same idct /
    chroma_expand / _sse_pixels, with the sample-buffer addressing of expanded_convert (:2735-2745) written as index arrays"""
    assert width % 16 == 0 and height % 16 == 0
    mr, mc = width // 16, height // 16
    co = np.asarray(coeffs, np.int16).reshape(mc * mr, 6, 64)
    yb = np.stack([idct(co[:, b], 64) for b in range(4)], axis=1)                       # (M, 4, 64)
    cb = np.moveaxis(chroma_expand(co[:, 4], 64)[1], 0, 1)                              # (M, 4, 64)
    cr = np.moveaxis(chroma_expand(co[:, 5], 64)[1], 0, 1)
    row = np.arange(16)[:, None]; col = np.arange(16)[None, :]
    idx_blk = (row // 8) * 2 + col // 8                                                 # Py: (row / 8) * 64 * 2 + (row & 7) * 8; Y_ofs = k * 8
    idx_in = (row & 7) * 8 + (col & 7)
    g = lambda a: a[:, idx_blk, idx_in].reshape(-1, 16, 4, 4)                            # rows of 4-pixel groups, as the j loop takes them
    px = _sse_pixels(g(yb), g(cb), g(cr))                                               # (M, 16, 4 groups, 4 px, 4 bytes)
    return px.reshape(mc, mr, 16, 16, 4).transpose(0, 2, 1, 3, 4).reshape(height, width * 4)
