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

# ---------------------------------------------------------------------------------------------- :120-152
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


def i32(x):
    return np.asarray(x).astype(I32)


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


# ---------------------------------------------------------------------------------------------- :295-306 (data)
s_idct_row_table = [
  1,0,0,0,0,0,0,0, 2,0,0,0,0,0,0,0, 2,1,0,0,0,0,0,0, 2,1,1,0,0,0,0,0, 2,2,1,0,0,0,0,0, 3,2,1,0,0,0,0,0, 4,2,1,0,0,0,0,0, 4,3,1,0,0,0,0,0,
  4,3,2,0,0,0,0,0, 4,3,2,1,0,0,0,0, 4,3,2,1,1,0,0,0, 4,3,2,2,1,0,0,0, 4,3,3,2,1,0,0,0, 4,4,3,2,1,0,0,0, 5,4,3,2,1,0,0,0, 6,4,3,2,1,0,0,0,
  6,5,3,2,1,0,0,0, 6,5,4,2,1,0,0,0, 6,5,4,3,1,0,0,0, 6,5,4,3,2,0,0,0, 6,5,4,3,2,1,0,0, 6,5,4,3,2,1,1,0, 6,5,4,3,2,2,1,0, 6,5,4,3,3,2,1,0,
  6,5,4,4,3,2,1,0, 6,5,5,4,3,2,1,0, 6,6,5,4,3,2,1,0, 7,6,5,4,3,2,1,0, 8,6,5,4,3,2,1,0, 8,7,5,4,3,2,1,0, 8,7,6,4,3,2,1,0, 8,7,6,5,3,2,1,0,
  8,7,6,5,4,2,1,0, 8,7,6,5,4,3,1,0, 8,7,6,5,4,3,2,0, 8,7,6,5,4,3,2,1, 8,7,6,5,4,3,2,2, 8,7,6,5,4,3,3,2, 8,7,6,5,4,4,3,2, 8,7,6,5,5,4,3,2,
  8,7,6,6,5,4,3,2, 8,7,7,6,5,4,3,2, 8,8,7,6,5,4,3,2, 8,8,8,6,5,4,3,2, 8,8,8,7,5,4,3,2, 8,8,8,7,6,4,3,2, 8,8,8,7,6,5,3,2, 8,8,8,7,6,5,4,2,
  8,8,8,7,6,5,4,3, 8,8,8,7,6,5,4,4, 8,8,8,7,6,5,5,4, 8,8,8,7,6,6,5,4, 8,8,8,7,7,6,5,4, 8,8,8,8,7,6,5,4, 8,8,8,8,8,6,5,4, 8,8,8,8,8,7,5,4,
  8,8,8,8,8,7,6,4, 8,8,8,8,8,7,6,5, 8,8,8,8,8,7,6,6, 8,8,8,8,8,7,7,6, 8,8,8,8,8,8,7,6, 8,8,8,8,8,8,8,6, 8,8,8,8,8,8,8,7, 8,8,8,8,8,8,8,8,
]
s_idct_col_table = [ 1, 1, 2, 3, 3, 3, 3, 3, 3, 4, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 6, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8 ]
assert len(s_idct_row_table) == 512 and len(s_idct_col_table) == 64


# ---------------------------------------------------------------------------------------------- :308-376
def idct(pSrc_ptr, block_max_zag):
    """pSrc_ptr: (N, 64) int16, ONE block_max_zag for all N blocks -> (N, 64) uint8"""
    assert 1 <= block_max_zag <= 64
    N = pSrc_ptr.shape[0]
    pDst = np.zeros((N, 64), np.uint8)
    if block_max_zag <= 1:
        k = i32(((pSrc_ptr[:, 0].astype(I32) + I32(4)) >> I32(3)) + I32(128))
        k = CLAMP(k)
        for r in range(64):                                   # k | k<<8 | k<<16 | k<<24 written to all 8 rows
            pDst[:, r] = k
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
        if m is not None:
            self.v = [[m.v[r][c].copy() for c in range(4)] for r in range(4)]
        else:
            self.v = [[np.zeros(N, I32) for _ in range(4)] for _ in range(4)]

    def at(self, r, c):
        return self.v[r][c]

    def set(self, r, c, x):                                   # `M.at(r, c) = x` (a literal 0 broadcasts over the N blocks)
        self.v[r][c] = np.broadcast_to(i32(x), self.v[r][c].shape).copy()

    def __iadd__(self, a):
        for r in range(self.NUM_ROWS):
            self.v[r][0] = i32(self.at(r, 0) + a.at(r, 0))
            self.v[r][1] = i32(self.at(r, 1) + a.at(r, 1))
            self.v[r][2] = i32(self.at(r, 2) + a.at(r, 2))
            self.v[r][3] = i32(self.at(r, 3) + a.at(r, 3))
        return self

    def __isub__(self, a):
        for r in range(self.NUM_ROWS):
            self.v[r][0] = i32(self.at(r, 0) - a.at(r, 0))
            self.v[r][1] = i32(self.at(r, 1) - a.at(r, 1))
            self.v[r][2] = i32(self.at(r, 2) - a.at(r, 2))
            self.v[r][3] = i32(self.at(r, 3) - a.at(r, 3))
        return self

    def __add__(a, b):
        ret = Matrix44(len(a.v[0][0]))
        for r in range(a.NUM_ROWS):
            ret.v[r][0] = i32(a.at(r, 0) + b.at(r, 0))
            ret.v[r][1] = i32(a.at(r, 1) + b.at(r, 1))
            ret.v[r][2] = i32(a.at(r, 2) + b.at(r, 2))
            ret.v[r][3] = i32(a.at(r, 3) + b.at(r, 3))
        return ret

    def __sub__(a, b):
        ret = Matrix44(len(a.v[0][0]))
        for r in range(a.NUM_ROWS):
            ret.v[r][0] = i32(a.at(r, 0) - b.at(r, 0))
            ret.v[r][1] = i32(a.at(r, 1) - b.at(r, 1))
            ret.v[r][2] = i32(a.at(r, 2) - b.at(r, 2))
            ret.v[r][3] = i32(a.at(r, 3) - b.at(r, 3))
        return ret

    @staticmethod
    def add_and_store(pDst, a, b):                           # pDst: (N, 64) int16; cast(jpgd_block_t) = astype(int16) (truncation)
        for r in range(4):
            pDst[:, 0 * 8 + r] = i32(a.at(r, 0) + b.at(r, 0)).astype(np.int16)
            pDst[:, 1 * 8 + r] = i32(a.at(r, 1) + b.at(r, 1)).astype(np.int16)
            pDst[:, 2 * 8 + r] = i32(a.at(r, 2) + b.at(r, 2)).astype(np.int16)
            pDst[:, 3 * 8 + r] = i32(a.at(r, 3) + b.at(r, 3)).astype(np.int16)

    @staticmethod
    def sub_and_store(pDst, a, b):
        for r in range(4):
            pDst[:, 0 * 8 + r] = i32(a.at(r, 0) - b.at(r, 0)).astype(np.int16)
            pDst[:, 1 * 8 + r] = i32(a.at(r, 1) - b.at(r, 1)).astype(np.int16)
            pDst[:, 2 * 8 + r] = i32(a.at(r, 2) - b.at(r, 2)).astype(np.int16)
            pDst[:, 3 * 8 + r] = i32(a.at(r, 3) - b.at(r, 3)).astype(np.int16)


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
s_max_rc = [
    17, 18, 34, 50, 50, 51, 52, 52, 52, 68, 84, 84, 84, 84, 85, 86, 86, 86, 86, 86,
    102, 118, 118, 118, 118, 118, 118, 119, 120, 120, 120, 120, 120, 120, 120, 136,
    136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136,
    136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136
]
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

    a = Matrix44(m=P + Q)
    P -= Q
    b = P
    c = Matrix44(m=R + S)
    R -= S
    d = R

    temps, samples = [], []
    temp_block = np.zeros((N, 64), np.int16)                  # jpgd_block_t[64] temp_block, zero-initialised, reused by all four
    Matrix44.add_and_store(temp_block, a, c)
    temps.append(temp_block.copy()); samples.append(idct_4x4(temp_block))
    Matrix44.sub_and_store(temp_block, a, c)
    temps.append(temp_block.copy()); samples.append(idct_4x4(temp_block))
    Matrix44.add_and_store(temp_block, b, d)
    temps.append(temp_block.copy()); samples.append(idct_4x4(temp_block))
    Matrix44.sub_and_store(temp_block, b, d)
    temps.append(temp_block.copy()); samples.append(idct_4x4(temp_block))
    return np.stack(temps), np.stack(samples)


# ---------------------------------------------------------------------------------------------- :2080-2094
SCALEBITS = 16
ONE_HALF = 1 << (SCALEBITS - 1)


def FIX(x):                        # enum FIX(float x) = (cast(int)((x) * (1L<<SCALEBITS) + 0.5f))
    return I32(int(np.float32(np.float32(x) * np.float32(1 << SCALEBITS)) + np.float32(0.5)))


def create_look_ups():
    m_crr = np.zeros(256, I32); m_cbb = np.zeros(256, I32); m_crg = np.zeros(256, I32); m_cbg = np.zeros(256, I32)
    for i in range(256):
        k = I32(i - 128)
        m_crr[i] = i32(FIX(1.40200) * k + I32(ONE_HALF)) >> I32(SCALEBITS)
        m_cbb[i] = i32(FIX(1.77200) * k + I32(ONE_HALF)) >> I32(SCALEBITS)
        m_crg[i] = i32((-FIX(0.71414)) * k)
        m_cbg[i] = i32((-FIX(0.34414)) * k + I32(ONE_HALF))
    return m_crr, m_cbb, m_crg, m_cbg


# ---------------------------------------------------------------------------------------------- :2731-2823
def _mm_packs_epi32(x):            # signed saturation int32 -> int16
    return np.clip(x, -32768, 32767).astype(np.int16)


def _mm_packus_epi16(x):           # unsigned saturation int16 -> uint8
    return np.clip(x, 0, 255).astype(np.uint8)


def _sse_pixels(mm_y, mm_cb, mm_cr):
    """the arithmetic of one loop body of expanded_convert (:2769-2816) on int32 lanes -> (lanes, 4) uint8 R,G,B,255"""
    mm_128 = I32(128)
    mm_crr = i32((mm_cr - mm_128) * FIX(1.40200))                           # _mm_mullo_epi32
    mm_crg = i32((mm_cr - mm_128) * (-FIX(0.71414)))
    mm_cbg = i32((mm_cb - mm_128) * (-FIX(0.34414)))
    mm_cbb = i32((mm_cb - mm_128) * FIX(1.77200))
    mm_ONE_HALF = I32(ONE_HALF)
    mm_crr = i32(mm_crr + mm_ONE_HALF)
    mm_cbg = i32(mm_cbg + mm_ONE_HALF)
    mm_cbb = i32(mm_cbb + mm_ONE_HALF)
    mm_crr = mm_crr >> I32(16)                                              # _mm_srai_epi32
    mm_cbb = mm_cbb >> I32(16)
    mm_crg = i32(mm_crg + mm_cbg) >> I32(16)
    mm_crr = i32(mm_crr + mm_y)
    mm_crg = i32(mm_crg + mm_y)
    mm_cbb = i32(mm_cbb + mm_y)
    # _MM_TRANSPOSE4_PS(A, B, C, D) with D = 255: lane p becomes (A[p], B[p], C[p], 255); then packs_epi32 + packus_epi16
    T = np.stack([mm_crr, mm_crg, mm_cbb, np.full(mm_crr.shape, 255, I32)], axis=-1)
    return _mm_packus_epi16(_mm_packs_epi32(T))


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
                mm_y = sample_buf[Py + Y_ofs + j: Py + Y_ofs + j + 4].astype(I32)      # loadu_si32 + two unpacklo with zero
                mm_cb = sample_buf[Py + Cb_ofs + j: Py + Cb_ofs + j + 4].astype(I32)
                mm_cr = sample_buf[Py + Cr_ofs + j: Py + Cr_ofs + j + 4].astype(I32)
                out.append(_sse_pixels(mm_y, mm_cb, mm_cr).reshape(16))
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
    px = _sse_pixels(yb[:, idx_blk, idx_in].astype(I32), cb[:, idx_blk, idx_in].astype(I32), cr[:, idx_blk, idx_in].astype(I32))   # (M, 16, 16, 4)
    return px.reshape(mc, mr, 16, 16, 4).transpose(0, 2, 1, 3, 4).reshape(height, width * 4)
