/*
 * oracle_jpeg.c -- CPU restatement of Gamut's JPEG block reconstruction
 * (jpgd v1.04 port) plus the baseline entropy-decode feeder it needs.
 * TEST INFRASTRUCTURE ONLY (see gamut_oracle.h).
 *
 * Follows /root/reference/source/gamut/codecs/jpegload.d:
 *   zig-zag :105-106; constants / DESCALE / CLAMP :120-152;
 *   Row!N / Col!N sparse 1-D IDCTs :156-292; dispatch tables :295-306;
 *   idct :308-376; idct_4x4 :378-397;
 *   DCT_Upsample (P_Q, R_S, Matrix44 stores) :827-1073; s_max_rc :2132-2137;
 *   transform_mcu :2120-2130; transform_mcu_expand :2139-2255;
 *   create_look_ups :2080-2094; H1V1/H2V1/H1V2/gray/expanded convert :2528-2823;
 *   decode_next_row (baseline entropy decode + dequantise) :2405-2525;
 *   markers :1160-1848; geometry :3038-3090, 3130-3268; driver :3720-3808.
 *
 * All integer arithmetic is 32-bit two's complement with wrap-around (D
 * semantics); it is written on uint32_t where C would otherwise be undefined.
 */
#include "gamut_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef int32_t i32;
typedef uint32_t u32;

/* wrap-around helpers (D int semantics) */
static inline i32 wadd(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
static inline i32 wsub(i32 a, i32 b) { return (i32)((u32)a - (u32)b); }
static inline i32 wmul(i32 a, i32 b) { return (i32)((u32)a * (u32)b); }
static inline i32 wshl(i32 a, int n) { return (i32)((u32)a << n); }

/* jpegload.d:105-106 */
static const int g_ZAG[64] = { 0,1,8,16,9,2,3,10,17,24,32,25,18,11,4,5,12,19,26,33,40,48,41,34,27,20,13,6,7,14,21,28,35,42,49,56,57,50,43,36,29,22,15,23,30,37,44,51,58,59,52,45,38,31,39,46,53,60,61,54,47,55,62,63 };

enum { CONST_BITS = 13, PASS1_BITS = 2 };
enum {
    FIX_0_298631336 = 2446,  FIX_0_390180644 = 3196,  FIX_0_541196100 = 4433,
    FIX_0_765366865 = 6270,  FIX_0_899976223 = 7373,  FIX_1_175875602 = 9633,
    FIX_1_501321110 = 12299, FIX_1_847759065 = 15137, FIX_1_961570560 = 16069,
    FIX_2_053119869 = 16819, FIX_2_562915447 = 20995, FIX_3_072711026 = 25172
};

static inline i32 DESCALE(i32 x, int n)           { return wadd(x, 1 << (n - 1)) >> n; }              /* :137-140 */
static inline i32 DESCALE_ZEROSHIFT(i32 x, int n) { return wadd(wadd(x, wshl(128, n)), 1 << (n - 1)) >> n; } /* :142-145 */
static inline uint8_t CLAMP(i32 i) { if (i < 0) i = 0; if (i > 255) i = 255; return (uint8_t)i; }      /* :147-152 */

/* The 1-D butterfly shared by Row!N and Col!N (:178-202 / :240-265). x[] are
 * the (possibly zero-substituted) inputs; y[] the 8 un-descaled outputs in the
 * order y0..y7. */
static inline void butterfly(const i32 x[8], i32 y[8])
{
    const i32 z2 = x[2], z3 = x[6];
    const i32 z1   = wmul(wadd(z2, z3), FIX_0_541196100);
    const i32 tmp2 = wadd(z1, wmul(z3, -FIX_1_847759065));
    const i32 tmp3 = wadd(z1, wmul(z2, FIX_0_765366865));
    const i32 tmp0 = wshl(wadd(x[0], x[4]), CONST_BITS);
    const i32 tmp1 = wshl(wsub(x[0], x[4]), CONST_BITS);
    const i32 tmp10 = wadd(tmp0, tmp3), tmp13 = wsub(tmp0, tmp3), tmp11 = wadd(tmp1, tmp2), tmp12 = wsub(tmp1, tmp2);
    const i32 atmp0 = x[7], atmp1 = x[5], atmp2 = x[3], atmp3 = x[1];
    const i32 bz1 = wadd(atmp0, atmp3), bz2 = wadd(atmp1, atmp2), bz3 = wadd(atmp0, atmp2), bz4 = wadd(atmp1, atmp3);
    const i32 bz5 = wmul(wadd(bz3, bz4), FIX_1_175875602);
    const i32 az1 = wmul(bz1, -FIX_0_899976223);
    const i32 az2 = wmul(bz2, -FIX_2_562915447);
    const i32 az3 = wadd(wmul(bz3, -FIX_1_961570560), bz5);
    const i32 az4 = wadd(wmul(bz4, -FIX_0_390180644), bz5);
    const i32 btmp0 = wadd(wadd(wmul(atmp0, FIX_0_298631336), az1), az3);
    const i32 btmp1 = wadd(wadd(wmul(atmp1, FIX_2_053119869), az2), az4);
    const i32 btmp2 = wadd(wadd(wmul(atmp2, FIX_3_072711026), az2), az3);
    const i32 btmp3 = wadd(wadd(wmul(atmp3, FIX_1_501321110), az1), az4);
    y[0] = wadd(tmp10, btmp3); y[7] = wsub(tmp10, btmp3);
    y[1] = wadd(tmp11, btmp2); y[6] = wsub(tmp11, btmp2);
    y[2] = wadd(tmp12, btmp1); y[5] = wsub(tmp12, btmp1);
    y[3] = wadd(tmp13, btmp0); y[4] = wsub(tmp13, btmp0);
}

/* Row!(N).idct :156-214 */
static void row_idct(int nonzero_cols, i32* pTemp, const int16_t* pSrc)
{
    if (nonzero_cols == 0) return;                 /* leaves pTemp untouched (never read: Col!N zero-substitutes) */
    if (nonzero_cols == 1) {
        const i32 dc = wshl(pSrc[0], PASS1_BITS);
        for (int i = 0; i < 8; ++i) pTemp[i] = dc;
        return;
    }
    i32 x[8], y[8];
    for (int i = 0; i < 8; ++i) x[i] = (i < nonzero_cols) ? (i32)pSrc[i] : 0;
    butterfly(x, y);
    for (int i = 0; i < 8; ++i) pTemp[i] = DESCALE(y[i], CONST_BITS - PASS1_BITS);
}

/* Col!(N).idct :218-292 */
static void col_idct(int nonzero_rows, uint8_t* pDst, const i32* pTemp)
{
    if (nonzero_rows == 1) {
        const uint8_t v = CLAMP(DESCALE_ZEROSHIFT(pTemp[0], PASS1_BITS + 3));
        for (int i = 0; i < 8; ++i) pDst[i * 8] = v;
        return;
    }
    i32 x[8], y[8];
    for (int i = 0; i < 8; ++i) x[i] = (i < nonzero_rows) ? pTemp[i * 8] : 0;
    butterfly(x, y);
    for (int i = 0; i < 8; ++i) pDst[i * 8] = CLAMP(DESCALE_ZEROSHIFT(y[i], CONST_BITS + PASS1_BITS + 3));
}

/* :295-306 */
static const uint8_t s_idct_row_table[512] = {
  1,0,0,0,0,0,0,0, 2,0,0,0,0,0,0,0, 2,1,0,0,0,0,0,0, 2,1,1,0,0,0,0,0, 2,2,1,0,0,0,0,0, 3,2,1,0,0,0,0,0, 4,2,1,0,0,0,0,0, 4,3,1,0,0,0,0,0,
  4,3,2,0,0,0,0,0, 4,3,2,1,0,0,0,0, 4,3,2,1,1,0,0,0, 4,3,2,2,1,0,0,0, 4,3,3,2,1,0,0,0, 4,4,3,2,1,0,0,0, 5,4,3,2,1,0,0,0, 6,4,3,2,1,0,0,0,
  6,5,3,2,1,0,0,0, 6,5,4,2,1,0,0,0, 6,5,4,3,1,0,0,0, 6,5,4,3,2,0,0,0, 6,5,4,3,2,1,0,0, 6,5,4,3,2,1,1,0, 6,5,4,3,2,2,1,0, 6,5,4,3,3,2,1,0,
  6,5,4,4,3,2,1,0, 6,5,5,4,3,2,1,0, 6,6,5,4,3,2,1,0, 7,6,5,4,3,2,1,0, 8,6,5,4,3,2,1,0, 8,7,5,4,3,2,1,0, 8,7,6,4,3,2,1,0, 8,7,6,5,3,2,1,0,
  8,7,6,5,4,2,1,0, 8,7,6,5,4,3,1,0, 8,7,6,5,4,3,2,0, 8,7,6,5,4,3,2,1, 8,7,6,5,4,3,2,2, 8,7,6,5,4,3,3,2, 8,7,6,5,4,4,3,2, 8,7,6,5,5,4,3,2,
  8,7,6,6,5,4,3,2, 8,7,7,6,5,4,3,2, 8,8,7,6,5,4,3,2, 8,8,8,6,5,4,3,2, 8,8,8,7,5,4,3,2, 8,8,8,7,6,4,3,2, 8,8,8,7,6,5,3,2, 8,8,8,7,6,5,4,2,
  8,8,8,7,6,5,4,3, 8,8,8,7,6,5,4,4, 8,8,8,7,6,5,5,4, 8,8,8,7,6,6,5,4, 8,8,8,7,7,6,5,4, 8,8,8,8,7,6,5,4, 8,8,8,8,8,6,5,4, 8,8,8,8,8,7,5,4,
  8,8,8,8,8,7,6,4, 8,8,8,8,8,7,6,5, 8,8,8,8,8,7,6,6, 8,8,8,8,8,7,7,6, 8,8,8,8,8,8,7,6, 8,8,8,8,8,8,8,6, 8,8,8,8,8,8,8,7, 8,8,8,8,8,8,8,8,
};
static const uint8_t s_idct_col_table[64] = { 1, 1, 2, 3, 3, 3, 3, 3, 3, 4, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 6, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8 };

/* :308-376 */
void orc_jpeg_idct(const int16_t* src, uint8_t* dst, int block_max_zag)
{
    if (block_max_zag <= 1) {
        const uint8_t k = CLAMP(wadd(wadd(src[0], 4) >> 3, 128));
        memset(dst, k, 64);
        return;
    }
    i32 temp[64];
    const uint8_t* row_tab = &s_idct_row_table[(block_max_zag - 1) * 8];
    for (int i = 0; i < 8; ++i)
        row_idct(row_tab[i], temp + i * 8, src + i * 8);
    const int nonzero_rows = s_idct_col_table[block_max_zag - 1];
    for (int i = 0; i < 8; ++i)
        col_idct(nonzero_rows, dst + i, temp + i);
}

/* :378-397 */
void orc_jpeg_idct_4x4(const int16_t* src, uint8_t* dst)
{
    i32 temp[64];
    for (int i = 0; i < 4; ++i) row_idct(4, temp + i * 8, src + i * 8);
    for (int i = 0; i < 8; ++i) col_idct(4, dst + i, temp + i);
}

/* TEST-ONLY: libjpeg pass order (columns, then rows); same butterfly and
 * descales.  Used to pin the arithmetic against libjpeg-turbo (Pillow). */
void orc_jpeg_idct_colfirst(const int16_t* src, uint8_t* dst)
{
    i32 temp[64], x[8], y[8];
    for (int c = 0; c < 8; ++c) {
        /* libjpeg's pass-1 AC-all-zero shortcut (jidctint.c) is algebraically
         * identical to the dense butterfly, so it is not restated */
        for (int r = 0; r < 8; ++r) x[r] = src[r * 8 + c];
        butterfly(x, y);
        for (int r = 0; r < 8; ++r) temp[r * 8 + c] = DESCALE(y[r], CONST_BITS - PASS1_BITS);
    }
    for (int r = 0; r < 8; ++r) {
        for (int c = 0; c < 8; ++c) x[c] = temp[r * 8 + c];
        butterfly(x, y);
        for (int c = 0; c < 8; ++c) dst[r * 8 + c] = CLAMP(DESCALE_ZEROSHIFT(y[c], CONST_BITS + PASS1_BITS + 3));
    }
}

/* ---- frequency-domain chroma upsample, :827-1073 ------------------------- */
static inline i32 D10(i32 i) { return wadd(i, 512) >> 10; }            /* :910 */
#define FX(x) ((i32)((x) * 1024 + 0.5f))                                 /* :911, float arithmetic, C truncation */

/* the two 4-output maps of 8 inputs used by both stages */
static void map_E(const i32 u[8], i32 e[4])   /* coefficients of :929,:945 / :956,:958 */
{
    e[0] = u[0];
    e[1] = D10(wadd(wadd(wadd(wmul(FX(0.415735f), u[1]), wmul(FX(0.791065f), u[3])), wmul(FX(-0.352443f), u[5])), wmul(FX(0.277785f), u[7])));
    e[2] = u[4];
    e[3] = D10(wadd(wadd(wadd(wmul(FX(0.022887f), u[1]), wmul(FX(-0.097545f), u[3])), wmul(FX(0.490393f), u[5])), wmul(FX(0.865723f), u[7])));
}
static void map_O(const i32 u[8], i32 o[4])   /* coefficients of :1001,:1017 / :974,:976 */
{
    o[0] = D10(wadd(wadd(wadd(wmul(FX(0.906127f), u[1]), wmul(FX(-0.318190f), u[3])), wmul(FX(0.212608f), u[5])), wmul(FX(-0.180240f), u[7])));
    o[1] = u[2];
    o[2] = D10(wadd(wadd(wadd(wmul(FX(-0.074658f), u[1]), wmul(FX(0.513280f), u[3])), wmul(FX(0.768178f), u[5])), wmul(FX(-0.375330f), u[7])));
    o[3] = u[6];
}

/* :2132-2137 */
static const uint8_t s_max_rc[64] = {
    17, 18, 34, 50, 50, 51, 52, 52, 52, 68, 84, 84, 84, 84, 85, 86, 86, 86, 86, 86,
    102, 118, 118, 118, 118, 118, 118, 119, 120, 120, 120, 120, 120, 120, 120, 136,
    136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136,
    136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136
};

void orc_jpeg_upsample_block(const int16_t* pSrc, int block_max_zag, int16_t* out4)
{
    int max_zag = block_max_zag - 1;
    if (max_zag <= 0) max_zag = 0;
    if (max_zag > 63) max_zag = 63;
    const int NUM_ROWS = s_max_rc[max_zag] >> 4, NUM_COLS = s_max_rc[max_zag] & 15;   /* P_Q!(R,C) selection :2164-2228 */

    /* AT(c, r) :917-919 */
    #define AT(c, r) (((c) >= NUM_COLS || (r) >= NUM_ROWS) ? 0 : (i32)pSrc[(c) + (r) * 8])
    i32 X0[4][8], X1[4][8];     /* X0ij / X1ij of :921-952 and :1001-1032; i = output index, j = source row */
    for (int j = 0; j < 8; ++j) {
        i32 u[8], e[4], o[4];
        for (int c = 0; c < 8; ++c) u[c] = AT(c, j);
        map_E(u, e); map_O(u, o);
        for (int i = 0; i < 4; ++i) { X0[i][j] = e[i]; X1[i][j] = o[i]; }
    }
    #undef AT
    i32 P[4][4], Q[4][4], R[4][4], S[4][4];
    for (int i = 0; i < 4; ++i) {
        map_E(X0[i], P[i]);   /* :955-970 */
        map_O(X0[i], Q[i]);   /* :974-989 */
        map_E(X1[i], R[i]);   /* :1036-1051 */
        map_O(X1[i], S[i]);   /* :1054-1069 */
    }
    /* :2230-2251 with the transposed store of :886-902: pDst[c*8 + r] = a[r][c] (+/-) b[r][c] */
    memset(out4, 0, 4 * 64 * sizeof(int16_t));
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            const i32 a = wadd(P[r][c], Q[r][c]), b = wsub(P[r][c], Q[r][c]);
            const i32 cc = wadd(R[r][c], S[r][c]), d = wsub(R[r][c], S[r][c]);
            out4[0 * 64 + c * 8 + r] = (int16_t)wadd(a, cc);
            out4[1 * 64 + c * 8 + r] = (int16_t)wsub(a, cc);
            out4[2 * 64 + c * 8 + r] = (int16_t)wadd(b, d);
            out4[3 * 64 + c * 8 + r] = (int16_t)wsub(b, d);
        }
}

/* ---- colour ( :2080-2094 ) ------------------------------------------------ */
#define SCALEBITS 16
#define ONE_HALF  ((i32)1 << (SCALEBITS - 1))
#define CFIX(x)   ((i32)((x) * 65536.0f + 0.5f))     /* :2082, float arithmetic */

typedef struct { i32 crr[256], cbb[256], crg[256], cbg[256]; } lookups;
static void create_look_ups(lookups* t)
{
    for (int i = 0; i <= 255; i++) {
        const int k = i - 128;
        t->crr[i] = (CFIX(1.40200f) * k + ONE_HALF) >> SCALEBITS;
        t->cbb[i] = (CFIX(1.77200f) * k + ONE_HALF) >> SCALEBITS;
        t->crg[i] = (-CFIX(0.71414f)) * k;
        t->cbg[i] = (-CFIX(0.34414f)) * k + ONE_HALF;
    }
}
static inline void ycc_to_rgba(const lookups* t, int y, int cb, int cr, uint8_t* d)
{
    d[0] = CLAMP(y + t->crr[cr]);
    d[1] = CLAMP(y + ((t->crg[cr] + t->cbg[cb]) >> 16));
    d[2] = CLAMP(y + t->cbb[cb]);
    d[3] = 255;
}

/* ---- frame geometry ------------------------------------------------------ */
static int frame_geometry(orc_jpeg_frame* f, const int h_samp[3], const int v_samp[3])
{
    /* init_frame :3130-3195 */
    if (f->comps == 1) {
        if (h_samp[0] != 1 || v_samp[0] != 1) return -1;
        f->scan_type = ORC_JPGD_GRAYSCALE; f->blocks_per_mcu = 1;
    } else if (f->comps == 3) {
        if (h_samp[1] != 1 || v_samp[1] != 1 || h_samp[2] != 1 || v_samp[2] != 1) return -1;
        if      (h_samp[0] == 1 && v_samp[0] == 1) { f->scan_type = ORC_JPGD_YH1V1; f->blocks_per_mcu = 3; }
        else if (h_samp[0] == 2 && v_samp[0] == 1) { f->scan_type = ORC_JPGD_YH2V1; f->blocks_per_mcu = 4; }
        else if (h_samp[0] == 1 && v_samp[0] == 2) { f->scan_type = ORC_JPGD_YH1V2; f->blocks_per_mcu = 4; }
        else if (h_samp[0] == 2 && v_samp[0] == 2) { f->scan_type = ORC_JPGD_YH2V2; f->blocks_per_mcu = 6; }
        else return -1;
    } else return -1;
    /* :3197-3198 (== calc_mcu_block_order :3064-3065 for these sampling modes) */
    const int mx = (f->scan_type == ORC_JPGD_YH2V1 || f->scan_type == ORC_JPGD_YH2V2) ? 16 : 8;
    const int my = (f->scan_type == ORC_JPGD_YH1V2 || f->scan_type == ORC_JPGD_YH2V2) ? 16 : 8;
    f->mcus_per_row = (f->width + mx - 1) / mx;
    f->mcus_per_col = (f->height + my - 1) / my;
    return 0;
}

/* ---- entropy decoder (feeder): baseline and progressive ------------------- */
typedef struct {
    int      present;
    uint8_t  num[17];
    uint8_t  val[256];
    /* canonical decode tables (ITU T.81 Annex F.2.2.3) */
    i32      mincode[17], maxcode[18], valptr[17];
    uint16_t look[512];      /* 9-bit lookahead: (len << 8) | symbol, 0 = miss */
} hufftab;

static int huff_build(hufftab* h)      /* 0 = ok, -1 = the length counts over-subscribe the code space */
{
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) { code += h->num[l]; if (code > (1 << l)) return -1; code <<= 1; }
    code = 0;
    for (int l = 1; l <= 16; ++l) {
        h->valptr[l] = k;
        h->mincode[l] = code;
        code += h->num[l];
        k += h->num[l];
        h->maxcode[l] = h->num[l] ? code - 1 : -1;
        code <<= 1;
    }
    h->maxcode[17] = 0x7fffffff;
    memset(h->look, 0, sizeof(h->look));
    code = 0; k = 0;
    for (int l = 1; l <= 9; ++l) {
        for (int i = 0; i < h->num[l]; ++i, ++k, ++code) {
            const int shift = 9 - l;
            for (int fill = 0; fill < (1 << shift); ++fill)
                h->look[(code << shift) | fill] = (uint16_t)((l << 8) | h->val[k]);
        }
        code <<= 1;
    }
    return 0;
}

typedef struct {
    const uint8_t* p; const uint8_t* end;
    u32 bitbuf; int bits;     /* MSB-first, `bits` valid bits in the low end */
    int hit_marker;
    const uint8_t* seg;       /* where the reader was (re)started, and the bits taken since: process_restart needs the REFERENCE's read position */
    uint64_t used;
} bitreader;

static inline void br_fill(bitreader* b)
{
    while (b->bits <= 24) {
        u32 c = 0;
        if (!b->hit_marker && b->p < b->end) {
            c = *b->p;
            if (c == 0xFF) {
                if (b->p + 1 < b->end && b->p[1] == 0x00) { b->p += 2; }
                else { b->hit_marker = 1; c = 0xFF; }   /* marker: feed 1-bits like get_octet :683-696 */
            } else b->p++;
        } else c = 0xFF;
        b->bitbuf = (b->bitbuf << 8) | c;
        b->bits += 8;
    }
}
static inline u32 br_peek(bitreader* b, int n) { return (b->bitbuf >> (b->bits - n)) & ((1u << n) - 1); }
static inline void br_skip(bitreader* b, int n) { b->bits -= n; b->used += (uint64_t)n; }
static inline u32 br_get(bitreader* b, int n)
{
    if (!n) return 0;
    br_fill(b);
    const u32 v = br_peek(b, n); br_skip(b, n); return v;
}
static int huff_decode(bitreader* b, const hufftab* h)
{
    br_fill(b);
    const uint16_t e = h->look[br_peek(b, 9)];
    if (e) { br_skip(b, e >> 8); return e & 0xFF; }
    i32 code = (i32)br_peek(b, 9); int l = 9;
    br_skip(b, 9);
    while (l < 17 && code > h->maxcode[l]) { code = (code << 1) | (i32)br_get(b, 1); ++l; }
    if (l > 16) return -1;
    return h->val[h->valptr[l] + code - h->mincode[l]];
}
/* JPGD_HUFF_EXTEND :816-822 */
static inline int huff_extend(int x, int s) { return (s && x < (1 << (s - 1))) ? x + (int)(((u32)-1) << s) + 1 : x; }

static inline int rd16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

/* decoder state across markers and scans (the members of jpeg_decoder the feeder needs, :401-520) */
typedef struct {
    const uint8_t* data; size_t len, pos;
    int16_t  quant[4][64]; int quant_present[4];
    hufftab* huff;                                       /* 0-3 DC, 4-7 AC (:1247) */
    int comp_id[3], h_samp[3], v_samp[3], comp_quant[3], comp_dc[3], comp_ac[3];
    int restart_interval, have_sof, progressive;
    int comps_in_scan, comp_list[3], spectral_start, spectral_end, successive_low, successive_high;   /* read_sos_marker */
} jstate;

/* process_markers (:1578-1848) up to and including the next SOS (read_sos_marker :1466-1540).
   Returns 0xDA with S->pos at the first entropy-coded byte, 0xD9 at EOI / end of data, -1 on error. */
static int scan_header(jstate* S, orc_jpeg_frame* f)
{
    const uint8_t* data = S->data; const size_t len = S->len;
    for (;;) {
        /* next_marker :1544-1572 */
        while (S->pos < len && data[S->pos] != 0xFF) S->pos++;
        while (S->pos < len && data[S->pos] == 0xFF) S->pos++;
        if (S->pos >= len) return 0xD9;
        const int m = data[S->pos++];
        if (m == 0) continue;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return 0xD9;
        if (S->pos + 2 > len) return -1;
        const int seglen = rd16(data + S->pos);
        if (seglen < 2 || S->pos + (size_t)seglen > len) return -1;
        const uint8_t* s = data + S->pos + 2; int n = seglen - 2;
        if (m == 0xDB) {                                 /* DQT :1274-1346 */
            while (n > 0) {
                const int pq = s[0] >> 4, tq = s[0] & 15; s++; n--;
                if (tq >= 4 || n < (pq ? 128 : 64)) return -1;
                for (int i = 0; i < 64; ++i) {
                    u32 t = *s++;
                    if (pq) t = (t << 8) + *s++;
                    S->quant[tq][i] = (int16_t)t;         /* jpgd_quant_t = short :409,1328 */
                }
                n -= pq ? 128 : 64; S->quant_present[tq] = 1;
            }
        } else if (m == 0xC4) {                          /* DHT :1173-1270 */
            while (n > 0) {
                int index = s[0]; s++; n--;
                if (n < 16) return -1;
                index = (index & 0x0F) + ((index & 0x10) >> 4) * 4;
                if (index >= 8) return -1;
                hufftab* h = &S->huff[index];
                int count = 0; h->num[0] = 0;
                for (int i = 1; i <= 16; ++i) { h->num[i] = s[i - 1]; count += s[i - 1]; }
                s += 16; n -= 16;
                if (count > 255 || n < count) return -1;
                memset(h->val, 0, 256); memcpy(h->val, s, (size_t)count);
                s += count; n -= count;
                if (huff_build(h)) return -1;
                h->present = 1;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) { /* SOF0/SOF1/SOF2 :1349-1417, :1596-1607 */
            if (S->have_sof) return -1;
            if (n < 6 || s[0] != 8) return -1;
            f->height = rd16(s + 1); f->width = rd16(s + 3); f->comps = s[5];
            if (f->height < 1 || f->height > 16384 || f->width < 1 || f->width > 16384) return -1;
            if ((f->comps != 1 && f->comps != 3) || n != f->comps * 3 + 6) return -1;
            for (int i = 0; i < f->comps; ++i) {
                S->comp_id[i] = s[6 + 3*i]; S->h_samp[i] = s[7 + 3*i] >> 4; S->v_samp[i] = s[7 + 3*i] & 15; S->comp_quant[i] = s[8 + 3*i];
                if (S->comp_quant[i] >= 4) return -1;
            }
            if (frame_geometry(f, S->h_samp, S->v_samp)) return -1;
            S->have_sof = 1; S->progressive = (m == 0xC2);
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8) {
            return -1;                                   /* lossless / hierarchical / arithmetic: rejected like :1608-1628 */
        } else if (m == 0xDD) {                          /* DRI :1445-1462 */
            if (seglen != 4) return -1;
            S->restart_interval = rd16(s);
        } else if (m == 0xE0) {                          /* APP0 JFIF density */
            if (n >= 14 && !memcmp(s, "JFIF\0", 5)) {
                /* :1636-1672 */
                const int unit = s[7], xd = rd16(s + 8), yd = rd16(s + 10);
                f->pixel_aspect_ratio = (float)(xd / (double)yd);
                if (unit == 0) f->dpi_y = -1;
                else if (unit == 1) f->dpi_y = (float)yd;
                else if (unit == 2) f->dpi_y = (yd * 100.0f) / 39.37007874f;
            }
        } else if (m == 0xDA) {                          /* SOS :1466-1540 */
            if (!S->have_sof || n < 1) return -1;
            const int ns = s[0];
            if (ns < 1 || ns > f->comps || n != ns * 2 + 4) return -1;
            for (int i = 0; i < ns; ++i) {
                int ci; for (ci = 0; ci < f->comps; ++ci) if (s[1 + 2*i] == S->comp_id[ci]) break;
                if (ci >= f->comps) return -1;
                S->comp_list[i] = ci;
                S->comp_dc[ci] = (s[2 + 2*i] >> 4) & 15; S->comp_ac[ci] = (s[2 + 2*i] & 15) + 4;
                if (S->comp_dc[ci] >= 4 || S->comp_ac[ci] >= 8) return -1;
            }
            S->comps_in_scan = ns;
            S->spectral_start = s[1 + 2*ns]; S->spectral_end = s[2 + 2*ns];
            S->successive_high = s[3 + 2*ns] >> 4; S->successive_low = s[3 + 2*ns] & 15;
            if (!S->progressive) { S->spectral_start = 0; S->spectral_end = 63; }
            S->pos += (size_t)seglen;
            return 0xDA;
        }
        S->pos += (size_t)seglen;
    }
}

/* process_restart :2335-2402, statement by statement.  It reads RAW bytes (get_char :631-652) from where the bit reader's input stands: the
   reference keeps 16 .. 32 bits buffered and fetches two octets per refill (get_bits_no_markers :722-743; four at a (re)start :2390-2397 /
   init_scan), an octet being a data byte or an FF 00 pair, and never steps over a marker (get_octet :683-696 puts it back) -- so after `used`
   bits it stands 4 + 2 * (used / 16) octets behind the (re)start, or at the marker that stopped it.  From there: up to 1536 bytes to the next
   0xFF, the 0xFF fill bytes behind it, and the byte after them must be the expected RSTn -- anything else (a stray marker, stuffed data left
   over before the marker, more than 1536 bytes of it) is JPGD_BAD_RESTART_MARKER.  Past the end of the file get_char pads with FF D9 FF D9. */
static int restart(bitreader* br, int* next_restart)
{
    const uint8_t* q = br->seg;
    for (uint64_t n = 4 + 2 * (br->used / 16); n > 0 && q < br->end; --n) {
        if (*q == 0xFF) { if (q + 1 < br->end && q[1] == 0x00) q += 2; else break; }
        else q++;
    }
    int tem = 0, i, c = 0;
#define ORC_GET_CHAR() (q < br->end ? (int)*q++ : ((tem ^= 1) ? 0xFF : 0xD9))
    for (i = 1536; i > 0; i--) if (ORC_GET_CHAR() == 0xFF) break;
    if (i == 0) return -1;
    for (; i > 0; i--) { c = ORC_GET_CHAR(); if (c != 0xFF) break; }
#undef ORC_GET_CHAR
    if (i == 0) return -1;
    if (c != 0xD0 + *next_restart) return -1;
    br->p = q; br->bitbuf = 0; br->bits = 0; br->hit_marker = 0; br->seg = q; br->used = 0;
    *next_restart = (*next_restart + 1) & 7;
    return 0;
}

/* ---- progressive frames (:3296-3683) ------------------------------------------------------------------------
   One 64-coefficient block per (component, block_x, block_y) in natural order: the reference keeps DC in a 1x1
   coeff_buf and AC in an 8x8 one and merges them in load_next_row (:2280-2284); AC scans never touch index 0
   (spectral_start >= 1, :3633-3646), so a single block holds both. */
typedef struct { int16_t* blk; int bw, bh; } cplane;        /* coeff_buf :3274-3294, block dims = max_mcus * samp (:3601-3604) */

typedef struct { jstate* S; bitreader br; cplane pl[3]; u32 last_dc[3]; int eob_run; } pstate;

static int dc_first(pstate* P, int c, int16_t* p)            /* decode_block_dc_first :3298-3319 */
{
    int s = huff_decode(&P->br, &P->S->huff[P->S->comp_dc[c]]);
    if (s < 0) return -1;
    if (s != 0) { const int r = (int)br_get(&P->br, s & 15); s = huff_extend(r, s & 15); }
    P->last_dc[c] = (u32)(s += (int)P->last_dc[c]);
    p[0] = (int16_t)wshl(s, P->S->successive_low);
    return 0;
}
static int dc_refine(pstate* P, int c, int16_t* p)           /* decode_block_dc_refine :3321-3333 */
{
    (void)c;
    if (br_get(&P->br, 1)) p[0] = (int16_t)(p[0] | (1 << P->S->successive_low));
    return 0;
}
static int ac_first(pstate* P, int c, int16_t* p)            /* decode_block_ac_first :3335-3398 */
{
    if (P->eob_run) { P->eob_run--; return 0; }
    for (int k = P->S->spectral_start; k <= P->S->spectral_end; k++) {
        int s = huff_decode(&P->br, &P->S->huff[P->S->comp_ac[c]]);
        if (s < 0) return -1;
        int r = s >> 4; s &= 15;
        if (s) {
            if ((k += r) > 63) return -1;
            r = (int)br_get(&P->br, s);
            s = huff_extend(r, s);
            p[g_ZAG[k]] = (int16_t)wshl(s, P->S->successive_low);
        } else if (r == 15) {
            if ((k += 15) > 63) return -1;
        } else {
            P->eob_run = 1 << r;
            if (r) P->eob_run += (int)br_get(&P->br, r);
            P->eob_run--;
            break;
        }
    }
    return 0;
}
static void refine_nonzero(pstate* P, int16_t* coef, int p1, int m1)   /* the correction-bit step shared by :3457-3472 and :3498-3511 */
{
    if (br_get(&P->br, 1)) {
        if ((*coef & p1) == 0) *coef = (int16_t)(*coef >= 0 ? *coef + p1 : *coef + m1);
    }
}
static int ac_refine(pstate* P, int c, int16_t* p)           /* decode_block_ac_refine :3400-3518 */
{
    const int p1 = 1 << P->S->successive_low, m1 = (int)(((u32)-1) << P->S->successive_low);
    int k = P->S->spectral_start;
    if (P->eob_run == 0) {
        for (; k <= P->S->spectral_end; k++) {
            int s = huff_decode(&P->br, &P->S->huff[P->S->comp_ac[c]]);
            if (s < 0) return -1;
            int r = s >> 4; s &= 15;
            if (s) {
                if (s != 1) return -1;
                s = br_get(&P->br, 1) ? p1 : m1;
            } else if (r != 15) {
                P->eob_run = 1 << r;
                if (r) P->eob_run += (int)br_get(&P->br, r);
                break;
            }
            do {
                int16_t* coef = p + g_ZAG[k & 63];
                if (*coef != 0) refine_nonzero(P, coef, p1, m1);
                else if (--r < 0) break;
                k++;
            } while (k <= P->S->spectral_end);
            if (s && k < 64) p[g_ZAG[k]] = (int16_t)s;
        }
    }
    if (P->eob_run > 0) {
        for (; k <= P->S->spectral_end; k++) {
            int16_t* coef = p + g_ZAG[k & 63];
            if (*coef != 0) refine_nonzero(P, coef, p1, m1);
        }
        P->eob_run--;
    }
    return 0;
}

/* decode_scan :3521-3582 with calc_mcu_block_order's scan geometry (:3038-3090) */
static int progressive_scan(pstate* P, const orc_jpeg_frame* f, int (*fn)(pstate*, int, int16_t*))
{
    const jstate* S = P->S;
    int max_h = 0, max_v = 0, mcu_org[6], nb = 0, mcus_per_row, mcus_per_col;
    for (int c = 0; c < f->comps; ++c) { if (S->h_samp[c] > max_h) max_h = S->h_samp[c]; if (S->v_samp[c] > max_v) max_v = S->v_samp[c]; }
    if (S->comps_in_scan == 1) {
        const int c = S->comp_list[0];
        mcus_per_row = (((f->width  * S->h_samp[c]) + (max_h - 1)) / max_h + 7) / 8;    /* m_comp_h_blocks :3054 */
        mcus_per_col = (((f->height * S->v_samp[c]) + (max_v - 1)) / max_v + 7) / 8;
        mcu_org[nb++] = c;
    } else {
        mcus_per_row = (((f->width  + 7) / 8) + (max_h - 1)) / max_h;
        mcus_per_col = (((f->height + 7) / 8) + (max_v - 1)) / max_v;
        /* a component listed twice: decode_scan (:3520-3583) steps its block_x_mcu / m_block_y_mcu twice per MCU, so the walk below leaves the
           component's plane inside the first MCU row (coeff_buf_getp's assert :3293) whatever the geometry -- and m_mcu_org (10 entries) may not
           even hold the list.  Rejected here, before the list is laid out. */
        for (int i = 0; i < S->comps_in_scan; ++i) for (int j = i + 1; j < S->comps_in_scan; ++j) if (S->comp_list[i] == S->comp_list[j]) return -1;
        for (int i = 0; i < S->comps_in_scan; ++i) { const int c = S->comp_list[i]; for (int k = 0; k < S->h_samp[c] * S->v_samp[c]; ++k) mcu_org[nb++] = c; }
    }
    int restarts_left = S->restart_interval, next_restart = 0;
    int block_y_mcu[3] = {0, 0, 0};
    for (int mcu_col = 0; mcu_col < mcus_per_col; ++mcu_col) {
        int block_x_mcu[3] = {0, 0, 0};
        for (int mcu_row = 0; mcu_row < mcus_per_row; ++mcu_row) {
            int xo = 0, yo = 0;
            if (S->restart_interval && restarts_left == 0) {
                if (restart(&P->br, &next_restart)) return -1;
                P->last_dc[0] = P->last_dc[1] = P->last_dc[2] = 0; P->eob_run = 0;
                restarts_left = S->restart_interval;
            }
            for (int b = 0; b < nb; ++b) {
                const int c = mcu_org[b];
                const int bx = block_x_mcu[c] + xo, by = block_y_mcu[c] + yo;
                if (bx >= P->pl[c].bw || by >= P->pl[c].bh) return -1;       /* coeff_buf_getp asserts :3293 */
                if (fn(P, c, P->pl[c].blk + ((size_t)by * P->pl[c].bw + bx) * 64)) return -1;
                if (S->comps_in_scan == 1) block_x_mcu[c]++;
                else if (++xo == S->h_samp[c]) { xo = 0; if (++yo == S->v_samp[c]) { yo = 0; block_x_mcu[c] += S->h_samp[c]; } }
            }
            restarts_left--;
        }
        if (S->comps_in_scan == 1) block_y_mcu[S->comp_list[0]]++;
        else for (int i = 0; i < S->comps_in_scan; ++i) block_y_mcu[S->comp_list[i]] += S->v_samp[S->comp_list[i]];
    }
    return 0;
}

int orc_jpeg_decode_coeffs(const uint8_t* data, size_t len, orc_jpeg_frame* f)
{
    memset(f, 0, sizeof(*f));
    f->pixel_aspect_ratio = -1; f->dpi_y = -1;
    if (len < 4 || data[0] != 0xFF || data[1] != 0xD8) return -1;

    jstate* S = (jstate*)calloc(1, sizeof(jstate));
    if (!S) return -1;
    S->huff = (hufftab*)calloc(8, sizeof(hufftab));
    S->data = data; S->len = len; S->pos = 2;
    int rc = -1;
    pstate* P = NULL;
    if (!S->huff) goto done;

    if (scan_header(S, f) != 0xDA) goto done;

    int mcu_org[6], nb = 0;                                        /* frame-interleaved block order :3076-3088 */
    if (f->comps == 1) mcu_org[nb++] = 0;
    else for (int c = 0; c < 3; ++c) for (int k = 0; k < S->h_samp[c] * S->v_samp[c]; ++k) mcu_org[nb++] = c;
    const size_t nmcu = (size_t)f->mcus_per_row * f->mcus_per_col;
    const size_t nblocks = nmcu * (size_t)nb;
    f->coeffs  = (int16_t*)calloc(nblocks * 64, sizeof(int16_t));
    f->max_zag = (uint8_t*)malloc(nblocks ? nblocks : 1);
    if (!f->coeffs || !f->max_zag) goto done;

    if (!S->progressive) {
        /* init_sequential :3666-3677: one interleaved scan carrying every component */
        if (S->comps_in_scan != f->comps) goto done;
        for (int i = 0; i < S->comps_in_scan; ++i) {                /* check tables :2990-3034: of the components the scan lists */
            const int c = S->comp_list[i];
            if (!S->quant_present[S->comp_quant[c]] || !S->huff[S->comp_dc[c]].present || !S->huff[S->comp_ac[c]].present) goto done;
        }
        /* calc_mcu_block_order :3068-3088: the MCU's blocks belong to the components in the order the SOS lists them (a conforming file: the
           frame's order).  Tables, quantisation and predictor of block b go by that list; everything behind the entropy decoder goes by the
           block's position.  A list with a component twice gives another number of blocks per MCU than init_frame (:3136-3260) sized the
           decoder's buffers for -- more overruns them, fewer leaves part of every MCU uninitialised: no result to restate, rejected. */
        if (f->comps > 1) {
            int n = 0;
            for (int i = 0; i < S->comps_in_scan; ++i) { const int c = S->comp_list[i]; for (int k = 0; k < S->h_samp[c] * S->v_samp[c]; ++k) { if (n < 6) mcu_org[n] = c; ++n; } }
            if (n != nb) goto done;
        }

        bitreader br = { data + S->pos, data + len, 0, 0, 0, data + S->pos, 0 };
        u32 last_dc[3] = {0,0,0};
        int restarts_left = S->restart_interval, next_restart = 0;
        int16_t* p = f->coeffs; uint8_t* mz = f->max_zag;

        for (size_t mcu = 0; mcu < nmcu; ++mcu) {
            if (S->restart_interval && restarts_left == 0) {       /* process_restart :2335-2402 */
                if (restart(&br, &next_restart)) goto done;
                last_dc[0] = last_dc[1] = last_dc[2] = 0;
                restarts_left = S->restart_interval;
            }
            for (int b = 0; b < nb; ++b, p += 64, ++mz) {       /* decode_next_row :2419-2515 (dense store: no stale data to clear) */
                const int c = mcu_org[b];
                const int16_t* q = S->quant[S->comp_quant[c]];
                int s = huff_decode(&br, &S->huff[S->comp_dc[c]]);
                if (s < 0) goto done;
                int r = (int)br_get(&br, s & 15);
                s = huff_extend(r, s & 15);
                last_dc[c] = (u32)(s += (int)last_dc[c]);
                p[0] = (int16_t)wmul(s, q[0]);
                int k;
                for (k = 1; k < 64; ++k) {
                    s = huff_decode(&br, &S->huff[S->comp_ac[c]]);
                    if (s < 0) goto done;
                    r = s >> 4; s &= 15;
                    if (s) {
                        if (r) { if (k + r > 63) goto done; k += r; }
                        const int extra = (int)br_get(&br, s);
                        s = huff_extend(extra, s);
                        p[g_ZAG[k]] = (int16_t)wmul(s, q[k]);
                    } else {
                        if (r == 15) { if (k + 16 > 64) goto done; k += 15; }
                        else break;
                    }
                }
                *mz = (uint8_t)k;                                /* :2512 */
            }
            restarts_left--;
        }
        rc = 0;
    } else {
        /* init_progressive :3585-3664: every scan into the coefficient planes, then load_next_row's hand-over */
        P = (pstate*)calloc(1, sizeof(pstate));
        if (!P) goto done;
        P->S = S;
        for (int c = 0; c < f->comps; ++c) {
            P->pl[c].bw = f->mcus_per_row * S->h_samp[c]; P->pl[c].bh = f->mcus_per_col * S->v_samp[c];
            P->pl[c].blk = (int16_t*)calloc((size_t)P->pl[c].bw * P->pl[c].bh * 64, sizeof(int16_t));
            if (!P->pl[c].blk) goto done;
        }
        for (int marker = 0xDA; marker == 0xDA; ) {
            const int dc_only = S->spectral_start == 0, refinement = S->successive_high != 0;
            if (S->spectral_start > S->spectral_end || S->spectral_end > 63) goto done;
            if (dc_only) { if (S->spectral_end) goto done; }
            else if (S->comps_in_scan != 1) goto done;             /* AC scans carry one component :3642-3646 */
            if (refinement && S->successive_low != S->successive_high - 1) goto done;
            for (int i = 0; i < S->comps_in_scan; ++i) {           /* check_huff_tables :3013-3034 (by scan kind), check_quant_tables */
                const int c = S->comp_list[i];
                if (!S->quant_present[S->comp_quant[c]]) goto done;
                if (dc_only ? (!refinement && !S->huff[S->comp_dc[c]].present) : !S->huff[S->comp_ac[c]].present) goto done;   /* DC refinement reads raw bits only */
            }
            P->br.p = data + S->pos; P->br.end = data + len; P->br.bitbuf = 0; P->br.bits = 0; P->br.hit_marker = 0; P->br.seg = P->br.p; P->br.used = 0;
            P->last_dc[0] = P->last_dc[1] = P->last_dc[2] = 0; P->eob_run = 0;                 /* init_scan :3108-3110 */
            if (progressive_scan(P, f, dc_only ? (refinement ? dc_refine : dc_first) : (refinement ? ac_refine : ac_first))) goto done;
            S->pos = (size_t)(P->br.p - data);                     /* the reader never steps over a marker */
            marker = scan_header(S, f);
            if (marker < 0) goto done;
        }
        /* load_next_row :2259-2333 for every MCU row: merge, find the last non-zero coefficient, de-quantise */
        int16_t* p = f->coeffs; uint8_t* mz = f->max_zag;
        for (int my = 0; my < f->mcus_per_col; ++my)
            for (int mx = 0; mx < f->mcus_per_row; ++mx) {
                int xo = 0, yo = 0, cprev = -1;
                for (int b = 0; b < nb; ++b, p += 64, ++mz) {
                    const int c = mcu_org[b];
                    if (c != cprev) { xo = yo = 0; cprev = c; }
                    const int bx = mx * S->h_samp[c] + xo, by = my * S->v_samp[c] + yo;
                    if (++xo == S->h_samp[c]) { xo = 0; ++yo; }
                    memcpy(p, P->pl[c].blk + ((size_t)by * P->pl[c].bw + bx) * 64, 128);
                    const int16_t* q = S->quant[S->comp_quant[c]];
                    int i;
                    for (i = 63; i > 0; i--) if (p[g_ZAG[i]]) break;
                    *mz = (uint8_t)(i + 1);
                    for (; i >= 0; i--) if (p[g_ZAG[i]]) p[g_ZAG[i]] = (int16_t)wmul(p[g_ZAG[i]], q[i]);
                }
            }
        rc = 0;
    }
done:
    if (P) { for (int c = 0; c < 3; ++c) free(P->pl[c].blk); free(P); }
    free(S->huff); free(S);
    if (rc) orc_jpeg_frame_free(f);
    return rc;
}

void orc_jpeg_frame_free(orc_jpeg_frame* f)
{
    free(f->coeffs); free(f->max_zag);
    f->coeffs = NULL; f->max_zag = NULL;
}

/* ---- reconstruction ------------------------------------------------------ */
int orc_jpeg_reconstruct(const orc_jpeg_frame* f, int req_comps, uint8_t* out, int out_pitch, int colfirst)
{
    if (req_comps != 1 && req_comps != 3 && req_comps != 4) return -1;
    if (!f->coeffs) return -1;
    lookups lut; create_look_ups(&lut);

    const int expand = (f->scan_type == ORC_JPGD_YH2V2);     /* m_freq_domain_chroma_upsample :3244-3247 */
    const int nb = f->blocks_per_mcu;
    const int sample_blocks = expand ? 12 : nb;               /* m_expanded_blocks_per_mcu :3240-3241 */
    const int mcu_x = (f->scan_type == ORC_JPGD_YH2V1 || f->scan_type == ORC_JPGD_YH2V2) ? 16 : 8;
    const int mcu_y = (f->scan_type == ORC_JPGD_YH1V2 || f->scan_type == ORC_JPGD_YH2V2) ? 16 : 8;
    const int bpp = (f->scan_type == ORC_JPGD_GRAYSCALE) ? 1 : 4;   /* :3201-3204 */
    const int line_px = f->mcus_per_row * mcu_x;

    uint8_t* sample_buf = (uint8_t*)malloc((size_t)f->mcus_per_row * sample_blocks * 64);  /* m_pSample_buf :3249-3260 */
    uint8_t* line = (uint8_t*)malloc((size_t)line_px * bpp + 64);                          /* m_pScan_line_0 */
    if (!sample_buf || !line) { free(sample_buf); free(line); return -1; }

    for (int my = 0; my < f->mcus_per_col; ++my) {
        /* transform one MCU row */
        for (int mx = 0; mx < f->mcus_per_row; ++mx) {
            const size_t blk0 = ((size_t)my * f->mcus_per_row + mx) * nb;
            const int16_t* src = f->coeffs + blk0 * 64;
            const uint8_t* mz  = f->max_zag ? f->max_zag + blk0 : NULL;
            uint8_t* dst = sample_buf + (size_t)mx * sample_blocks * 64;
            if (!expand) {                                           /* transform_mcu :2120-2130 */
                for (int b = 0; b < nb; ++b, src += 64, dst += 64) {
                    if (colfirst) orc_jpeg_idct_colfirst(src, dst);
                    else orc_jpeg_idct(src, dst, mz ? mz[b] : 64);
                }
            } else {                                                 /* transform_mcu_expand :2139-2255 */
                for (int b = 0; b < 4; ++b, src += 64, dst += 64) {
                    if (colfirst) orc_jpeg_idct_colfirst(src, dst);
                    else orc_jpeg_idct(src, dst, mz ? mz[b] : 64);
                }
                for (int i = 0; i < 2; ++i, src += 64) {
                    int16_t tmp[4 * 64];
                    orc_jpeg_upsample_block(src, mz ? mz[4 + i] : 64, tmp);
                    for (int q = 0; q < 4; ++q, dst += 64) orc_jpeg_idct_4x4(tmp + q * 64, dst);
                }
            }
        }
        /* emit the MCU row's scanlines */
        for (int row = 0; row < mcu_y; ++row) {
            const int y = my * mcu_y + row;
            if (y >= f->height) break;
            uint8_t* d = line;
            switch (f->scan_type) {
            case ORC_JPGD_GRAYSCALE: {                               /* gray_convert :2715-2728 */
                const uint8_t* s = sample_buf + row * 8;
                for (int i = 0; i < f->mcus_per_row; ++i, s += 64, d += 8) memcpy(d, s, 8);
            } break;
            case ORC_JPGD_YH1V1: {                                   /* H1V1Convert :2528-2555 */
                const uint8_t* s = sample_buf + row * 8;
                for (int i = 0; i < f->mcus_per_row; ++i, s += 64 * 3)
                    for (int j = 0; j < 8; ++j, d += 4) ycc_to_rgba(&lut, s[j], s[64 + j], s[128 + j], d);
            } break;
            case ORC_JPGD_YH2V1: {                                   /* H2V1Convert :2558-2600 */
                const uint8_t* yb = sample_buf + row * 8;
                const uint8_t* c  = sample_buf + 2 * 64 + row * 8;
                for (int i = 0; i < f->mcus_per_row; ++i) {
                    for (int l = 0; l < 2; ++l) {
                        for (int j = 0; j < 4; ++j, d += 8, ++c) {
                            ycc_to_rgba(&lut, yb[j << 1],       c[0], c[64], d);
                            ycc_to_rgba(&lut, yb[(j << 1) + 1], c[0], c[64], d + 4);
                        }
                        yb += 64;
                    }
                    yb += 64 * 4 - 64 * 2;
                    c  += 64 * 4 - 8;
                }
            } break;
            case ORC_JPGD_YH1V2: {                                   /* H1V2Convert :2603-2647 (one row at a time) */
                const uint8_t* yb = (row < 8) ? sample_buf + row * 8 : sample_buf + 64 + (row & 7) * 8;
                const uint8_t* c  = sample_buf + 64 * 2 + (row >> 1) * 8;
                for (int i = 0; i < f->mcus_per_row; ++i, yb += 64 * 4, c += 64 * 4)
                    for (int j = 0; j < 8; ++j, d += 4) ycc_to_rgba(&lut, yb[j], c[j], c[64 + j], d);
            } break;
            default: {                                               /* expanded_convert :2731-2823 */
                const uint8_t* Py = sample_buf + (row / 8) * 64 * 2 + (row & 7) * 8;
                for (int i = 0; i < f->mcus_per_row; ++i, Py += 64 * 12)
                    for (int k = 0; k < 16; k += 8) {
                        const int Y_ofs = k * 8, Cb_ofs = Y_ofs + 64 * 4, Cr_ofs = Y_ofs + 64 * 8;
                        for (int j = 0; j < 8; ++j, d += 4) {
                            /* SSE path of the reference, restated per pixel: packs_epi32 + packus_epi16 == clamp to [0,255] */
                            const i32 yy = Py[Y_ofs + j], cb = Py[Cb_ofs + j], cr = Py[Cr_ofs + j];
                            const i32 rr = (wmul(cr - 128, CFIX(1.40200f)) + ONE_HALF) >> 16;
                            const i32 gg = (wmul(cr - 128, -CFIX(0.71414f)) + (wmul(cb - 128, -CFIX(0.34414f)) + ONE_HALF)) >> 16;
                            const i32 bb = (wmul(cb - 128, CFIX(1.77200f)) + ONE_HALF) >> 16;
                            d[0] = CLAMP(rr + yy); d[1] = CLAMP(gg + yy); d[2] = CLAMP(bb + yy); d[3] = 255;
                        }
                    }
            } break;
            }
            /* output packing :3761-3801 */
            uint8_t* pDst = out + (size_t)y * (size_t)out_pitch;
            const uint8_t* sl = line;
            if ((req_comps == 1 && f->comps == 1) || (req_comps == 4 && f->comps == 3)) {
                memcpy(pDst, sl, (size_t)f->width * req_comps);
            } else if (f->comps == 1) {
                for (int x = 0; x < f->width; ++x) {
                    pDst[0] = pDst[1] = pDst[2] = sl[x];
                    if (req_comps == 4) pDst[3] = 255;
                    pDst += req_comps;
                }
            } else if (req_comps == 1) {
                for (int x = 0; x < f->width; ++x) {
                    const int r = sl[x*4], g = sl[x*4+1], b = sl[x*4+2];
                    *pDst++ = (uint8_t)((r * 19595 + g * 38470 + b * 7471 + 32768) >> 16);
                }
            } else {
                for (int x = 0; x < f->width; ++x) { pDst[0] = sl[x*4]; pDst[1] = sl[x*4+1]; pDst[2] = sl[x*4+2]; pDst += 3; }
            }
        }
    }
    free(sample_buf); free(line);
    return 0;
}

uint8_t* orc_decompress_jpeg_image_from_memory(const uint8_t* data, size_t len,
        int* width, int* height, int* actual_comps, float* pixelAspectRatio,
        float* dotsPerInchY, int req_comps)
{
    if (req_comps != -1 && req_comps != 1 && req_comps != 3 && req_comps != 4) return NULL;   /* :3727 */
    orc_jpeg_frame f;
    if (orc_jpeg_decode_coeffs(data, len, &f)) return NULL;
    *width = f.width; *height = f.height; *actual_comps = f.comps;
    if (req_comps < 0) req_comps = f.comps;
    const int dst_bpl = f.width * req_comps;
    uint8_t* img = (uint8_t*)malloc((size_t)dst_bpl * f.height);
    if (img && orc_jpeg_reconstruct(&f, req_comps, img, dst_bpl, 0)) { free(img); img = NULL; }
    if (pixelAspectRatio) *pixelAspectRatio = f.pixel_aspect_ratio;
    if (dotsPerInchY) *dotsPerInchY = f.dpi_y;
    orc_jpeg_frame_free(&f);
    return img;
}
