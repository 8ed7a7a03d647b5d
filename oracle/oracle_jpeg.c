/*
 * oracle_jpeg.c -- CPU restatement of Gamut's JPEG block reconstruction
 * (jpgd v1.04 port).
 * TEST INFRASTRUCTURE ONLY (see gamut_oracle.h).
 *
 * Follows /root/reference/source/gamut/codecs/jpegload.d:
 *   zig-zag :105-106; constants / DESCALE / CLAMP :120-152;
 *   Row!N / Col!N sparse 1-D IDCTs :156-292; dispatch tables :295-306;
 *   idct :308-376; idct_4x4 :378-397;
 *   DCT_Upsample (P_Q, R_S, Matrix44 stores) :827-1073; s_max_rc :2132-2137;
 *   transform_mcu :2120-2130; transform_mcu_expand :2139-2255;
 *   create_look_ups :2080-2094; H1V1/H2V1/H1V2/gray/expanded convert :2528-2823;
 *   driver's output packing :3753-3808 (the input layer -- markers, bit readers, Huffman, restarts, decode_next_row,
 *   progressive scans -- is oracle_jpeg_input.c).
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

/* The feeder (markers, bit reader, Huffman, restarts, baseline and progressive coefficient decoding, orc_jpeg_decode_coeffs /
   orc_jpeg_frame_free) is oracle_jpeg_input.c: jpgd's input layer with its state kept. */

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
