/*
 * oracle_jpeg_input.c -- CPU restatement of jpgd's INPUT LAYER as Gamut carries it: the buffered byte reader
 * with its pad areas and stuffed-back bytes, the two bit readers, the marker walk (tables, JFIF / EXIF density,
 * verdicts), make_huff_table's look-up / tree tables and the two huff_decode variants, restart handling,
 * baseline and progressive coefficient decoding, and the driver's order of calls.
 * TEST INFRASTRUCTURE ONLY (see gamut_oracle.h).
 *
 * Follows /root/reference/source/gamut/codecs/jpegload.d function for function, with the decoder's state KEPT
 * (m_in_buf and its neighbours, m_pIn_buf_ofs / m_in_buf_left, m_tem_flag, m_bit_buf / m_bits_left) rather than
 * modelled -- the member and function names are the reference's so the two texts can be read side by side:
 *   get_char / stuff_char / get_octet :631-696;  get_bits / get_bits_no_markers :699-743;  huff_decode x2 :746-813;
 *   JPGD_HUFF_EXTEND :816-822;  set_error :1097-1101;  prep_in_buffer :1149-1174;  read_dht / dqt / sof /
 *   skip_variable / dri / sos :1177-1543;  next_marker :1546-1573;  process_markers (APP0 JFIF, APP1 EXIF, RSTn /
 *   TEM / JPG) :1578-1848;  locate_soi / sof / sos :1854-1967;  initit :1971-2078;  fix_in_buffer :2098-2118;
 *   load_next_row :2259-2332;  process_restart :2335-2402;  decode_next_row :2405-2525;  find_eoi :2826-2848;
 *   make_huff_table :2851-2987;  check_quant / huff_tables :2990-3034;  calc_mcu_block_order :3038-3090;
 *   init_scan :3093-3125;  init_frame :3130-3268;  coeff_buf :3274-3295;  decode_block_* :3299-3518;
 *   decode_scan :3521-3584;  init_progressive / init_sequential / decode_start / decode_init :3587-3713;
 *   begin_decoding / decode :530-612;  decompress_jpeg_image_from_stream :3720-3808.
 *
 * set_error() leaves through longjmp: once m_error_code is set every later check of it (:3733, :532, :546) makes the
 * driver return null, whatever the decoder does in between.  A second exit, undefined(), is taken where the
 * reference would read or write outside an array, use memory nobody wrote, trip an assert or never return: there
 * is no result to restate, the oracle rejects the file and says which (orc_jpeg_decode_coeffs returns -2).
 *
 * Checked against a second reading of the same lines (tools/ref_literal_input.py, Python, written independently)
 * by tools/fuzz_input.py: verdict, geometry, every coefficient, every max_zag, pixelAspectRatio / dotsPerInchY.
 */
#include "gamut_oracle.h"
#include <math.h>
#include <setjmp.h>
#include <stdlib.h>
#include <string.h>

typedef int32_t i32;
typedef uint32_t u32;

enum { JPGD_IN_BUF_SIZE = 8192, JPGD_MAX_BLOCKS_PER_MCU = 10, JPGD_MAX_HUFF_TABLES = 8, JPGD_MAX_QUANT_TABLES = 4,
       JPGD_MAX_COMPONENTS = 4, JPGD_MAX_COMPS_IN_SCAN = 4, JPGD_MAX_BLOCKS_PER_ROW = 8192, JPGD_MAX_HEIGHT = 16384, JPGD_MAX_WIDTH = 16384 };
enum { JPGD_SUCCESS = 0, JPGD_FAILED = -1, JPGD_DONE = 1 };
enum { M_SOF0 = 0xC0, M_SOF1 = 0xC1, M_SOF2 = 0xC2, M_SOF3 = 0xC3, M_SOF5 = 0xC5, M_SOF6 = 0xC6, M_SOF7 = 0xC7, M_JPG = 0xC8,
       M_SOF9 = 0xC9, M_SOF10 = 0xCA, M_SOF11 = 0xCB, M_SOF13 = 0xCD, M_SOF14 = 0xCE, M_SOF15 = 0xCF, M_DHT = 0xC4, M_DAC = 0xCC,
       M_RST0 = 0xD0, M_RST7 = 0xD7, M_SOI = 0xD8, M_EOI = 0xD9, M_SOS = 0xDA, M_DQT = 0xDB, M_DRI = 0xDD, M_APP0 = 0xE0, M_TEM = 0x01 };

static const int g_ZAG[64] = { 0,1,8,16,9,2,3,10,17,24,32,25,18,11,4,5,12,19,26,33,40,48,41,34,27,20,13,6,7,14,21,28,35,42,49,56,57,50,43,36,29,22,15,23,30,37,44,51,58,59,52,45,38,31,39,46,53,60,61,54,47,55,62,63 };

typedef struct { int ac_table; u32 look_up[256], look_up2[256]; uint8_t code_size[256]; u32 tree[512]; } huff_tables;          /* :413-419 */
typedef struct { int16_t* pData; int block_num_x, block_num_y, block_len_x, block_len_y, block_size; } coeff_buf;               /* :421-426, block_size in words */

typedef struct {
    /* the stream: stream_read_jpeg (plugins/jpeg.d:158-169) over a memory file (io.d:428-471) */
    const uint8_t* src; size_t src_len, src_ofs;
    jmp_buf out; int undefined;

    int m_image_x_size, m_image_y_size, m_progressive_flag;
    uint8_t  m_huff_ac[JPGD_MAX_HUFF_TABLES];
    uint8_t* m_huff_num[JPGD_MAX_HUFF_TABLES];
    uint8_t* m_huff_val[JPGD_MAX_HUFF_TABLES];
    int16_t* m_quant[JPGD_MAX_QUANT_TABLES];
    int m_scan_type, m_comps_in_frame;
    int m_comp_h_samp[JPGD_MAX_COMPONENTS], m_comp_v_samp[JPGD_MAX_COMPONENTS], m_comp_quant[JPGD_MAX_COMPONENTS], m_comp_ident[JPGD_MAX_COMPONENTS];
    int m_comp_h_blocks[JPGD_MAX_COMPONENTS], m_comp_v_blocks[JPGD_MAX_COMPONENTS];
    int m_comps_in_scan, m_comp_list[JPGD_MAX_COMPS_IN_SCAN], m_comp_dc_tab[JPGD_MAX_COMPONENTS], m_comp_ac_tab[JPGD_MAX_COMPONENTS];
    int m_spectral_start, m_spectral_end, m_successive_low, m_successive_high;
    int m_max_mcu_x_size, m_max_mcu_y_size, m_blocks_per_mcu, m_max_blocks_per_row, m_mcus_per_row, m_mcus_per_col;
    int m_mcu_org[JPGD_MAX_BLOCKS_PER_MCU];
    int m_total_lines_left, m_mcu_lines_left;
    huff_tables* m_pHuff_tabs[JPGD_MAX_HUFF_TABLES];
    coeff_buf* m_dc_coeffs[JPGD_MAX_COMPONENTS]; coeff_buf* m_ac_coeffs[JPGD_MAX_COMPONENTS];
    int m_eob_run, m_block_y_mcu[JPGD_MAX_COMPONENTS];
    uint8_t* m_pIn_buf_ofs; int m_in_buf_left, m_tem_flag, m_eof_flag;
    uint8_t m_in_buf_pad_start[128], m_in_buf[JPGD_IN_BUF_SIZE + 128], m_in_buf_pad_end[128];      /* consecutive members, as at :481-483 */
    int m_bits_left; u32 m_bit_buf;
    int m_restart_interval, m_restarts_left, m_next_restart_num;
    int m_max_mcus_per_row, m_max_blocks_per_mcu, m_max_mcus_per_col;
    u32 m_last_dc_val[JPGD_MAX_COMPONENTS];
    int16_t* m_pMCU_coefficients;
    int m_mcu_block_max_zag[JPGD_MAX_BLOCKS_PER_MCU];
    int m_error_code, m_ready_flag, m_total_bytes_read;
    float m_pixelsPerInchX, m_pixelsPerInchY, m_pixelAspectRatio;

    /* where transform_mcu / transform_mcu_expand's inputs go (the dense form of gamut_oracle.h) */
    int16_t* out_coeffs; uint8_t* out_max_zag; size_t out_blocks, out_cap;
} jd;

static void set_error(jd* d, int status) { d->m_error_code = status; longjmp(d->out, 1); }                    /* :1097-1101 */
static void undefined(jd* d) { d->undefined = 1; longjmp(d->out, 2); }
#define ERR(name) set_error(d, -1)          /* the status codes (:78-92) are not observable through the driver: null is null */

static int readfn(jd* d, uint8_t* pBuf, int max_bytes_to_read, int* pEOF_flag)
{
    size_t n = d->src_len - d->src_ofs;
    if (n > (size_t)max_bytes_to_read) n = (size_t)max_bytes_to_read;
    if (n) memcpy(pBuf, d->src + d->src_ofs, n);
    d->src_ofs += n;
    *pEOF_flag = d->src_ofs >= d->src_len;
    return (int)n;
}

static void prep_in_buffer(jd* d)                                                                             /* :1149-1174 */
{
    d->m_in_buf_left = 0;
    d->m_pIn_buf_ofs = d->m_in_buf;
    if (d->m_eof_flag) return;
    do {
        int bytes_read = readfn(d, d->m_in_buf + d->m_in_buf_left, JPGD_IN_BUF_SIZE - d->m_in_buf_left, &d->m_eof_flag);
        d->m_in_buf_left += bytes_read;
    } while (d->m_in_buf_left < JPGD_IN_BUF_SIZE && !d->m_eof_flag);
    d->m_total_bytes_read += d->m_in_buf_left;
    uint8_t* pD = d->m_pIn_buf_ofs + d->m_in_buf_left;                                                        /* word_clear(.., 0xD9FF, 64) :1136-1144 */
    for (int n = 64; n; n--) { pD[0] = 0xFF; pD[1] = 0xD9; pD += 2; }
}

static u32 get_char(jd* d)                                                                                    /* :631-652 */
{
    if (!d->m_in_buf_left) {
        prep_in_buffer(d);
        if (!d->m_in_buf_left) {
            int t = d->m_tem_flag;
            d->m_tem_flag ^= 1;
            return t ? 0xD9 : 0xFF;
        }
    }
    u32 c = *d->m_pIn_buf_ofs++;
    --d->m_in_buf_left;
    return c;
}

static void stuff_char(jd* d, uint8_t q)                                                                      /* :677-680 */
{
    if (d->m_pIn_buf_ofs <= d->m_in_buf_pad_start) undefined(d);
    *(--d->m_pIn_buf_ofs) = q;
    d->m_in_buf_left++;
}

/* :683-696.  `get_char(&padding_flag)` binds to the one-argument overload get_char(bool* err): padding_flag receives the error flag, which a
   memory stream never raises, so the two `if (padding_flag)` branches are dead and pad characters are handled like bytes of the file. */
static uint8_t get_octet(jd* d)
{
    int c = (int)get_char(d);
    if (c == 0xFF) {
        c = (int)get_char(d);
        if (c == 0x00) return 0xFF;
        stuff_char(d, (uint8_t)c);
        stuff_char(d, 0xFF);
        return 0xFF;
    }
    return (uint8_t)c;
}

static u32 get_bits(jd* d, int num_bits)                                                                      /* :699-719 */
{
    if (!num_bits) return 0;
    u32 i = d->m_bit_buf >> (32 - num_bits);
    if ((d->m_bits_left -= num_bits) <= 0) {
        d->m_bit_buf <<= (num_bits += d->m_bits_left);
        u32 c1 = get_char(d);
        u32 c2 = get_char(d);
        d->m_bit_buf = (d->m_bit_buf & 0xFFFF0000) | (c1 << 8) | c2;
        d->m_bit_buf <<= -d->m_bits_left;
        d->m_bits_left += 16;
        if (d->m_bits_left < 0) undefined(d);
    } else
        d->m_bit_buf <<= num_bits;
    return i;
}

static u32 get_bits_no_markers(jd* d, int num_bits)                                                           /* :722-743 */
{
    if (!num_bits) return 0;
    if (num_bits > 32) undefined(d);
    u32 i = d->m_bit_buf >> (32 - num_bits);
    if ((d->m_bits_left -= num_bits) <= 0) {
        num_bits += d->m_bits_left;
        if (num_bits < 0) undefined(d);
        d->m_bit_buf <<= num_bits;
        if (d->m_in_buf_left < 2 || d->m_pIn_buf_ofs[0] == 0xFF || d->m_pIn_buf_ofs[1] == 0xFF) {
            u32 c1 = get_octet(d);
            u32 c2 = get_octet(d);
            d->m_bit_buf |= (c1 << 8) | c2;
        } else {
            d->m_bit_buf |= ((u32)d->m_pIn_buf_ofs[0] << 8) | d->m_pIn_buf_ofs[1];
            d->m_in_buf_left -= 2;
            d->m_pIn_buf_ofs += 2;
        }
        d->m_bit_buf <<= -d->m_bits_left;
        d->m_bits_left += 16;
        if (d->m_bits_left < 0) undefined(d);
    } else
        d->m_bit_buf <<= num_bits;
    return i;
}

static int tree_walk(jd* d, huff_tables* pH, int symbol, int* pOfs)                                          /* the do / while of :752-756, :775-779 */
{
    int ofs = 23;
    do {
        int idx = -(int)(symbol + ((d->m_bit_buf >> ofs) & 1));
        if (idx < 0 || idx >= 512 || ofs < 0) undefined(d);
        symbol = (int)pH->tree[idx];
        --ofs;
    } while (symbol < 0);
    *pOfs = ofs;
    return symbol;
}

static int huff_decode(jd* d, huff_tables* pH)                                                                /* :746-766 */
{
    int symbol;
    if ((symbol = (int)pH->look_up[d->m_bit_buf >> 24]) < 0) {
        int ofs;
        symbol = tree_walk(d, pH, symbol, &ofs);
        get_bits_no_markers(d, 8 + (23 - ofs));
    } else
        get_bits_no_markers(d, pH->code_size[symbol]);
    return symbol;
}

static int huff_decode2(jd* d, huff_tables* pH, int* extra_bits)                                              /* :769-813 */
{
    int symbol;
    if ((symbol = (int)pH->look_up2[d->m_bit_buf >> 24]) < 0) {
        int ofs;
        symbol = tree_walk(d, pH, symbol, &ofs);
        get_bits_no_markers(d, 8 + (23 - ofs));
        *extra_bits = (int)get_bits_no_markers(d, symbol & 0xF);
    } else {
        if (symbol & 0x8000) {
            get_bits_no_markers(d, (symbol >> 8) & 31);
            *extra_bits = symbol >> 16;
        } else {
            int code_size = (symbol >> 8) & 31;
            int num_extra_bits = symbol & 0xF;
            int bits = code_size + num_extra_bits;
            if (bits <= d->m_bits_left + 16)
                *extra_bits = (int)(get_bits_no_markers(d, bits) & ((1u << num_extra_bits) - 1));
            else {
                get_bits_no_markers(d, code_size);
                *extra_bits = (int)get_bits_no_markers(d, num_extra_bits);
            }
        }
        symbol &= 0xFF;
    }
    return symbol;
}

static const int s_extend_test[16]   = { 0, 0x0001, 0x0002, 0x0004, 0x0008, 0x0010, 0x0020, 0x0040, 0x0080, 0x0100, 0x0200, 0x0400, 0x0800, 0x1000, 0x2000, 0x4000 };
static const int s_extend_offset[16] = { 0, -1, -3, -7, -15, -31, -63, -127, -255, -511, -1023, -2047, -4095, -8191, -16383, -32767 };        /* ((-1)<<n) + 1 */
static int JPGD_HUFF_EXTEND(jd* d, int x, int s)                                                              /* :816-822 */
{
    if (s < 0 || s > 15) undefined(d);
    return x < s_extend_test[s] ? x + s_extend_offset[s] : x;
}

/* ---- marker segments :1177-1543 -------------------------------------------------------------------------------- */
static void read_dht_marker(jd* d)
{
    int i, index, count;
    uint8_t huff_num[17] = {0}, huff_val[256] = {0};
    u32 num_left = get_bits(d, 16);
    if (num_left < 2) ERR(JPGD_BAD_DHT_MARKER);
    num_left -= 2;
    while (num_left) {
        index = (int)get_bits(d, 8);
        huff_num[0] = 0;
        count = 0;
        for (i = 1; i <= 16; i++) { huff_num[i] = (uint8_t)get_bits(d, 8); count += huff_num[i]; }
        if (count > 255) ERR(JPGD_BAD_DHT_COUNTS);
        for (i = 0; i < count; i++) huff_val[i] = (uint8_t)get_bits(d, 8);
        i = 1 + 16 + count;
        if (num_left < (u32)i) ERR(JPGD_BAD_DHT_MARKER);
        num_left -= i;
        if ((index & 0x10) > 0x10) ERR(JPGD_BAD_DHT_INDEX);
        index = (index & 0x0F) + ((index & 0x10) >> 4) * (JPGD_MAX_HUFF_TABLES >> 1);
        if (index >= JPGD_MAX_HUFF_TABLES) ERR(JPGD_BAD_DHT_INDEX);
        if (!d->m_huff_num[index]) d->m_huff_num[index] = (uint8_t*)calloc(17, 1);
        if (!d->m_huff_val[index]) d->m_huff_val[index] = (uint8_t*)calloc(256, 1);
        if (!d->m_huff_num[index] || !d->m_huff_val[index]) ERR(JPGD_NOTENOUGHMEM);
        d->m_huff_ac[index] = (index & 0x10) != 0;
        memcpy(d->m_huff_num[index], huff_num, 17);
        memcpy(d->m_huff_val[index], huff_val, 256);
    }
}

static void read_dqt_marker(jd* d)
{
    int n, i, prec;
    u32 num_left = get_bits(d, 16), temp;
    if (num_left < 2) ERR(JPGD_BAD_DQT_MARKER);
    num_left -= 2;
    while (num_left) {
        n = (int)get_bits(d, 8);
        prec = n >> 4;
        n &= 0x0F;
        if (n >= JPGD_MAX_QUANT_TABLES) ERR(JPGD_BAD_DQT_TABLE);
        if (!d->m_quant[n]) { d->m_quant[n] = (int16_t*)calloc(64, sizeof(int16_t)); if (!d->m_quant[n]) ERR(JPGD_NOTENOUGHMEM); }
        for (i = 0; i < 64; i++) {
            temp = get_bits(d, 8);
            if (prec) temp = (temp << 8) + get_bits(d, 8);
            d->m_quant[n][i] = (int16_t)temp;
        }
        i = 64 + 1;
        if (prec) i += 64;
        if (num_left < (u32)i) ERR(JPGD_BAD_DQT_LENGTH);
        num_left -= i;
    }
}

static void read_sof_marker(jd* d)
{
    u32 num_left = get_bits(d, 16);
    if (get_bits(d, 8) != 8) ERR(JPGD_BAD_PRECISION);
    d->m_image_y_size = (int)get_bits(d, 16);
    if (d->m_image_y_size < 1 || d->m_image_y_size > JPGD_MAX_HEIGHT) ERR(JPGD_BAD_HEIGHT);
    d->m_image_x_size = (int)get_bits(d, 16);
    if (d->m_image_x_size < 1 || d->m_image_x_size > JPGD_MAX_WIDTH) ERR(JPGD_BAD_WIDTH);
    d->m_comps_in_frame = (int)get_bits(d, 8);
    if (d->m_comps_in_frame > JPGD_MAX_COMPONENTS) ERR(JPGD_TOO_MANY_COMPONENTS);
    if (num_left != (u32)(d->m_comps_in_frame * 3 + 8)) ERR(JPGD_BAD_SOF_LENGTH);
    for (int i = 0; i < d->m_comps_in_frame; i++) {
        d->m_comp_ident[i]  = (int)get_bits(d, 8);
        d->m_comp_h_samp[i] = (int)get_bits(d, 4);
        d->m_comp_v_samp[i] = (int)get_bits(d, 4);
        d->m_comp_quant[i]  = (int)get_bits(d, 8);
    }
}

static void skip_variable_marker(jd* d)
{
    u32 num_left = get_bits(d, 16);
    if (num_left < 2) ERR(JPGD_BAD_VARIABLE_MARKER);
    num_left -= 2;
    while (num_left) { get_bits(d, 8); num_left--; }
}

static void read_dri_marker(jd* d)
{
    if (get_bits(d, 16) != 4) ERR(JPGD_BAD_DRI_LENGTH);
    d->m_restart_interval = (int)get_bits(d, 16);
}

static void read_sos_marker(jd* d)
{
    int i, ci, n, c, cc;
    u32 num_left = get_bits(d, 16);
    n = (int)get_bits(d, 8);
    d->m_comps_in_scan = n;
    num_left -= 3;
    if (num_left != (u32)(n * 2 + 3) || n < 1 || n > JPGD_MAX_COMPS_IN_SCAN) ERR(JPGD_BAD_SOS_LENGTH);
    for (i = 0; i < n; i++) {
        cc = (int)get_bits(d, 8);
        c = (int)get_bits(d, 8);
        num_left -= 2;
        for (ci = 0; ci < d->m_comps_in_frame; ci++) if (cc == d->m_comp_ident[ci]) break;
        if (ci >= d->m_comps_in_frame) ERR(JPGD_BAD_SOS_COMP_ID);
        d->m_comp_list[i] = ci;
        d->m_comp_dc_tab[ci] = (c >> 4) & 15;
        d->m_comp_ac_tab[ci] = (c & 15) + (JPGD_MAX_HUFF_TABLES >> 1);
    }
    d->m_spectral_start  = (int)get_bits(d, 8);
    d->m_spectral_end    = (int)get_bits(d, 8);
    d->m_successive_high = (int)get_bits(d, 4);
    d->m_successive_low  = (int)get_bits(d, 4);
    if (!d->m_progressive_flag) { d->m_spectral_start = 0; d->m_spectral_end = 63; }
    num_left -= 3;
    while (num_left) { get_bits(d, 8); num_left--; }
}

static int next_marker(jd* d)                                                                                 /* :1546-1573 */
{
    u32 c;
    do {
        do { c = get_bits(d, 8); } while (c != 0xFF);
        do { c = get_bits(d, 8); } while (c == 0xFF);
    } while (c == 0);
    return (int)c;
}

static float convertInchesToMeters(float x) { return x / 39.37007874f; }                                      /* types.d:127-130 */

/* `case M_APP0+1` behind its read loop, :1728-1815.  read_* of internals/binop.d advance a pointer and check nothing; a read outside exifData
   (an IFD, a value offset, a segment shorter than its "Exif\0\0") is outside the malloc block: undefined(). */
typedef struct { jd* d; const uint8_t* base; u32 len; } exif_seg;
static u32 ex_rd(exif_seg* e, u32* at, int n, int le)
{
    if ((uint64_t)*at + (uint64_t)n > e->len) undefined(e->d);
    const uint8_t* s = e->base + *at;
    *at += (u32)n;
    u32 v = 0;
    if (le) for (int k = n - 1; k >= 0; --k) v = (v << 8) | s[k];
    else    for (int k = 0; k < n; ++k)      v = (v << 8) | s[k];
    return v;
}

static void exif_segment(jd* d, const uint8_t* exifData, u32 length)
{
    exif_seg e = { d, exifData, length };
    static const uint8_t ExifIdentifierCode[6] = { 0x45, 0x78, 0x69, 0x66, 0x00, 0x00 };
    u32 s = 0;
    /* `exif_id[i] = read_ubyte(s)` x 6 reads past a shorter malloc block: whatever the heap holds there would have to spell "Exif\0\0" for the
       segment to matter.  The oracle takes it as not EXIF. */
    if (length < 6) return;
    if (memcmp(exifData, ExifIdentifierCode, 6)) return;
    s = 6;
    const u32 tiffFile = s;
    u32 byteOrder = ex_rd(&e, &s, 2, 0);
    if (byteOrder != 0x4949 && byteOrder != 0x4D4D) ERR(JPGD_DECODE_ERROR);
    const int littleEndian = byteOrder == 0x4949;
    if (ex_rd(&e, &s, 2, littleEndian) != 42) ERR(JPGD_DECODE_ERROR);
    u32 offset = ex_rd(&e, &s, 4, littleEndian);
    double resolutionX = 72, resolutionY = 72;
    int unit = 2;
    u32 hops = 0;
    while (offset != 0) {
        if (offset > length) ERR(JPGD_DECODE_ERROR);
        if (++hops > length) undefined(d);                       /* more IFDs than bytes: the chain loops, the reference never returns */
        if ((uint64_t)tiffFile + offset > 0xFFFFFFFFull) undefined(d);
        u32 pIFD = tiffFile + offset;
        u32 numEntries = ex_rd(&e, &pIFD, 2, littleEndian);
        for (u32 entry = 0; entry < numEntries; ++entry) {
            u32 tag = ex_rd(&e, &pIFD, 2, littleEndian);
            ex_rd(&e, &pIFD, 2, littleEndian);                   /* type */
            ex_rd(&e, &pIFD, 4, littleEndian);                   /* count */
            u32 valueOffset = ex_rd(&e, &pIFD, 4, littleEndian);
            if (tag == 282 || tag == 283) {
                if ((uint64_t)tiffFile + valueOffset > 0xFFFFFFFFull) undefined(d);
                u32 tagData = tiffFile + valueOffset;
                double num = ex_rd(&e, &tagData, 4, littleEndian);
                double denom = ex_rd(&e, &tagData, 4, littleEndian);
                double frac = num / denom;
                if (tag == 282) resolutionX = frac; else resolutionY = frac;
            }
            if (tag == 296) unit = (int)valueOffset;
        }
        offset = ex_rd(&e, &pIFD, 4, littleEndian);
    }
    if (unit == 2) {
        d->m_pixelsPerInchX = (float)resolutionX;
        d->m_pixelsPerInchY = (float)resolutionY;
        d->m_pixelAspectRatio = (float)(resolutionX / resolutionY);
    } else if (unit == 3) {
        d->m_pixelsPerInchX = convertInchesToMeters((float)(resolutionX * 100));
        d->m_pixelsPerInchY = convertInchesToMeters((float)(resolutionY * 100));
        d->m_pixelAspectRatio = (float)(resolutionX / resolutionY);
    }
}

/* :1578-1848.  Returns the marker; *err is the reference's `*err = true` WITHOUT an error code (RSTn / TEM / JPG :1818-1838). */
static int process_markers(jd* d, int* err, int allow_restarts)
{
    *err = 0;
    for (;;) {
        int c = next_marker(d);
        switch (c) {
        case M_SOF0: case M_SOF1: case M_SOF2: case M_SOF3: case M_SOF5: case M_SOF6: case M_SOF7:
        case M_SOF9: case M_SOF10: case M_SOF11: case M_SOF13: case M_SOF14: case M_SOF15: case M_SOI: case M_EOI: case M_SOS:
            return c;
        case M_DHT: read_dht_marker(d); break;
        case M_DAC: ERR(JPGD_NO_ARITHMITIC_SUPPORT); break;
        case M_DQT: read_dqt_marker(d); break;
        case M_DRI: read_dri_marker(d); break;
        case M_APP0: {                                                                                        /* :1634-1702 */
            u32 num_left = get_bits(d, 16);
            if (num_left < 7) ERR(JPGD_BAD_VARIABLE_MARKER);        /* the D sets the code and walks on (2^32 get_bits); null in the end */
            num_left -= 2;
            uint8_t jfif_id[5];
            for (int i = 0; i < 5; ++i) jfif_id[i] = (uint8_t)get_bits(d, 8);
            num_left -= 5;
            static const uint8_t JFIF[5] = { 0x4A, 0x46, 0x49, 0x46, 0x00 };
            if (!memcmp(jfif_id, JFIF, 5) && num_left >= 7) {
                get_bits(d, 16);
                u32 units = get_bits(d, 8);
                int Xdensity = (int)get_bits(d, 16);
                int Ydensity = (int)get_bits(d, 16);
                num_left -= 7;
                d->m_pixelAspectRatio = (float)(Xdensity / (double)Ydensity);
                switch (units) {
                case 0: d->m_pixelsPerInchX = -1; d->m_pixelsPerInchY = -1; break;
                case 1: d->m_pixelsPerInchX = (float)Xdensity; d->m_pixelsPerInchY = (float)Ydensity; break;
                case 2: d->m_pixelsPerInchX = convertInchesToMeters(Xdensity * 100.0f); d->m_pixelsPerInchY = convertInchesToMeters(Ydensity * 100.0f); break;
                default: break;
                }
            }
            while (num_left) { get_bits(d, 8); num_left--; }
        } break;
        case M_APP0 + 1: {                                                                                    /* :1704-1816 */
            u32 num_left = get_bits(d, 16);
            if (num_left < 2) ERR(JPGD_BAD_VARIABLE_MARKER);
            num_left -= 2;
            uint8_t exifData[65536];
            for (u32 i = 0; i < num_left; ++i) exifData[i] = (uint8_t)get_bits(d, 8);
            exif_segment(d, exifData, num_left);
        } break;
        case M_RST0: case M_RST0 + 1: case M_RST0 + 2: case M_RST0 + 3: case M_RST0 + 4: case M_RST0 + 5: case M_RST0 + 6: case M_RST7:
            if (allow_restarts) continue;
            *err = 1; return 0;
        case M_JPG: case M_TEM:
            *err = 1; return 0;
        default:
            skip_variable_marker(d);
            break;
        }
    }
}

static void locate_soi_marker(jd* d)                                                                          /* :1854-1908 */
{
    u32 lastchar = get_bits(d, 8), thischar = get_bits(d, 8), bytesleft;
    if (lastchar == 0xFF && thischar == M_SOI) return;
    bytesleft = 4096;
    for (;;) {
        if (--bytesleft == 0) ERR(JPGD_NOT_JPEG);
        lastchar = thischar;
        thischar = get_bits(d, 8);
        if (lastchar == 0xFF) {
            if (thischar == M_SOI) break;
            else if (thischar == M_EOI) ERR(JPGD_NOT_JPEG);
        }
    }
    thischar = (d->m_bit_buf >> 24) & 0xFF;
    if (thischar != 0xFF) ERR(JPGD_NOT_JPEG);
}

static int locate_sof_marker(jd* d)                                                                           /* :1911-1941 */
{
    locate_soi_marker(d);
    int err, c = process_markers(d, &err, 0);
    if (err) return 0;
    switch (c) {
    case M_SOF2: d->m_progressive_flag = 1; /* fall through */
    case M_SOF0: case M_SOF1: read_sof_marker(d); break;
    case M_SOF9: ERR(JPGD_NO_ARITHMITIC_SUPPORT); break;
    default: ERR(JPGD_UNSUPPORTED_MARKER); break;
    }
    return 1;
}

static int locate_sos_marker(jd* d, int* err)                                                                 /* :1944-1967 */
{
    int c = process_markers(d, err, 0);
    if (*err) return 0;
    if (c == M_EOI) return 0;
    else if (c != M_SOS) ERR(JPGD_UNEXPECTED_MARKER);
    read_sos_marker(d);
    return 1;
}

static void initit(jd* d)                                                                                     /* :1971-2078 (the struct arrives zeroed) */
{
    d->m_pIn_buf_ofs = d->m_in_buf;
    prep_in_buffer(d);
    d->m_bits_left = 16;
    d->m_bit_buf = 0;
    get_bits(d, 16);
    get_bits(d, 16);
    for (int i = 0; i < JPGD_MAX_BLOCKS_PER_MCU; i++) d->m_mcu_block_max_zag[i] = 64;
}

static void fix_in_buffer(jd* d)                                                                              /* :2098-2118 */
{
    if (d->m_bits_left & 7) undefined(d);
    if (d->m_bits_left == 16) stuff_char(d, (uint8_t)(d->m_bit_buf & 0xFF));
    if (d->m_bits_left >= 8) stuff_char(d, (uint8_t)((d->m_bit_buf >> 8) & 0xFF));
    stuff_char(d, (uint8_t)((d->m_bit_buf >> 16) & 0xFF));
    stuff_char(d, (uint8_t)((d->m_bit_buf >> 24) & 0xFF));
    d->m_bits_left = 16;
    get_bits_no_markers(d, 16);
    get_bits_no_markers(d, 16);
}

/* transform_mcu :2120-2130 / transform_mcu_expand :2139-2255 take m_pMCU_coefficients and m_mcu_block_max_zag from here; the arithmetic behind
   them is oracle_jpeg.c's (orc_jpeg_reconstruct).  A scan whose MCU is not the frame's writes m_pSample_buf in another layout than the *Convert
   functions read it in: parts of every row are memory nobody wrote. */
static void transform_mcu(jd* d, int mcu_row)
{
    (void)mcu_row;
    const int n = d->m_blocks_per_mcu;
    if (n != d->m_max_blocks_per_mcu || d->m_mcus_per_row != d->m_max_mcus_per_row) undefined(d);
    if (d->out_blocks + (size_t)n > d->out_cap) undefined(d);
    memcpy(d->out_coeffs + d->out_blocks * 64, d->m_pMCU_coefficients, (size_t)n * 64 * sizeof(int16_t));
    for (int b = 0; b < n; ++b) d->out_max_zag[d->out_blocks + b] = (uint8_t)d->m_mcu_block_max_zag[b];
    d->out_blocks += (size_t)n;
}

static int16_t* coeff_buf_getp(jd* d, coeff_buf* cb, int block_x, int block_y)                                /* :3292-3295 */
{
    if (!(block_x >= 0 && block_y >= 0 && block_x < cb->block_num_x && block_y < cb->block_num_y)) undefined(d);     /* the assert */
    return cb->pData + (size_t)block_x * cb->block_size + (size_t)block_y * ((size_t)cb->block_size * cb->block_num_x);
}

static int16_t* quant_of(jd* d, int n) { if (n < 0 || n >= JPGD_MAX_QUANT_TABLES || !d->m_quant[n]) undefined(d); return d->m_quant[n]; }
static huff_tables* huff_of(jd* d, int n) { if (n < 0 || n >= JPGD_MAX_HUFF_TABLES || !d->m_pHuff_tabs[n]) undefined(d); return d->m_pHuff_tabs[n]; }

static void load_next_row(jd* d)                                                                              /* :2259-2332 */
{
    int i, mcu_row, mcu_block, component_num, component_id;
    int block_x_mcu[JPGD_MAX_COMPONENTS];
    memset(block_x_mcu, 0, sizeof(block_x_mcu));
    for (mcu_row = 0; mcu_row < d->m_mcus_per_row; mcu_row++) {
        int block_x_mcu_ofs = 0, block_y_mcu_ofs = 0;
        for (mcu_block = 0; mcu_block < d->m_blocks_per_mcu; mcu_block++) {
            if (mcu_block >= d->m_max_blocks_per_mcu) undefined(d);
            component_id = d->m_mcu_org[mcu_block];
            int16_t* q = quant_of(d, d->m_comp_quant[component_id]);
            int16_t* p = d->m_pMCU_coefficients + 64 * mcu_block;
            int16_t* pAC = coeff_buf_getp(d, d->m_ac_coeffs[component_id], block_x_mcu[component_id] + block_x_mcu_ofs, d->m_block_y_mcu[component_id] + block_y_mcu_ofs);
            int16_t* pDC = coeff_buf_getp(d, d->m_dc_coeffs[component_id], block_x_mcu[component_id] + block_x_mcu_ofs, d->m_block_y_mcu[component_id] + block_y_mcu_ofs);
            p[0] = pDC[0];
            memcpy(&p[1], &pAC[1], 63 * sizeof(int16_t));
            for (i = 63; i > 0; i--) if (p[g_ZAG[i]]) break;
            d->m_mcu_block_max_zag[mcu_block] = i + 1;
            for (; i >= 0; i--) if (p[g_ZAG[i]]) p[g_ZAG[i]] = (int16_t)((u32)(i32)p[g_ZAG[i]] * (u32)(i32)q[i]);
            if (d->m_comps_in_scan == 1) block_x_mcu[component_id]++;
            else if (++block_x_mcu_ofs == d->m_comp_h_samp[component_id]) {
                block_x_mcu_ofs = 0;
                if (++block_y_mcu_ofs == d->m_comp_v_samp[component_id]) { block_y_mcu_ofs = 0; block_x_mcu[component_id] += d->m_comp_h_samp[component_id]; }
            }
        }
        transform_mcu(d, mcu_row);
    }
    if (d->m_comps_in_scan == 1) d->m_block_y_mcu[d->m_comp_list[0]]++;
    else for (component_num = 0; component_num < d->m_comps_in_scan; component_num++) {
        component_id = d->m_comp_list[component_num];
        d->m_block_y_mcu[component_id] += d->m_comp_v_samp[component_id];
    }
}

static void process_restart(jd* d)                                                                            /* :2335-2402 */
{
    int i, c = 0;
    for (i = 1536; i > 0; i--) if (get_char(d) == 0xFF) break;
    if (i == 0) ERR(JPGD_BAD_RESTART_MARKER);
    for (; i > 0; i--) { c = (int)get_char(d); if (c != 0xFF) break; }
    if (i == 0) ERR(JPGD_BAD_RESTART_MARKER);
    if (c != d->m_next_restart_num + M_RST0) ERR(JPGD_BAD_RESTART_MARKER);
    memset(d->m_last_dc_val, 0, (size_t)d->m_comps_in_frame * sizeof(u32));
    d->m_eob_run = 0;
    d->m_restarts_left = d->m_restart_interval;
    d->m_next_restart_num = (d->m_next_restart_num + 1) & 7;
    d->m_bits_left = 16;
    get_bits_no_markers(d, 16);
    get_bits_no_markers(d, 16);
}

static void decode_next_row(jd* d)                                                                            /* :2405-2525 */
{
    for (int mcu_row = 0; mcu_row < d->m_mcus_per_row; mcu_row++) {
        if (d->m_restart_interval && d->m_restarts_left == 0) process_restart(d);
        int16_t* p = d->m_pMCU_coefficients;
        for (int mcu_block = 0; mcu_block < d->m_blocks_per_mcu; mcu_block++, p += 64) {
            if (mcu_block >= d->m_max_blocks_per_mcu) undefined(d);
            int component_id = d->m_mcu_org[mcu_block];
            int16_t* q = quant_of(d, d->m_comp_quant[component_id]);
            int r, s;
            s = huff_decode2(d, huff_of(d, d->m_comp_dc_tab[component_id]), &r);
            s = JPGD_HUFF_EXTEND(d, r, s);
            d->m_last_dc_val[component_id] = (u32)(s = (int)((u32)s + d->m_last_dc_val[component_id]));
            p[0] = (int16_t)((u32)s * (u32)(i32)q[0]);
            int prev_num_set = d->m_mcu_block_max_zag[mcu_block];
            huff_tables* pH = huff_of(d, d->m_comp_ac_tab[component_id]);
            int k;
            for (k = 1; k < 64; k++) {
                int extra_bits;
                s = huff_decode2(d, pH, &extra_bits);
                r = s >> 4;
                s &= 15;
                if (s) {
                    if (r) {
                        if (k + r > 63) ERR(JPGD_DECODE_ERROR);
                        if (k < prev_num_set) {
                            int n = r < prev_num_set - k ? r : prev_num_set - k;
                            int kt = k;
                            while (n--) p[g_ZAG[kt++]] = 0;
                        }
                        k += r;
                    }
                    s = JPGD_HUFF_EXTEND(d, extra_bits, s);
                    p[g_ZAG[k]] = (int16_t)((u32)s * (u32)(i32)q[k]);
                } else {
                    if (r == 15) {
                        if (k + 16 > 64) ERR(JPGD_DECODE_ERROR);
                        if (k < prev_num_set) {
                            int n = 16 < prev_num_set - k ? 16 : prev_num_set - k;
                            int kt = k;
                            while (n--) p[g_ZAG[kt++]] = 0;
                        }
                        k += 16 - 1;
                    } else
                        break;
                }
            }
            if (k < prev_num_set) { int kt = k; while (kt < prev_num_set) p[g_ZAG[kt++]] = 0; }
            d->m_mcu_block_max_zag[mcu_block] = k;
        }
        transform_mcu(d, mcu_row);
        d->m_restarts_left--;
    }
}

static int find_eoi(jd* d)                                                                                    /* :2826-2848 */
{
    if (!d->m_progressive_flag) {
        d->m_bits_left = 16;
        get_bits(d, 16);
        get_bits(d, 16);
        int err;
        process_markers(d, &err, 1);
        if (err) return 0;
    }
    d->m_total_bytes_read -= d->m_in_buf_left;
    return 1;
}

static void make_huff_table(jd* d, int index, huff_tables* pH)                                                /* :2851-2987 */
{
    int p, i, l, si;
    uint8_t huffsize[257];
    u32 huffcode[257];
    u32 code, subtree;
    int code_size, lastp, nextfreeentry, currententry;

    pH->ac_table = d->m_huff_ac[index] != 0;
    p = 0;
    for (l = 1; l <= 16; l++) for (i = 1; i <= d->m_huff_num[index][l]; i++) { if (p > 256) undefined(d); huffsize[p++] = (uint8_t)l; }
    if (p > 256) undefined(d);
    huffsize[p] = 0;
    lastp = p;
    {   /* codes past 2^length: `look_up[code]` is written behind the array for lengths <= 8 (:2917), longer ones alias other prefixes through
           `& 0xFF` (:2946) and overwrite tree nodes: no table to restate */
        u32 kraft = 0;
        for (l = 1; l <= 16; l++) kraft += (u32)d->m_huff_num[index][l] << (16 - l);
        if (kraft > (1u << 16)) undefined(d);
    }
    code = 0;
    si = huffsize[0];
    p = 0;
    while (huffsize[p]) {
        while (huffsize[p] == si) { huffcode[p++] = code; code++; }
        code <<= 1;
        si++;
    }
    memset(pH->look_up, 0, sizeof(pH->look_up));
    memset(pH->look_up2, 0, sizeof(pH->look_up2));
    memset(pH->tree, 0, sizeof(pH->tree));
    memset(pH->code_size, 0, sizeof(pH->code_size));
    nextfreeentry = -1;
    p = 0;
    while (p < lastp) {
        i = d->m_huff_val[index][p];
        code = huffcode[p];
        code_size = huffsize[p];
        pH->code_size[i] = (uint8_t)code_size;
        if (code_size <= 8) {
            code <<= (8 - code_size);
            for (l = 1 << (8 - code_size); l > 0; l--) {
                if (code > 255) undefined(d);
                pH->look_up[code] = (u32)i;
                int has_extrabits = 0, extra_bits = 0, num_extra_bits = i & 15, bits_to_fetch = code_size;
                if (num_extra_bits) {
                    int total_codesize = code_size + num_extra_bits;
                    if (total_codesize <= 8) {
                        has_extrabits = 1;
                        extra_bits = ((1 << num_extra_bits) - 1) & (int)(code >> (8 - total_codesize));
                        bits_to_fetch += num_extra_bits;
                    }
                }
                if (!has_extrabits) pH->look_up2[code] = (u32)(i | (bits_to_fetch << 8));
                else pH->look_up2[code] = (u32)(i | 0x8000 | (extra_bits << 16) | (bits_to_fetch << 8));
                code++;
            }
        } else {
            subtree = (code >> (code_size - 8)) & 0xFF;
            currententry = (int)pH->look_up[subtree];
            if (currententry == 0) {
                pH->look_up[subtree] = (u32)(currententry = nextfreeentry);
                pH->look_up2[subtree] = (u32)(currententry = nextfreeentry);
                nextfreeentry -= 2;
            } else if (currententry > 0) undefined(d);
            code <<= (16 - (code_size - 8));
            for (l = code_size; l > 9; l--) {
                if ((code & 0x8000) == 0) currententry--;
                if (-currententry - 1 < 0 || -currententry - 1 >= 512) undefined(d);
                if (pH->tree[-currententry - 1] == 0) {
                    pH->tree[-currententry - 1] = (u32)nextfreeentry;
                    currententry = nextfreeentry;
                    nextfreeentry -= 2;
                } else {
                    currententry = (int)pH->tree[-currententry - 1];
                    if (currententry > 0) undefined(d);
                }
                code <<= 1;
            }
            if ((code & 0x8000) == 0) currententry--;
            if (-currententry - 1 < 0 || -currententry - 1 >= 512) undefined(d);
            pH->tree[-currententry - 1] = (u32)i;
        }
        p++;
    }
}

static void check_quant_tables(jd* d)                                                                         /* :2990-3000 */
{
    for (int i = 0; i < d->m_comps_in_scan; i++) {
        int n = d->m_comp_quant[d->m_comp_list[i]];
        if (n < 0 || n >= JPGD_MAX_QUANT_TABLES) undefined(d);
        if (!d->m_quant[n]) ERR(JPGD_UNDEFINED_QUANT_TABLE);
    }
}

static void check_huff_tables(jd* d)                                                                          /* :3003-3034 */
{
    for (int i = 0; i < d->m_comps_in_scan; i++) {
        const int dc = d->m_comp_dc_tab[d->m_comp_list[i]], ac = d->m_comp_ac_tab[d->m_comp_list[i]];
        if (d->m_spectral_start == 0) { if (dc >= JPGD_MAX_HUFF_TABLES) undefined(d); if (!d->m_huff_num[dc]) ERR(JPGD_UNDEFINED_HUFF_TABLE); }
        if (d->m_spectral_end > 0)    { if (ac >= JPGD_MAX_HUFF_TABLES) undefined(d); if (!d->m_huff_num[ac]) ERR(JPGD_UNDEFINED_HUFF_TABLE); }
    }
    for (int i = 0; i < JPGD_MAX_HUFF_TABLES; i++)
        if (d->m_huff_num[i]) {
            if (!d->m_pHuff_tabs[i]) { d->m_pHuff_tabs[i] = (huff_tables*)calloc(1, sizeof(huff_tables)); if (!d->m_pHuff_tabs[i]) ERR(JPGD_NOTENOUGHMEM); }
            make_huff_table(d, i, d->m_pHuff_tabs[i]);
        }
}

static void calc_mcu_block_order(jd* d)                                                                       /* :3038-3090 */
{
    int component_num, component_id, max_h_samp = 0, max_v_samp = 0;
    for (component_id = 0; component_id < d->m_comps_in_frame; component_id++) {
        if (d->m_comp_h_samp[component_id] > max_h_samp) max_h_samp = d->m_comp_h_samp[component_id];
        if (d->m_comp_v_samp[component_id] > max_v_samp) max_v_samp = d->m_comp_v_samp[component_id];
    }
    for (component_id = 0; component_id < d->m_comps_in_frame; component_id++) {
        d->m_comp_h_blocks[component_id] = ((((d->m_image_x_size * d->m_comp_h_samp[component_id]) + (max_h_samp - 1)) / max_h_samp) + 7) / 8;
        d->m_comp_v_blocks[component_id] = ((((d->m_image_y_size * d->m_comp_v_samp[component_id]) + (max_v_samp - 1)) / max_v_samp) + 7) / 8;
    }
    if (d->m_comps_in_scan == 1) {
        d->m_mcus_per_row = d->m_comp_h_blocks[d->m_comp_list[0]];
        d->m_mcus_per_col = d->m_comp_v_blocks[d->m_comp_list[0]];
    } else {
        d->m_mcus_per_row = (((d->m_image_x_size + 7) / 8) + (max_h_samp - 1)) / max_h_samp;
        d->m_mcus_per_col = (((d->m_image_y_size + 7) / 8) + (max_v_samp - 1)) / max_v_samp;
    }
    if (d->m_comps_in_scan == 1) {
        d->m_mcu_org[0] = d->m_comp_list[0];
        d->m_blocks_per_mcu = 1;
    } else {
        d->m_blocks_per_mcu = 0;
        for (component_num = 0; component_num < d->m_comps_in_scan; component_num++) {
            component_id = d->m_comp_list[component_num];
            int num_blocks = d->m_comp_h_samp[component_id] * d->m_comp_v_samp[component_id];
            while (num_blocks--) {
                if (d->m_blocks_per_mcu >= JPGD_MAX_BLOCKS_PER_MCU) undefined(d);
                d->m_mcu_org[d->m_blocks_per_mcu++] = component_id;
            }
        }
    }
}

static int init_scan(jd* d, int* err)                                                                         /* :3093-3125 */
{
    if (!locate_sos_marker(d, err)) return 0;
    calc_mcu_block_order(d);
    check_huff_tables(d);
    check_quant_tables(d);
    memset(d->m_last_dc_val, 0, (size_t)d->m_comps_in_frame * sizeof(u32));
    d->m_eob_run = 0;
    if (d->m_restart_interval) { d->m_restarts_left = d->m_restart_interval; d->m_next_restart_num = 0; }
    fix_in_buffer(d);
    return 1;
}

static void init_frame(jd* d)                                                                                 /* :3130-3268 */
{
    const int* hs = d->m_comp_h_samp; const int* vs = d->m_comp_v_samp;
    if (d->m_comps_in_frame == 1) {
        if (hs[0] != 1 || vs[0] != 1) ERR(JPGD_UNSUPPORTED_SAMP_FACTORS);
        d->m_scan_type = ORC_JPGD_GRAYSCALE; d->m_max_blocks_per_mcu = 1; d->m_max_mcu_x_size = 8; d->m_max_mcu_y_size = 8;
    } else if (d->m_comps_in_frame == 3) {
        if ((hs[1] != 1 || vs[1] != 1) || (hs[2] != 1 || vs[2] != 1)) ERR(JPGD_UNSUPPORTED_SAMP_FACTORS);
        if      (hs[0] == 1 && vs[0] == 1) { d->m_scan_type = ORC_JPGD_YH1V1; d->m_max_blocks_per_mcu = 3; d->m_max_mcu_x_size = 8;  d->m_max_mcu_y_size = 8; }
        else if (hs[0] == 2 && vs[0] == 1) { d->m_scan_type = ORC_JPGD_YH2V1; d->m_max_blocks_per_mcu = 4; d->m_max_mcu_x_size = 16; d->m_max_mcu_y_size = 8; }
        else if (hs[0] == 1 && vs[0] == 2) { d->m_scan_type = ORC_JPGD_YH1V2; d->m_max_blocks_per_mcu = 4; d->m_max_mcu_x_size = 8;  d->m_max_mcu_y_size = 16; }
        else if (hs[0] == 2 && vs[0] == 2) { d->m_scan_type = ORC_JPGD_YH2V2; d->m_max_blocks_per_mcu = 6; d->m_max_mcu_x_size = 16; d->m_max_mcu_y_size = 16; }
        else ERR(JPGD_UNSUPPORTED_SAMP_FACTORS);
    } else
        ERR(JPGD_UNSUPPORTED_COLORSPACE);
    d->m_max_mcus_per_row = (d->m_image_x_size + (d->m_max_mcu_x_size - 1)) / d->m_max_mcu_x_size;
    d->m_max_mcus_per_col = (d->m_image_y_size + (d->m_max_mcu_y_size - 1)) / d->m_max_mcu_y_size;
    d->m_max_blocks_per_row = d->m_max_mcus_per_row * d->m_max_blocks_per_mcu;
    if (d->m_max_blocks_per_row > JPGD_MAX_BLOCKS_PER_ROW) ERR(JPGD_ASSERTION_ERROR);
    d->m_pMCU_coefficients = (int16_t*)calloc((size_t)d->m_max_blocks_per_mcu * 64, sizeof(int16_t));      /* the reference does not clear it; the first MCU does (max_zag starts at 64) */
    if (!d->m_pMCU_coefficients) ERR(JPGD_NOTENOUGHMEM);
    for (int i = 0; i < d->m_max_blocks_per_mcu; i++) d->m_mcu_block_max_zag[i] = 64;
    d->m_total_lines_left = d->m_image_y_size;
    d->m_mcu_lines_left = 0;
    /* the dense store the caller gets */
    d->out_cap = (size_t)d->m_max_mcus_per_row * d->m_max_mcus_per_col * d->m_max_blocks_per_mcu;
    d->out_coeffs = (int16_t*)calloc(d->out_cap * 64, sizeof(int16_t));
    d->out_max_zag = (uint8_t*)calloc(d->out_cap ? d->out_cap : 1, 1);
    if (!d->out_coeffs || !d->out_max_zag) ERR(JPGD_NOTENOUGHMEM);
}

static coeff_buf* coeff_buf_open(jd* d, int block_num_x, int block_num_y, int block_len_x, int block_len_y)   /* :3274-3290 */
{
    coeff_buf* cb = (coeff_buf*)calloc(1, sizeof(coeff_buf));
    if (!cb) ERR(JPGD_NOTENOUGHMEM);
    cb->block_num_x = block_num_x; cb->block_num_y = block_num_y; cb->block_len_x = block_len_x; cb->block_len_y = block_len_y;
    cb->block_size = block_len_x * block_len_y;
    cb->pData = (int16_t*)calloc((size_t)cb->block_size * block_num_x * block_num_y + 1, sizeof(int16_t));
    if (!cb->pData) { free(cb); ERR(JPGD_NOTENOUGHMEM); }
    return cb;
}

/* ---- progressive block decoders :3299-3518 --------------------------------------------------------------------- */
static void decode_block_dc_first(jd* d, int component_id, int block_x, int block_y)
{
    int s, r;
    int16_t* p = coeff_buf_getp(d, d->m_dc_coeffs[component_id], block_x, block_y);
    s = huff_decode(d, huff_of(d, d->m_comp_dc_tab[component_id]));
    if (s != 0) {
        if (s > 15) undefined(d);                                /* get_bits_no_markers(s) / s_extend_test[s] */
        r = (int)get_bits_no_markers(d, s);
        s = JPGD_HUFF_EXTEND(d, r, s);
    }
    d->m_last_dc_val[component_id] = (u32)(s = (int)((u32)s + d->m_last_dc_val[component_id]));
    p[0] = (int16_t)((u32)s << d->m_successive_low);
}

static void decode_block_dc_refine(jd* d, int component_id, int block_x, int block_y)
{
    if (get_bits_no_markers(d, 1)) {
        int16_t* p = coeff_buf_getp(d, d->m_dc_coeffs[component_id], block_x, block_y);
        p[0] = (int16_t)(p[0] | (1 << d->m_successive_low));
    }
}

static void decode_block_ac_first(jd* d, int component_id, int block_x, int block_y)
{
    int k, s, r;
    if (d->m_eob_run) { d->m_eob_run--; return; }
    int16_t* p = coeff_buf_getp(d, d->m_ac_coeffs[component_id], block_x, block_y);
    huff_tables* pH = huff_of(d, d->m_comp_ac_tab[component_id]);
    for (k = d->m_spectral_start; k <= d->m_spectral_end; k++) {
        s = huff_decode(d, pH);
        r = s >> 4;
        s &= 15;
        if (s) {
            if ((k += r) > 63) ERR(JPGD_DECODE_ERROR);
            r = (int)get_bits_no_markers(d, s);
            s = JPGD_HUFF_EXTEND(d, r, s);
            p[g_ZAG[k]] = (int16_t)((u32)s << d->m_successive_low);
        } else {
            if (r == 15) {
                if ((k += 15) > 63) ERR(JPGD_DECODE_ERROR);
            } else {
                d->m_eob_run = 1 << r;
                if (r) d->m_eob_run += (int)get_bits_no_markers(d, r);
                d->m_eob_run--;
                break;
            }
        }
    }
}

static void decode_block_ac_refine(jd* d, int component_id, int block_x, int block_y)
{
    int s, k, r;
    int p1 = 1 << d->m_successive_low;
    int m1 = (int)(0xFFFFFFFFu << d->m_successive_low);
    int16_t* p = coeff_buf_getp(d, d->m_ac_coeffs[component_id], block_x, block_y);
    if (d->m_spectral_end > 63) undefined(d);
    huff_tables* pH = huff_of(d, d->m_comp_ac_tab[component_id]);
    k = d->m_spectral_start;
    if (d->m_eob_run == 0) {
        for (; k <= d->m_spectral_end; k++) {
            s = huff_decode(d, pH);
            r = s >> 4;
            s &= 15;
            if (s) {
                if (s != 1) ERR(JPGD_DECODE_ERROR);
                if (get_bits_no_markers(d, 1)) s = p1; else s = m1;
            } else {
                if (r != 15) {
                    d->m_eob_run = 1 << r;
                    if (r) d->m_eob_run += (int)get_bits_no_markers(d, r);
                    break;
                }
            }
            do {
                int16_t* this_coef = p + g_ZAG[k & 63];
                if (*this_coef != 0) {
                    if (get_bits_no_markers(d, 1)) {
                        if ((*this_coef & p1) == 0) {
                            if (*this_coef >= 0) *this_coef = (int16_t)(*this_coef + p1);
                            else *this_coef = (int16_t)(*this_coef + m1);
                        }
                    }
                } else {
                    if (--r < 0) break;
                }
                k++;
            } while (k <= d->m_spectral_end);
            if (s && k < 64) p[g_ZAG[k]] = (int16_t)s;
        }
    }
    if (d->m_eob_run > 0) {
        for (; k <= d->m_spectral_end; k++) {
            int16_t* this_coef = p + g_ZAG[k & 63];
            if (*this_coef != 0) {
                if (get_bits_no_markers(d, 1)) {
                    if ((*this_coef & p1) == 0) {
                        if (*this_coef >= 0) *this_coef = (int16_t)(*this_coef + p1);
                        else *this_coef = (int16_t)(*this_coef + m1);
                    }
                }
            }
        }
        d->m_eob_run--;
    }
}

typedef void (*pDecode_block_func)(jd*, int, int, int);

static void decode_scan(jd* d, pDecode_block_func decode_block_func)                                          /* :3521-3584 */
{
    int mcu_row, mcu_col, mcu_block;
    int block_x_mcu[JPGD_MAX_COMPONENTS], m_block_y_mcu[JPGD_MAX_COMPONENTS];       /* the local shadows the member */
    memset(m_block_y_mcu, 0, sizeof(m_block_y_mcu));
    for (mcu_col = 0; mcu_col < d->m_mcus_per_col; mcu_col++) {
        int component_num, component_id;
        memset(block_x_mcu, 0, sizeof(block_x_mcu));
        for (mcu_row = 0; mcu_row < d->m_mcus_per_row; mcu_row++) {
            int block_x_mcu_ofs = 0, block_y_mcu_ofs = 0;
            if (d->m_restart_interval && d->m_restarts_left == 0) process_restart(d);
            for (mcu_block = 0; mcu_block < d->m_blocks_per_mcu; mcu_block++) {
                component_id = d->m_mcu_org[mcu_block];
                decode_block_func(d, component_id, block_x_mcu[component_id] + block_x_mcu_ofs, m_block_y_mcu[component_id] + block_y_mcu_ofs);
                if (d->m_comps_in_scan == 1) block_x_mcu[component_id]++;
                else if (++block_x_mcu_ofs == d->m_comp_h_samp[component_id]) {
                    block_x_mcu_ofs = 0;
                    if (++block_y_mcu_ofs == d->m_comp_v_samp[component_id]) { block_y_mcu_ofs = 0; block_x_mcu[component_id] += d->m_comp_h_samp[component_id]; }
                }
            }
            d->m_restarts_left--;
        }
        if (d->m_comps_in_scan == 1) m_block_y_mcu[d->m_comp_list[0]]++;
        else for (component_num = 0; component_num < d->m_comps_in_scan; component_num++) {
            component_id = d->m_comp_list[component_num];
            m_block_y_mcu[component_id] += d->m_comp_v_samp[component_id];
        }
    }
}

static int init_progressive(jd* d)                                                                            /* :3587-3683 */
{
    int i;
    if (d->m_comps_in_frame == 4) ERR(JPGD_UNSUPPORTED_COLORSPACE);
    for (i = 0; i < d->m_comps_in_frame; i++) {
        d->m_dc_coeffs[i] = coeff_buf_open(d, d->m_max_mcus_per_row * d->m_comp_h_samp[i], d->m_max_mcus_per_col * d->m_comp_v_samp[i], 1, 1);
        d->m_ac_coeffs[i] = coeff_buf_open(d, d->m_max_mcus_per_row * d->m_comp_h_samp[i], d->m_max_mcus_per_col * d->m_comp_v_samp[i], 8, 8);
    }
    for (;;) {
        int dc_only_scan, refinement_scan, err;
        pDecode_block_func decode_block_func;
        int scanInit = init_scan(d, &err);
        if (err) return 0;
        if (!scanInit) break;
        dc_only_scan = (d->m_spectral_start == 0);
        refinement_scan = (d->m_successive_high != 0);
        if (d->m_spectral_start > d->m_spectral_end || d->m_spectral_end > 63) ERR(JPGD_BAD_SOS_SPECTRAL);
        if (dc_only_scan) { if (d->m_spectral_end) ERR(JPGD_BAD_SOS_SPECTRAL); }
        else if (d->m_comps_in_scan != 1) ERR(JPGD_BAD_SOS_SPECTRAL);
        if (refinement_scan && d->m_successive_low != d->m_successive_high - 1) ERR(JPGD_BAD_SOS_SUCCESSIVE);
        if (dc_only_scan) decode_block_func = refinement_scan ? decode_block_dc_refine : decode_block_dc_first;
        else              decode_block_func = refinement_scan ? decode_block_ac_refine : decode_block_ac_first;
        decode_scan(d, decode_block_func);
        d->m_bits_left = 16;
        get_bits(d, 16);
        get_bits(d, 16);
    }
    d->m_comps_in_scan = d->m_comps_in_frame;
    for (i = 0; i < d->m_comps_in_frame; i++) d->m_comp_list[i] = i;
    calc_mcu_block_order(d);
    return 1;
}

static int init_sequential(jd* d)                                                                             /* :3685-3695 */
{
    int err;
    if (!init_scan(d, &err)) ERR(JPGD_UNEXPECTED_MARKER);
    return 1;
}

static int decode_start(jd* d)                                                                                /* :3697-3706 */
{
    init_frame(d);
    return d->m_progressive_flag ? init_progressive(d) : init_sequential(d);
}

static int begin_decoding(jd* d)                                                                              /* :530-537 */
{
    if (d->m_ready_flag) return JPGD_SUCCESS;
    if (d->m_error_code) return JPGD_FAILED;
    /* The D drops decode_start's result.  It is false without an error code in one case: init_progressive left through `if (err) return false`
       (:3615) on RSTn / TEM / JPG between two scans (:1818-1838) -- decode() then runs load_next_row over whatever the scans so far left, in the
       block order of the last one, into a sample buffer that order does not fill. */
    if (!decode_start(d)) undefined(d);
    d->m_ready_flag = 1;
    return JPGD_SUCCESS;
}

static int decode(jd* d)                                                                                      /* :545-612, up to the colour conversion */
{
    if (d->m_error_code || !d->m_ready_flag) return JPGD_FAILED;
    if (d->m_total_lines_left == 0) return JPGD_DONE;
    if (d->m_mcu_lines_left == 0) {
        if (d->m_progressive_flag) load_next_row(d);
        else decode_next_row(d);
        if (d->m_total_lines_left <= d->m_max_mcu_y_size) if (!find_eoi(d)) return JPGD_FAILED;
        d->m_mcu_lines_left = d->m_max_mcu_y_size;
    }
    --d->m_mcu_lines_left;
    --d->m_total_lines_left;
    return JPGD_SUCCESS;
}

static void free_all(jd* d)
{
    for (int i = 0; i < JPGD_MAX_HUFF_TABLES; ++i) { free(d->m_huff_num[i]); free(d->m_huff_val[i]); free(d->m_pHuff_tabs[i]); }
    for (int i = 0; i < JPGD_MAX_QUANT_TABLES; ++i) free(d->m_quant[i]);
    for (int i = 0; i < JPGD_MAX_COMPONENTS; ++i) {
        if (d->m_dc_coeffs[i]) { free(d->m_dc_coeffs[i]->pData); free(d->m_dc_coeffs[i]); }
        if (d->m_ac_coeffs[i]) { free(d->m_ac_coeffs[i]->pData); free(d->m_ac_coeffs[i]); }
    }
    free(d->m_pMCU_coefficients);
}

/* decompress_jpeg_image_from_stream :3720-3808 up to the pixel loop's colour work: constructor (decode_init :3708-3713), begin_decoding, one
   decode() per line.  0 = the reference returns an image, -1 = null, -2 = the reference has no defined result (rejected by the oracle). */
int orc_jpeg_decode_coeffs(const uint8_t* data, size_t len, orc_jpeg_frame* f)
{
    memset(f, 0, sizeof(*f));
    jd* d = (jd*)calloc(1, sizeof(jd));
    if (!d) return -1;
    d->src = data; d->src_len = data ? len : 0;
    d->m_pixelsPerInchX = d->m_pixelsPerInchY = d->m_pixelAspectRatio = NAN;          /* float members of a D struct start as NaN; initit never assigns them */
    volatile int rc = -1;
    const int jumped = setjmp(d->out);
    if (jumped == 0) {
        initit(d);
        if (locate_sof_marker(d) && d->m_error_code == JPGD_SUCCESS) {
            const int image_height = d->m_image_y_size;
            if (begin_decoding(d) == JPGD_SUCCESS) {
                int y;
                for (y = 0; y < image_height; ++y) if (decode(d) != JPGD_SUCCESS) break;
                if (y == image_height) rc = 0;
            }
        }
    } else
        rc = jumped == 2 ? -2 : -1;
    if (rc == 0) {
        f->width = d->m_image_x_size; f->height = d->m_image_y_size; f->comps = d->m_comps_in_frame; f->scan_type = d->m_scan_type;
        f->mcus_per_row = d->m_max_mcus_per_row; f->mcus_per_col = d->m_max_mcus_per_col; f->blocks_per_mcu = d->m_max_blocks_per_mcu;
        f->coeffs = d->out_coeffs; f->max_zag = d->out_max_zag;
        f->pixel_aspect_ratio = d->m_pixelAspectRatio; f->dpi_y = d->m_pixelsPerInchY;
        if (d->out_blocks != d->out_cap) rc = -2;                 /* cannot happen: every MCU row was transformed */
    }
    if (rc != 0) { free(d->out_coeffs); free(d->out_max_zag); f->coeffs = NULL; f->max_zag = NULL; }
    free_all(d);
    free(d);
    return rc;
}

void orc_jpeg_frame_free(orc_jpeg_frame* f)
{
    free(f->coeffs); free(f->max_zag);
    f->coeffs = NULL; f->max_zag = NULL;
}
