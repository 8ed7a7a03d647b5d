/*
 * gamut_oracle.h -- CPU restatement ("oracle") of the Gamut hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (gamut_amd/, include/)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / CPU baseline.
 *
 * Every function restates, in plain C, what the reference computes in the
 * cited file:line of /root/reference (AuburnSounds/gamut, D language).  The
 * reference cannot be compiled here (no D compiler in the image), so parity
 * is pinned through the reference's own fixtures and independent decoders:
 * see tests/test_oracle_*.py and DESIGN.md "Oracle pinning".
 */
#ifndef GAMUT_ORACLE_H
#define GAMUT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* PixelType ordinals: source/gamut/types.d:32-59 */
enum {
    ORC_unknown = -1,
    ORC_l8 = 0, ORC_l16, ORC_lf32,
    ORC_la8, ORC_la16, ORC_laf32,
    ORC_lap8, ORC_lap16, ORC_lapf32,
    ORC_rgb8, ORC_rgb16, ORC_rgbf32,
    ORC_rgba8, ORC_rgba16, ORC_rgbaf32,
    ORC_rgbap8, ORC_rgbap16, ORC_rgbapf32,
    ORC_NUM_TYPES
};

/* ---- scanline conversion (source/gamut/scanline.d) ---------------------- */
int  orc_pixel_type_size(int type);                         /* types.d:62-86 */
int  orc_scanlines_inter_type(int srcType, int dstType);    /* scanline.d:25-31 */
/* scanline.d:37-55 */
int  orc_scanlines_copy(int type, const uint8_t* src, int srcPitch,
                        uint8_t* dst, int dstPitch, int width, int height);
/* scanline.d:70-121; interBuf must hold width*size(interType) bytes */
int  orc_scanlines_convert(int srcType, const uint8_t* src, int srcPitch,
                           int dstType, uint8_t* dst, int dstPitch,
                           int width, int height,
                           int interType, uint8_t* interBuf);

/* ---- JPEG (source/gamut/codecs/jpegload.d) ------------------------------ */
enum { ORC_JPGD_GRAYSCALE = 0, ORC_JPGD_YH1V1, ORC_JPGD_YH2V1, ORC_JPGD_YH1V2, ORC_JPGD_YH2V2 };

/* one 8x8 block: jpegload.d:308-376 (block_max_zag selects the sparse paths) */
void orc_jpeg_idct(const int16_t* src, uint8_t* dst, int block_max_zag);
/* jpegload.d:378-397 */
void orc_jpeg_idct_4x4(const int16_t* src, uint8_t* dst);
/* test-only variant: columns first, then rows (libjpeg order), used to pin the
 * butterfly + colour arithmetic against Pillow; never the product order. */
void orc_jpeg_idct_colfirst(const int16_t* src, uint8_t* dst);
/* frequency-domain 2x chroma upsample of ONE chroma block into four 4x4-sparse
 * coefficient blocks (jpegload.d:827-1073, 2155-2251). out = 4*64 int16 */
void orc_jpeg_upsample_block(const int16_t* src, int max_zag, int16_t* out4);

typedef struct {
    int width, height;
    int comps;            /* 1 or 3 (m_comps_in_frame) */
    int scan_type;        /* ORC_JPGD_* (jpegload.d:3130-3195) */
    int mcus_per_row, mcus_per_col, blocks_per_mcu;
    /* dense coefficient store: mcus_per_col*mcus_per_row*blocks_per_mcu blocks
     * of 64 int16, natural order, de-quantised (jpegload.d:2432,2474), MCU
     * block order of calc_mcu_block_order (jpegload.d:3076-3088) */
    int16_t* coeffs;
    uint8_t* max_zag;     /* m_mcu_block_max_zag per block (jpegload.d:2512) */
    float pixel_aspect_ratio, dpi_y;
} orc_jpeg_frame;

/* Baseline (SOF0/SOF1, Huffman, 8-bit) entropy decode of a whole file into the
 * dense coefficient form above (jpegload.d:1578-1848 markers, 2405-2525 decode).
 * Returns 0 on success. Free with orc_jpeg_frame_free. */
int  orc_jpeg_decode_coeffs(const uint8_t* data, size_t len, orc_jpeg_frame* out);
void orc_jpeg_frame_free(orc_jpeg_frame* f);

/* Reconstruct pixels from the dense coefficients exactly as
 * transform_mcu(_expand) + *Convert + decompress_jpeg_image_from_stream do
 * (jpegload.d:2120-2255, 2528-2823, 3753-3802).  req_comps in {1,3,4};
 * out rows are out_pitch bytes apart (reference: width*req_comps).
 * colfirst != 0 uses the libjpeg pass order (test-only). Returns 0 on success. */
int  orc_jpeg_reconstruct(const orc_jpeg_frame* f, int req_comps,
                          uint8_t* out, int out_pitch, int colfirst);

/* whole decompress_jpeg_image_from_stream (jpegload.d:3720-3808) on a memory
 * buffer: returns malloc'd width*req_comps*height bytes or NULL. */
uint8_t* orc_decompress_jpeg_image_from_memory(const uint8_t* data, size_t len,
        int* width, int* height, int* actual_comps, float* pixelAspectRatio,
        float* dotsPerInchY, int req_comps);

/* ---- PNG (source/gamut/codecs/stbdec.d) --------------------------------- */
/* stbi__create_png_image_raw, stbdec.d:1406-1635.  raw = inflated stream.
 * out must hold x*y*out_n*(depth==16?2:1) bytes. returns 1 ok / 0 corrupt. */
int  orc_png_create_image_raw(const uint8_t* raw, uint32_t raw_len, int img_n, int out_n,
                              uint32_t x, uint32_t y, int depth, int color, uint8_t* out);
/* stbi__create_png_image incl. Adam7 (stbdec.d:1637-1680) */
int  orc_png_create_image(const uint8_t* raw, uint32_t raw_len, int img_n, int out_n,
                          uint32_t x, uint32_t y, int depth, int color, int interlaced,
                          uint8_t* out);

typedef struct {
    uint32_t width, height;
    int depth, color, interlace;
    int img_n;             /* channels that are filtered (1 for palette) */
    int pal_img_n;         /* 0, 3 or 4 */
    int has_trans;
    int is_iphone;
    uint8_t  palette[1024];
    uint32_t pal_len;
    uint8_t  tc[3];
    uint16_t tc16[3];
    uint8_t* raw;          /* inflated stream (malloc) */
    uint32_t raw_len;
    float ppmX, ppmY, pixelAspectRatio;
} orc_png_info;

/* chunk parse + IDAT concat + inflate (stbdec.d:1777-2023; inflate by zlib in
 * place of the reference's miniz, stbdec.d:1262-1321). returns 1 ok. */
int  orc_png_parse(const uint8_t* data, size_t len, orc_png_info* info);
void orc_png_info_free(orc_png_info* info);

/* stbi_load_from_callbacks / stbi_load_16_from_callbacks on a memory buffer
 * (stbdec.d:669-735, 2025-2062).  Returns malloc'd pixels or NULL. */
uint8_t*  orc_stbi_load_from_memory(const uint8_t* data, size_t len, int* x, int* y,
                                    int* comp, int req_comp);
uint16_t* orc_stbi_load_16_from_memory(const uint8_t* data, size_t len, int* x, int* y,
                                       int* comp, int req_comp);

/* post passes, exposed for unit tests (stbdec.d:635-666, 916-1199, 1682-1765) */
void orc_png_convert_format8(const uint8_t* src, int img_n, int req_comp, uint32_t x, uint32_t y, uint8_t* dst);
void orc_png_convert_format16(const uint16_t* src, int img_n, int req_comp, uint32_t x, uint32_t y, uint16_t* dst);

/* ---- QOI (codecs/qoi.d:448-550) -------------------------------------------------------------------------------- */
typedef struct { uint32_t width, height; uint8_t channels, colorspace; } orc_qoi_desc;
/* qoi_decode: channels = 0 (as in the file), 3 or 4; returns malloc'd width*height*channels bytes or NULL */
uint8_t*  orc_qoi_decode(const uint8_t* data, int size, orc_qoi_desc* desc, int channels);

#ifdef __cplusplus
}
#endif
#endif
