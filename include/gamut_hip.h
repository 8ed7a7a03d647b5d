/*
 * gamut_hip.h -- C ABI of the MI355X (gfx950) batched image-decode /
 * pixel-convert path for Gamut (AuburnSounds/gamut).
 *
 * This header is the drop-in boundary: plain C, pointers and sizes only, no
 * C++/torch types.  A D program binds it with `extern(C) nothrow @nogc`
 * prototypes (INTEGRATION.md shows the binding file).  Each entry point names
 * the reference interface it replaces (paths relative to the reference repo,
 * source/gamut/...).
 *
 * Conventions (mirroring the reference):
 *   - no exceptions, no aborts: functions return an int status (0 = ok) or a
 *     NULL pointer; gamut_hip_last_error() gives a static/thread-local C string
 *   - PixelType ordinals are exactly types.d:32-59 (l8 = 0 ... rgbapf32 = 17,
 *     unknown = -1); pitches are signed bytes (negative = stored bottom-up)
 *   - "host" entry points take host pointers, are synchronous, and return
 *     malloc()-compatible memory where the reference does (free() it)
 *   - "device" entry points take HBM pointers, enqueue on the given hipStream_t
 *     (passed as void*, NULL = HIP's null stream) and do not synchronise;
 *     results stay resident in HBM
 *   - thread-safe: no mutable global state besides lazily created per-thread
 *     streams; concurrent calls on different images are safe
 */
#ifndef GAMUT_HIP_H
#define GAMUT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes -------------------------------------------------------- */
enum {
    GAMUT_HIP_OK              = 0,
    GAMUT_HIP_ERR_INVALID_ARG = 1,
    GAMUT_HIP_ERR_UNSUPPORTED = 2,   /* e.g. planar/compressed PixelType: scanline.d:82-85 */
    GAMUT_HIP_ERR_OUT_OF_MEMORY = 3,
    GAMUT_HIP_ERR_HIP         = 4,   /* a HIP runtime call failed; see last_error */
    GAMUT_HIP_ERR_DECODE      = 5,   /* corrupt / unsupported stream */
    GAMUT_HIP_ERR_NO_DEVICE   = 6
};

/* ---- PixelType (types.d:32-59) ------------------------------------------- */
enum {
    GAMUT_PIXEL_unknown = -1,
    GAMUT_PIXEL_l8 = 0, GAMUT_PIXEL_l16, GAMUT_PIXEL_lf32,
    GAMUT_PIXEL_la8, GAMUT_PIXEL_la16, GAMUT_PIXEL_laf32,
    GAMUT_PIXEL_lap8, GAMUT_PIXEL_lap16, GAMUT_PIXEL_lapf32,
    GAMUT_PIXEL_rgb8, GAMUT_PIXEL_rgb16, GAMUT_PIXEL_rgbf32,
    GAMUT_PIXEL_rgba8, GAMUT_PIXEL_rgba16, GAMUT_PIXEL_rgbaf32,
    GAMUT_PIXEL_rgbap8, GAMUT_PIXEL_rgbap16, GAMUT_PIXEL_rgbapf32,
    GAMUT_PIXEL_COUNT
};

/* JPEG sampling modes (jpegload.d:117-118, JPEG_SUBSAMPLING) */
enum { GAMUT_JPGD_GRAYSCALE = 0, GAMUT_JPGD_YH1V1, GAMUT_JPGD_YH2V1, GAMUT_JPGD_YH1V2, GAMUT_JPGD_YH2V2 };

/* ---- runtime ------------------------------------------------------------- */
const char* gamut_hip_version(void);
/* number of visible GPUs (0 if none / no driver) */
int  gamut_hip_device_count(void);
/* bind the calling thread to `device` (-1 = keep current). Returns status. */
int  gamut_hip_init(int device);
void gamut_hip_shutdown(void);
/* message for the last non-zero status on this thread ("" if none) */
const char* gamut_hip_last_error(void);

/* device memory / stream helpers so a non-HIP host (D, C) can drive the batched API */
void* gamut_hip_device_malloc(size_t bytes);
void  gamut_hip_device_free(void* p);
void* gamut_hip_host_malloc_pinned(size_t bytes);
void  gamut_hip_host_free_pinned(void* p);
int   gamut_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int   gamut_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
void* gamut_hip_stream_create(void);
void  gamut_hip_stream_destroy(void* stream);
int   gamut_hip_stream_synchronize(void* stream);

/* ---- K8/K9: scanline conversion matrix ------------------------------------
 * replaces scanlinesConvert (scanline.d:70-121) / scanlinesCopy (:37-55).
 * Same argument meaning; the interType/interBuf scratch arguments disappear
 * (the intermediate lives in registers).  Returns GAMUT_HIP_OK where the
 * reference returns true. */
int gamut_hip_pixel_type_size(int type);                       /* types.d:62-86 */
int gamut_hip_scanlines_inter_type(int srcType, int dstType);  /* scanline.d:25-31 */

/* host pointers; synchronous (H2D, kernel, D2H inside) */
int gamut_hip_scanlines_convert(int srcType, const uint8_t* src, int srcPitch,
                                int dstType, uint8_t* dst, int dstPitch,
                                int width, int height);
int gamut_hip_scanlines_copy(int type, const uint8_t* src, int srcPitch,
                             uint8_t* dst, int dstPitch, int width, int height);

/* HBM pointers; layered like Image.convertTo's layer loop (image.d:1273-1311):
 * layer L's first scanline is src + L*srcLayerOffset.  Asynchronous. */
int gamut_hip_scanlines_convert_device(int srcType, const void* src, int64_t srcPitch, int64_t srcLayerOffset,
                                       int dstType, void* dst, int64_t dstPitch, int64_t dstLayerOffset,
                                       int width, int height, int layers, void* stream);

/* Image.flipHorizontal (image.d:1475-1509) / flipVerticalPhysical (:1926-1954) in place: pixel x <-> pixel W - 1 - x of every
 * row, or row y <-> row H - 1 - y, of every layer.  Device pointers + signed pitch + layer offset, asynchronous; the host variant
 * stages one image through the GPU (up, flip, down) like the other host drop-ins. */
int gamut_hip_flip_device(int type, void* data, int64_t pitch, int64_t layerOffset, int width, int height, int layers, int vertical, void* stream);
int gamut_hip_flip(int type, uint8_t* data, int pitch, int width, int height, int vertical);

/* ---- K1-K4: JPEG block reconstruction --------------------------------------
 * replaces transform_mcu / transform_mcu_expand (jpegload.d:2120-2255), the
 * *Convert row functions (:2528-2823) and the output packing of
 * decompress_jpeg_image_from_stream (:3753-3802).
 *
 * Input per image = what decode_next_row (:2405-2525) hands to transform_mcu:
 * de-quantised int16 coefficients in natural order, 64 per block, blocks in
 * MCU order (calc_mcu_block_order :3076-3088: Y.. Cb Cr), MCUs row-major,
 * mcus_per_row = ceil(width / mcu_w), mcus_per_col = ceil(height / mcu_h).
 * max_zag (optional, may be NULL) is m_mcu_block_max_zag per block (:2512);
 * NULL means "dense": identical results for every block whose pass-1 outputs
 * stay below 2^18 in magnitude, which holds for all 8-bit sample data
 * ("sparse IDCT paths").
 * Output: rows of width*out_comps bytes (out_comps 1, 3 or 4: l8 / rgb8 / rgba8,
 * :3761-3801), out_pitch bytes apart. */
typedef struct gamut_hip_jpeg_desc {
    const int16_t* coeffs;      /* HBM */
    const uint8_t* max_zag;     /* HBM or NULL */
    uint8_t*       out;         /* HBM */
    int64_t        out_pitch;   /* bytes */
    int32_t        width, height;
    int32_t        scan_type;   /* GAMUT_JPGD_* */
    int32_t        out_comps;   /* 1, 3, 4 */
} gamut_hip_jpeg_desc;

/* `count` independent images, arbitrary sizes. descs is a HOST array. Async. */
int gamut_hip_jpeg_reconstruct_device(const gamut_hip_jpeg_desc* descs, int count, void* stream);

/* uniform batch: image i uses coeffs + i*coeff_stride (int16 elements),
 * max_zag + i*zag_stride (bytes, ignored if max_zag NULL), out + i*out_stride
 * (bytes).  One launch for the whole batch.  Async. */
int gamut_hip_jpeg_reconstruct_batch_device(const int16_t* coeffs, int64_t coeff_stride,
                                            const uint8_t* max_zag, int64_t zag_stride,
                                            uint8_t* out, int64_t out_pitch, int64_t out_stride,
                                            int width, int height, int scan_type, int out_comps,
                                            int count, void* stream);

/* host-side feeder: entropy decode of one file into the dense form above --
 * baseline SOF0/SOF1 (jpegload.d:1160-1848 markers, :2405-2525 decode_next_row) and
 * progressive SOF2 (:3296-3664 scans into coefficient planes, :2259-2333 hand-over).
 * The frame's buffers are malloc'd; release with gamut_hip_jpeg_frame_free. */
typedef struct gamut_hip_jpeg_frame {
    int32_t  width, height, comps, scan_type;
    int32_t  mcus_per_row, mcus_per_col, blocks_per_mcu;
    int16_t* coeffs;            /* host */
    uint8_t* max_zag;           /* host */
    float    pixel_aspect_ratio, dpi_y;   /* what decompress_jpeg_image_from_stream hands out (jpegload.d:3804-3805): from the last JFIF (APP0) / EXIF (APP1) segment the
                                             decoder met -- in front of the frame, between scans or behind the last MCU row (find_eoi); JFIF without a unit: dpi_y -1;
                                             a file with neither: NaN (the D struct's float members are never assigned, :510-512), NOT the -1 of :3719's comment */
} gamut_hip_jpeg_frame;
int  gamut_hip_jpeg_decode_coeffs(const uint8_t* data, size_t len, gamut_hip_jpeg_frame* out);
void gamut_hip_jpeg_frame_free(gamut_hip_jpeg_frame* f);
/* header walk only (markers up to the first SOS): geometry and JFIF density; coeffs / max_zag stay NULL.  Lets a caller
 * size and lay out device buffers: an image needs mcus_per_row * mcus_per_col * blocks_per_mcu blocks of 64 int16. */
int  gamut_hip_jpeg_read_header(const uint8_t* data, size_t len, gamut_hip_jpeg_frame* out);
/* what the device entropy decoder would be given for this file: the geometry, the number of independently
 * decodable segments (restart intervals, or 1; for a progressive file: summed over its scans) and the size of the
 * unstuffed entropy-coded data with its padding.
 * Host only; the same preparation code gamut_hip_jpeg_entropy_decode_device runs per file. */
int  gamut_hip_jpeg_scan_layout(const uint8_t* data, size_t len, gamut_hip_jpeg_frame* info, int32_t* segments, uint64_t* entropy_bytes);
/* Entropy decode ON THE DEVICE of `count` baseline files given in host memory: the compressed
 * scans are uploaded (unstuffed on host threads straight into pinned memory, DMA on a private copy stream overlapped with
 * the decode of the files already there) and a workgroup of self-synchronising lanes per scan -- a lane per restart
 * interval where the file has short ones -- writes the
 * dense de-quantised coefficient form (what decode_next_row, jpegload.d:2405-2525, leaves per MCU row) into
 * coeffs[coeff_offset[i] ..] (int16 elements) and max_zag[zag_offset[i] ..] (device pointers, caller-sized from
 * gamut_hip_jpeg_read_header).  Progressive (SOF2) files of the batch (jpegload.d:3296-3664) go to the same buffers: their
 * scans are decoded level by level on the device -- every scan of a level of every file in one launch; AC refinement
 * blocks a wave each -- then de-quantised in place (gamut_amd/csrc/jpeg_prog.hpp); fewer than twelve such files per host
 * thread are decoded by the host feeder and uploaded instead (GAMUT_HIP_JPEG_PROGRESSIVE=host / device forces either).
 * info[i] receives the geometry and the density (host), status_host[i] (may be NULL) the per-file verdict, and status_dev[i]
 * (device, may be NULL) is non-zero only for a file the call refuses.  What a kernel cannot vouch for -- a bit pattern no code word
 * begins, a segment that runs out, a wrong or missing RSTn, octets between an interval's last bit and its marker; the reference has a
 * result for most of these (jpgd decodes symbol 0 and carries on, huff_decode :746-813) -- is decoded again by the host feeder
 * (gamut_hip_jpeg_decode_coeffs) behind the device pass and uploaded, so a damaged file costs a host decode, an intact one nothing; the
 * markers behind the scan (find_eoi :2826-2848) are walked on the host from where the device saw the scan end.
 * Returns when the decode has finished on `stream`;
 * the status of the lowest-numbered failing file, GAMUT_HIP_OK if none. */
int  gamut_hip_jpeg_entropy_decode_device(const uint8_t* const* data, const size_t* len, int count,
                                          const int64_t* coeff_offset, const int64_t* zag_offset,
                                          int16_t* coeffs, uint8_t* max_zag, uint32_t* status_dev,
                                          gamut_hip_jpeg_frame* info, int* status_host, void* stream);
/* Files -> pixels: `count` JPEG files in host memory (baseline and progressive, any sampling mode) -> rows of
 * width * req_comps bytes (req_comps = 1 / 3 / 4: l8 / rgb8 / rgba8, as decompress_jpeg_image_from_memory converts) at
 * out + out_offset[i] (device).  The JPEG member of the trio of file-level batch calls (with gamut_hip_png_decode_batch_device and
 * gamut_hip_qoi_decode_batch_device: BASELINE.json config 5 end to end).  It is gamut_hip_jpeg_entropy_decode_device followed by
 * the reconstruction kernels, except that the library owns the coefficient buffers (per-thread device scratch, sized from the
 * headers) and that a group's reconstruction is queued right behind ITS entropy decode, beside the decode of the next group and
 * the upload of the one after.  info[i] receives the geometry, status_host[i] / status_dev[i]
 * (both may be NULL) as above.  Returns when the pixels are in place; the status of the lowest-numbered failing file. */
int  gamut_hip_jpeg_decode_batch_device(const uint8_t* const* data, const size_t* len, int count, int req_comps,
                                        const int64_t* out_offset, uint8_t* out, gamut_hip_jpeg_frame* info, int* status_host,
                                        uint32_t* status_dev, void* stream);
/* the same for `count` independent files on up to `threads` host threads (<= 0: one per hardware thread).  The reference
 * decodes one image at a time (the serial Huffman stage is what bounds a batch once the GPU stages
 * run at HBM speed).  status[i] (may be NULL) receives image i's status and out[i] its frame (zeroed on failure);
 * returns GAMUT_HIP_OK when every image decoded, otherwise the status of the lowest-numbered failing image. */
int  gamut_hip_jpeg_decode_coeffs_batch(const uint8_t* const* data, const size_t* len, int count,
                                        gamut_hip_jpeg_frame* out, int* status, int threads);

/* drop-in for decompress_jpeg_image_from_stream (jpegload.d:3720-3723) on a
 * memory buffer: host entropy decode, GPU reconstruction, malloc'd
 * width*req_comps*height result (NULL on failure).  req_comps -1/1/3/4. */
uint8_t* gamut_hip_decompress_jpeg_image_from_memory(const uint8_t* data, size_t len,
        int* width, int* height, int* actual_comps,
        float* pixelAspectRatio, float* dotsPerInchY, int req_comps);

/* the same with the reference's own argument list (jpegload.d:3720-3723): the input arrives through a JpegStreamReadFunc
 * (jpegload.d:61-70): `int function(void* pBuf, int max_bytes_to_read, bool* pEOF_flag, void* userData)`, -1 = error, called
 * until it raises *pEOF_flag.  D's bool is one byte: unsigned char here.  The stream is pulled in jpgd's own 8 KiB pieces
 * (prep_in_buffer, jpegload.d:1971-2003) and no further call is made once the image's EOI marker has arrived -- an image
 * embedded in a longer stream is over-read by less than one piece, as by the reference -- then decoded as above. */
typedef int (*gamut_hip_jpeg_stream_read_func)(void* pBuf, int max_bytes_to_read, unsigned char* pEOF_flag, void* userData);
uint8_t* gamut_hip_decompress_jpeg_image_from_stream(gamut_hip_jpeg_stream_read_func rfn, void* userData,
        int* width, int* height, int* actual_comps,
        float* pixelAspectRatio, float* dotsPerInchY, int req_comps);

/* ---- K5-K7: PNG de-filter / expand -----------------------------------------
 * replaces stbi__create_png_image_raw (stbdec.d:1406-1635).  raw = inflated
 * stream of one non-interlaced image (or one Adam7 pass): per row one filter
 * byte + ceil(img_n*x*depth/8) bytes.  out = x*y*out_n*(depth==16?2:1) bytes,
 * tightly packed, out_n == img_n or img_n+1 (alpha = 255 inserted), 1/2/4-bit
 * samples expanded (grey scaled when color==0), 16-bit samples native-endian. */
typedef struct gamut_hip_png_desc {
    const uint8_t* raw;         /* HBM */
    uint8_t*       out;         /* HBM */
    uint32_t       raw_len;
    uint32_t       x, y;
    int32_t        img_n, out_n, depth, color;
} gamut_hip_png_desc;
/* status (HBM, may be NULL): one uint32 per image, bit 0 is set when a row carries an invalid
 * filter type (> 4, "Corrupt PNG" stbdec.d:1438); zero it before the call. */
int gamut_hip_png_defilter_device(const gamut_hip_png_desc* descs, int count, uint32_t* status, void* stream);

/* uniform batch: image i uses raw + i*raw_stride, out + i*out_stride (bytes). */
int gamut_hip_png_defilter_batch_device(const uint8_t* raw, int64_t raw_stride, uint32_t raw_len,
                                        uint8_t* out, int64_t out_stride,
                                        uint32_t x, uint32_t y, int img_n, int out_n, int depth, int color,
                                        int count, uint32_t* status, void* stream);

/* drop-ins for stbi_load_from_callbacks / stbi_load_16_from_callbacks
 * (stbdec.d:713-735) on a memory buffer: host chunk parse + inflate, GPU
 * de-filter/expand/post passes, malloc'd result (NULL on failure). */
uint8_t*  gamut_hip_stbi_load_from_memory(const uint8_t* data, size_t len, int* x, int* y, int* comp, int req_comp,
                                          float* ppmX, float* ppmY, float* pixelRatio);
uint16_t* gamut_hip_stbi_load_16_from_memory(const uint8_t* data, size_t len, int* x, int* y, int* comp, int req_comp,
                                             float* ppmX, float* ppmY, float* pixelRatio);
/* stbi__png_is16 (stbdec.d:2091-2109) */
int gamut_hip_png_is16(const uint8_t* data, size_t len);
/* the same three with the reference's own argument lists: stbi_io_callbacks (stbdec.d:408-419) + the user pointer, then
 * exactly the parameters of stbi_load_from_callbacks / stbi_load_16_from_callbacks (stbdec.d:713-735).  The stream is walked
 * chunk by chunk as stbi__parse_png_file does (stbdec.d:1777-2023): `read` for chunk headers and the chunks the parser looks
 * at (a read of 0 bytes ends the data, as in stbi__refill_buffer :780-795), `skip` for every other ancillary chunk
 * (stbi__skip :822-842; NULL = read and drop), `eof` for the file without IEND (:2008-2012; NULL = "not yet").  The walk stops
 * behind IEND's CRC and leaves the stream there (stb's 128-byte buffer leaves it up to 127 bytes further): a second image may
 * follow in the same stream.  is16 reads up to IHDR only; the caller rewinds its stream before loading, as
 * plugins/png.d:50-62 does. */
typedef struct gamut_hip_stbi_io_callbacks {
    int  (*read)(void* user, char* data, int size);
    void (*skip)(void* user, int n);
    int  (*eof)(void* user);
} gamut_hip_stbi_io_callbacks;
uint8_t*  gamut_hip_stbi_load_from_callbacks(const gamut_hip_stbi_io_callbacks* clbk, void* user, int* x, int* y, int* comp, int req_comp,
                                             float* ppmX, float* ppmY, float* pixelRatio);
uint16_t* gamut_hip_stbi_load_16_from_callbacks(const gamut_hip_stbi_io_callbacks* clbk, void* user, int* x, int* y, int* comp, int req_comp,
                                                float* ppmX, float* ppmY, float* pixelRatio);
int gamut_hip_stbi_png_is16_from_callbacks(const gamut_hip_stbi_io_callbacks* clbk, void* user);

/* ---- inflate on the GPU -------------------------------------------------------------------
 * replaces, for batches, stbi_zlib_decode_malloc_guesssize_headerflag (stbdec.d:1267-1321 -> miniz).  Stream i is the raw
 * DEFLATE data src[0 .. src_len) in HBM (a zlib stream without its 2-byte header, which the host checks as stbdec.d does;
 * the adler32 trailer is not read, as with the reference's trusted_input); its bytes go to dst[0 .. dst_cap) in HBM -- output
 * beyond dst_cap is dropped (the de-filter needs (bytes per line + 1) * height, no more), but the stream is still decoded to its
 * end: damage anywhere makes it a corrupt stream, as for the reference, which inflates all of it.
 * out_len_dev[i] = bytes written, status_dev[i] = 0 or the reason the stream is corrupt (GAMUT_HIP_INFLATE_E_*); both are
 * device arrays, the call is asynchronous on `stream`.  One 1024-thread workgroup per stream: speculative parallel Huffman
 * decode into a token list, then parallel match resolution tile by tile (gamut_amd/csrc/inflate.hip); the library keeps 256 KiB
 * of HBM scratch per RESIDENT workgroup (two per compute unit: 128 MiB on MI355X however many streams a batch has) plus small per-stream
 * tables, per (thread, device, stream).  descs is a host array.  One deviation from zlib, as a
 * bound on the work a hostile stream can demand: a stream with more than 4096 + src_len / 8 blocks (no encoder comes near:
 * that is a block per 8 compressed bytes) is reported as GAMUT_HIP_INFLATE_E_INPUT. */
typedef struct gamut_hip_inflate_desc { const uint8_t* src; uint8_t* dst; uint32_t src_len, dst_cap; } gamut_hip_inflate_desc;
enum { GAMUT_HIP_INFLATE_E_BLOCK_TYPE = 1, GAMUT_HIP_INFLATE_E_STORED = 2, GAMUT_HIP_INFLATE_E_LENGTHS = 3, GAMUT_HIP_INFLATE_E_CODE = 4,
       GAMUT_HIP_INFLATE_E_DISTANCE = 5, GAMUT_HIP_INFLATE_E_INPUT = 6 };
int gamut_hip_inflate_batch_device(const gamut_hip_inflate_desc* descs, int count, uint32_t* out_len_dev, uint32_t* status_dev, void* stream);
/* The same in several launches, each allowed `slice_bytes` more of every stream than the one before: a stream stops in front of what it
 * cannot finish within the bytes it may read and goes on from its saved state (position, code lengths of the block it stands in; the
 * window comes back from its output).  This is how gamut_hip_png_decode_batch_device inflates while the files are still on their way
 * over PCIe; here all bytes are in HBM already -- same results as the call above, for tests and measurements (this one returns when the
 * last launch has finished). */
int gamut_hip_inflate_batch_device_sliced(const gamut_hip_inflate_desc* descs, int count, uint32_t* out_len_dev, uint32_t* status_dev, uint32_t slice_bytes, void* stream);

/* ---- PNG files in batches ----------------------------------------------------------------------------------- */
typedef struct gamut_hip_png_info {
    uint32_t width, height;
    int32_t  channels_in_file;      /* what stbi reports as *comp: 1..4 (palette images: 3 or 4) */
    int32_t  channels;              /* of the decoded pixels: req_comp, or channels_in_file when req_comp == 0 */
    int32_t  bits;                  /* of the decoded samples: 8 or 16 */
    float    pixels_per_meter_x, pixels_per_meter_y, pixel_aspect_ratio;   /* pHYs, -1 when absent */
} gamut_hip_png_info;
/* IHDR only (host): width, height, bit depth class and the channel count implied by the colour type */
int gamut_hip_png_read_header(const uint8_t* data, size_t len, gamut_hip_png_info* info);
/* `count` PNG files in host memory -> pixels at out + out_offset[i] (device): chunk walk + inflate on up to `threads` host
 * threads (<= 0: one per hardware thread; the inflate is what bounds a PNG pipeline), the inflated streams going to the
 * device as they finish; then the rest of stbi__do_png on the GPU -- files that only need de-filter + expand are
 * de-filtered together, one launch per geometry (one workgroup per image), the others stage by stage per file.  Batches of
 * more files than host threads inflate on the GPU instead (gamut_hip_inflate_batch_device: the threads then only walk
 * the chunks and gather the IDAT bytes for one upload); the environment variable GAMUT_HIP_PNG_INFLATE=host / device forces
 * either.  req_comp as in stbi_load (0 = as in the file), bits = 8 / 16 as stbi_load / stbi_load_16 convert, 0 = as
 * in the file.  info[i] / status_host[i] (may be NULL) per file; returns the status of the lowest-numbered failing file. */
int gamut_hip_png_decode_batch_device(const uint8_t* const* data, const size_t* len, int count, int req_comp, int bits,
                                      const int64_t* out_offset, uint8_t* out, gamut_hip_png_info* info, int* status_host,
                                      int threads, void* stream);

/* ---- QOI (codecs/qoi.d) -- SURVEY.md 8f row N4 ---------------------------------------------------------------- */
typedef struct gamut_hip_qoi_desc { uint32_t width, height; uint8_t channels, colorspace; } gamut_hip_qoi_desc;   /* qoi_desc, decode fields */
/* drop-in for qoi_decode (qoi.d:448-550): channels = 0 (as in the file), 3 or 4; malloc'd width*height*channels bytes, or
 * NULL with the same header checks (:458-480).  Decoded on the GPU (one lane per stream). */
void* gamut_hip_qoi_decode(const void* data, int size, gamut_hip_qoi_desc* desc, int channels);
/* header only (host) */
int   gamut_hip_qoi_read_header(const void* data, int size, gamut_hip_qoi_desc* desc);
/* batch: `count` streams in host memory -> pixels at out + out_offset[i] (device), one lane per image.  descs[i] (host)
 * receives the headers, status_host[i] (may be NULL) the per-file header status.  Returns when the decode has finished. */
int   gamut_hip_qoi_decode_batch_device(const uint8_t* const* data, const int* size, int count, int channels,
                                        const int64_t* out_offset, uint8_t* out, gamut_hip_qoi_desc* descs, int* status_host, void* stream);
/* the same decode for files that are already resident in HBM (a device-side file cache; bench.py's mixed workload): file i
 * is blob[begin[i] .. begin[i] + size[i]) and must be followed by GAMUT_HIP_QOI_SLACK readable bytes inside the blob (the
 * lanes read whole 64-byte blocks); begin / size / descs (from gamut_hip_qoi_read_header) / out_offset are host arrays,
 * read before the call returns.  Asynchronous on `stream`, like the other device entry points. */
#define GAMUT_HIP_QOI_SLACK 160
int   gamut_hip_qoi_decode_resident_device(const uint8_t* blob, int64_t blob_len, const int64_t* begin, const int* size,
                                           const gamut_hip_qoi_desc* descs, int count, int channels, const int64_t* out_offset,
                                           uint8_t* out, void* stream);

/* ---- files of any of the three formats, one call ---------------------------------------------------------------------
 * The reference loads any file through Image.loadFromMemory: identifyFormatFromStream (image.d:1045-1061 -- the plugins' detect
 * procedures, a signature test each: plugins/jpeg.d:106-110, png.d:165-169, qoi.d:143-147) picks g_plugins[fif].loadProc
 * (image.d:1751-1772).  gamut_hip_identify_format is that test (GAMUT_HIP_FORMAT_*, the ImageFormat ordinals of types.d:14-21, or
 * GAMUT_HIP_FORMAT_UNKNOWN); gamut_hip_decode_batch_device takes `count` files of ANY of the three formats in host memory and
 * leaves rows of width * req_comps bytes (req_comps = 3 / 4: rgb8 / rgba8, what all three decoders produce; 16-bit PNG samples
 * are reduced as stbi_load does) at out + out_offset[i] (device) -- BASELINE.json config 5 from files in one call.  It is the three
 * per-format batch calls run SIDE BY SIDE, each on a stream of its own behind `stream` and on a worker thread of the library: the
 * PNG leg is bound by the inflate kernels while the QOI leg is bound by PCIe, and the JPEG leg is short.  info[i] receives format
 * and geometry, status_host[i] (may be NULL) the file's status (GAMUT_HIP_ERR_UNSUPPORTED: format not identified).  Returns when
 * the pixels are in place; the status of the lowest-numbered failing file, GAMUT_HIP_OK if none. */
enum { GAMUT_HIP_FORMAT_UNKNOWN = -1, GAMUT_HIP_FORMAT_JPEG = 0, GAMUT_HIP_FORMAT_PNG = 1, GAMUT_HIP_FORMAT_QOI = 2 };
typedef struct gamut_hip_image_info { int32_t format, width, height, channels_in_file, channels; } gamut_hip_image_info;
int gamut_hip_identify_format(const uint8_t* data, size_t len);
int gamut_hip_decode_batch_device(const uint8_t* const* data, const size_t* len, int count, int req_comps,
                                  const int64_t* out_offset, uint8_t* out, gamut_hip_image_info* info, int* status_host, void* stream);

/* ---- multi-GPU (north_star: "image-index round-robin, RCCL over xGMI only for the gather of decoded outputs") --
 * One process per GPU.  Images are independent, so the data path has no collective: image i of a batch belongs to rank
 * i % world and is the (i / world)-th image that rank holds.  The reference has no counterpart (single-threaded library);
 * this is what gamut_amd/shard.py does, for a host that is not Python. */
int     gamut_hip_shard_owner(int64_t image_index, int world);                    /* rank that decodes image i */
int64_t gamut_hip_shard_count(int rank, int world, int64_t total_images);         /* images a rank holds */
int64_t gamut_hip_shard_local_index(int64_t image_index, int world);              /* position among its owner's images */
int64_t gamut_hip_shard_global_index(int64_t local_index, int rank, int world);   /* and back */
/* worker threads the host-side feeders start by default: the cores the process may use (cgroup quota) divided by the
 * ranks on this node (LOCAL_WORLD_SIZE / OMPI_COMM_WORLD_LOCAL_SIZE), or GAMUT_HIP_HOST_THREADS */
int     gamut_hip_host_threads(void);

/* The one exchange: gathering decoded outputs.  An RCCL communicator behind an opaque handle; librccl is loaded at run time.
 * Rank 0 calls comm_get_unique_id (128 bytes) and hands the bytes to the other ranks by whatever channel the host has;
 * every rank then calls comm_init with the device it decodes on current (gamut_hip_init).  world == 1 needs no id and no RCCL. */
#define GAMUT_HIP_COMM_ID_BYTES 128
typedef struct gamut_hip_comm gamut_hip_comm;
int  gamut_hip_comm_get_unique_id(void* id128);
int  gamut_hip_comm_init(gamut_hip_comm** comm, int world, int rank, const void* id128);
void gamut_hip_comm_destroy(gamut_hip_comm* comm);
int  gamut_hip_comm_rank(const gamut_hip_comm* comm);
int  gamut_hip_comm_world(const gamut_hip_comm* comm);
/* `local` (HBM): this rank's decoded images, the k-th one at local + k * local_stride, bytes_per_image each.  Image i of
 * the batch arrives at dst + i * dst_stride on `root` (root = -1: on every rank, an all-gather).  Grouped ncclSend /
 * ncclRecv per image on `stream` (asynchronous); a rank's own images are copied device to device.  Collective: every rank
 * of the communicator must call it with the same total_images / bytes_per_image / root. */
int  gamut_hip_gather_outputs_device(gamut_hip_comm* comm, const void* local, int64_t local_stride, int64_t bytes_per_image,
                                     int64_t total_images, void* dst, int64_t dst_stride, int root, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GAMUT_HIP_H */
