/*
 * gamut_image.h -- C ABI of the host-side mirror of Gamut's `Image` for the GPU path.
 *
 * The reference's host API is D (`struct Image`, source/gamut/image.d).  No D compiler exists in the
 * build image, so the same interface is provided in C++ (gamut_amd/csrc/image_host.hip) over the
 * kernels' ABI (gamut_hip.h) and exported here as C functions named after the D members they mirror:
 * same argument meaning, same state machine (error state = static C string + type unknown,
 * image.d:1563-1570), same storage rules (allocatePixelStorage, internals/types.d:355-540), same
 * LoadFlags / LayoutConstraints bits (types.d:139-348).  Pixel storage is host malloc memory exactly as
 * in the reference (the user may disown and free() it); every pixel operation (decode, convertTo) runs
 * on the GPU through gamut_hip_* -- there is no CPU pixel path.
 * Formats: JPEG (baseline) and PNG, i.e. the path of this project; other signatures report
 * "Unidentified image format".
 */
#ifndef GAMUT_IMAGE_H
#define GAMUT_IMAGE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ImageFormat (types.d:14-28) */
enum { GAMUT_FORMAT_unknown = -1, GAMUT_FORMAT_JPEG = 0, GAMUT_FORMAT_PNG = 1, GAMUT_FORMAT_QOI = 2 };   /* ImageFormat, types.d:14-21 */

/* LoadFlags (types.d:139-197) */
enum {
    GAMUT_LOAD_NORMAL = 0, GAMUT_LOAD_GREYSCALE = 0x10000, GAMUT_LOAD_ALPHA = 0x20000, GAMUT_LOAD_NO_ALPHA = 0x40000,
    GAMUT_LOAD_RGB = 0x80000, GAMUT_LOAD_8BIT = 0x100000, GAMUT_LOAD_16BIT = 0x200000, GAMUT_LOAD_FP32 = 0x400000,
    GAMUT_LOAD_NO_PIXELS = 0x800000, GAMUT_LOAD_PREMUL = 0x1000000, GAMUT_LOAD_NO_PREMUL = 0x2000000
};
/* LayoutConstraints (types.d:266-348), a ushort */
enum {
    GAMUT_LAYOUT_DEFAULT = 0,
    GAMUT_LAYOUT_MULTIPLICITY_1 = 0, GAMUT_LAYOUT_MULTIPLICITY_2 = 1, GAMUT_LAYOUT_MULTIPLICITY_4 = 2, GAMUT_LAYOUT_MULTIPLICITY_8 = 3,
    GAMUT_LAYOUT_TRAILING_0 = 0, GAMUT_LAYOUT_TRAILING_1 = 4, GAMUT_LAYOUT_TRAILING_3 = 8, GAMUT_LAYOUT_TRAILING_7 = 12,
    GAMUT_LAYOUT_SCANLINE_ALIGNED_1 = 0, GAMUT_LAYOUT_SCANLINE_ALIGNED_2 = 16, GAMUT_LAYOUT_SCANLINE_ALIGNED_4 = 32,
    GAMUT_LAYOUT_SCANLINE_ALIGNED_8 = 48, GAMUT_LAYOUT_SCANLINE_ALIGNED_16 = 64, GAMUT_LAYOUT_SCANLINE_ALIGNED_32 = 80,
    GAMUT_LAYOUT_SCANLINE_ALIGNED_64 = 96, GAMUT_LAYOUT_SCANLINE_ALIGNED_128 = 112,
    GAMUT_LAYOUT_BORDER_0 = 0, GAMUT_LAYOUT_BORDER_1 = 128, GAMUT_LAYOUT_BORDER_2 = 256, GAMUT_LAYOUT_BORDER_3 = 384,
    GAMUT_LAYOUT_VERT_FLIPPED = 512, GAMUT_LAYOUT_VERT_STRAIGHT = 1024, GAMUT_LAYOUT_GAPLESS = 2048
};
/* ops of gamut_convert_pixel_type (types.d:351-602) */
enum { GAMUT_TO_GREYSCALE = 0, GAMUT_TO_RGB, GAMUT_TO_ADD_ALPHA, GAMUT_TO_DROP_ALPHA, GAMUT_TO_PREMUL, GAMUT_TO_NO_PREMUL,
       GAMUT_TO_8BIT, GAMUT_TO_16BIT, GAMUT_TO_FP32 };

typedef struct gamut_image gamut_image;

/* free functions */
int  gamut_convert_pixel_type(int type, int op);                       /* convertPixelTypeTo* */
int  gamut_apply_load_flags(int type, int flags);                      /* internals/types.d:627-661 */
int  gamut_compute_requested_image_components(int flags);              /* internals/types.d:587-609 */
int  gamut_valid_load_flags(int flags);                                /* internals/types.d:563-578 */
int  gamut_layout_constraints_valid(int constraints);                  /* internals/types.d:267-289 */
int  gamut_layout_constraints_compatible(int newer, int older);        /* internals/types.d:241-264 */
int  gamut_identify_format_from_memory(const uint8_t* bytes, size_t len);   /* image.d:1038-1061 (JPEG, PNG, QOI) */
void gamut_free_image_data(void* mallocArea);                          /* freeImageData, image.d:27-30 */

/* lifetime: a new image is Image.init = errored with "Uninitialized image" (image.d:1609-1613) */
gamut_image* gamut_image_new(void);
void         gamut_image_delete(gamut_image* img);                     /* ~this: frees owned storage */

/* creation (image.d:565-617, 760-789); return 1 on success like the D members that return bool */
int gamut_image_create(gamut_image* img, int width, int height, int type, int layout);
int gamut_image_create_layered(gamut_image* img, int width, int height, int layers, int type, int layout);
int gamut_image_create_no_init(gamut_image* img, int width, int height, int type, int layout);
int gamut_image_create_layered_no_init(gamut_image* img, int width, int height, int layers, int type, int layout);
int gamut_image_create_with_no_data(gamut_image* img, int width, int height, int type, int layout);
/* createView (image.d:697): borrow caller memory; pitch may be negative */
int gamut_image_create_view(gamut_image* img, void* data, int width, int height, int type, int pitchInBytes);

/* load (image.d:886-906): flags = LOAD_* | LAYOUT_*; returns isValid() */
int gamut_image_load_from_memory(gamut_image* img, const uint8_t* bytes, size_t len, int flags);

/* conversion (image.d:1082-1332): all forward to convertTo */
int gamut_image_convert_to(gamut_image* img, int targetType, int layout);
int gamut_image_set_layout(gamut_image* img, int layout);
int gamut_image_convert_op(gamut_image* img, int op, int layout);      /* convertToGreyscale ... convertToFP32 by GAMUT_TO_* */
int gamut_image_convert_to_greyscale_alpha(gamut_image* img, int layout);
int gamut_image_convert_to_rgba(gamut_image* img, int layout);
int gamut_image_flip_vertical(gamut_image* img);                       /* image.d:1524: the logical flip (pitch negated), or -- when a LAYOUT_VERT_* constraint
                                                                          pins the storage order -- flipVerticalPhysical (:1926-1954), rows swapped on the GPU */
int gamut_image_flip_horizontal(gamut_image* img);                     /* image.d:1475-1509, pixels swapped on the GPU */
/* views and copies (image.d:645-679, 706-752, 795-841).  The returned images are new objects (gamut_image_delete them): a view
 * borrows the pixels (not owned, LAYOUT_DEFAULT), a clone owns its own with the source's layout constraints; both are errored
 * images when the arguments are not what the reference asserts. */
gamut_image* gamut_image_layer_range(gamut_image* img, int layerStart, int layerEnd);     /* layer(i) = layer_range(i, i + 1) */
int gamut_image_create_layered_view(gamut_image* img, void* data, int width, int height, int layers, int type, int pitchInBytes, int layerOffsetBytes);
gamut_image* gamut_image_clone(gamut_image* img);
int gamut_image_copy_pixels_to(gamut_image* img, gamut_image* dst);

/* state (image.d:97-551, 1405-1459) */
int   gamut_image_type(const gamut_image* img);
int   gamut_image_width(const gamut_image* img);
int   gamut_image_height(const gamut_image* img);
int   gamut_image_layers(const gamut_image* img);
int   gamut_image_pitch_in_bytes(const gamut_image* img);
int   gamut_image_layer_offset_in_bytes(const gamut_image* img);
int   gamut_image_scanline_in_bytes(const gamut_image* img);
int   gamut_image_layout_constraints(const gamut_image* img);
int   gamut_image_is_error(const gamut_image* img);
int   gamut_image_is_valid(const gamut_image* img);
const char* gamut_image_error_message(const gamut_image* img);         /* NULL when valid */
int   gamut_image_has_data(const gamut_image* img);
int   gamut_image_is_owned(const gamut_image* img);
int   gamut_image_is_stored_upside_down(const gamut_image* img);
float gamut_image_pixel_aspect_ratio(const gamut_image* img);
float gamut_image_dots_per_inch_y(const gamut_image* img);
uint8_t* gamut_image_scanptr(gamut_image* img, int y);                 /* layer 0 */
uint8_t* gamut_image_layerptr(gamut_image* img, int layer, int y);
uint8_t* gamut_image_disown_data(gamut_image* img);                    /* image.d:483-490 */

/* ---- device-resident pixel storage (an extension: the reference keeps pixels in host memory) --------------------
 * With it on, create*() allocates in HBM, loadFromMemory decodes straight into HBM (JPEG entropy decode included for
 * baseline files) and convertTo / setLayout chains run device to device; scanptr() / layerptr() return DEVICE addresses.
 * Can only be switched while the image owns no pixels; returns 0 otherwise or when there is no GPU. */
int gamut_image_set_device_storage(gamut_image* img, int on);
int gamut_image_is_device(const gamut_image* img);
/* one layer, logical top-down order, into host rows of dst_pitch bytes -- for host and device images alike */
int gamut_image_copy_pixels_to_host(gamut_image* img, int layer, void* dst, int64_t dst_pitch);

#ifdef __cplusplus
}
#endif
#endif
