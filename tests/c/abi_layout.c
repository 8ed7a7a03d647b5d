/* abi_layout.c -- prints sizeof / offsetof of every struct include/gamut_hip.h declares, one line per struct:
 *     <struct> <sizeof> <field>=<offset> ...
 * tests/test_capi_cpu.py::test_d_binding_struct_layouts compares these numbers (the C compiler's) with the `static assert`s in
 * bindings/gamut_hip.d (what a D compiler will check when the binding is first built) and with the layout the D declarations
 * of that file yield under the C ABI rules D's extern(C) structs follow.  gcc -std=gnu99 -I include tests/c/abi_layout.c */
#include <stddef.h>
#include <stdio.h>
#include "gamut_hip.h"

#define S(T) printf("\n%s %zu", #T, sizeof(T))
#define F(T, f) printf(" %s=%zu", #f, offsetof(T, f))

int main(void)
{
    S(gamut_hip_jpeg_desc); F(gamut_hip_jpeg_desc, coeffs); F(gamut_hip_jpeg_desc, max_zag); F(gamut_hip_jpeg_desc, out); F(gamut_hip_jpeg_desc, out_pitch);
    F(gamut_hip_jpeg_desc, width); F(gamut_hip_jpeg_desc, height); F(gamut_hip_jpeg_desc, scan_type); F(gamut_hip_jpeg_desc, out_comps);
    S(gamut_hip_jpeg_frame); F(gamut_hip_jpeg_frame, width); F(gamut_hip_jpeg_frame, height); F(gamut_hip_jpeg_frame, comps); F(gamut_hip_jpeg_frame, scan_type);
    F(gamut_hip_jpeg_frame, mcus_per_row); F(gamut_hip_jpeg_frame, mcus_per_col); F(gamut_hip_jpeg_frame, blocks_per_mcu); F(gamut_hip_jpeg_frame, coeffs);
    F(gamut_hip_jpeg_frame, max_zag); F(gamut_hip_jpeg_frame, pixel_aspect_ratio); F(gamut_hip_jpeg_frame, dpi_y);
    S(gamut_hip_png_desc); F(gamut_hip_png_desc, raw); F(gamut_hip_png_desc, out); F(gamut_hip_png_desc, raw_len); F(gamut_hip_png_desc, x); F(gamut_hip_png_desc, y);
    F(gamut_hip_png_desc, img_n); F(gamut_hip_png_desc, out_n); F(gamut_hip_png_desc, depth); F(gamut_hip_png_desc, color);
    S(gamut_hip_stbi_io_callbacks); F(gamut_hip_stbi_io_callbacks, read); F(gamut_hip_stbi_io_callbacks, skip); F(gamut_hip_stbi_io_callbacks, eof);
    S(gamut_hip_inflate_desc); F(gamut_hip_inflate_desc, src); F(gamut_hip_inflate_desc, dst); F(gamut_hip_inflate_desc, src_len); F(gamut_hip_inflate_desc, dst_cap);
    S(gamut_hip_png_info); F(gamut_hip_png_info, width); F(gamut_hip_png_info, height); F(gamut_hip_png_info, channels_in_file); F(gamut_hip_png_info, channels);
    F(gamut_hip_png_info, bits); F(gamut_hip_png_info, pixels_per_meter_x); F(gamut_hip_png_info, pixels_per_meter_y); F(gamut_hip_png_info, pixel_aspect_ratio);
    S(gamut_hip_qoi_desc); F(gamut_hip_qoi_desc, width); F(gamut_hip_qoi_desc, height); F(gamut_hip_qoi_desc, channels); F(gamut_hip_qoi_desc, colorspace);
    S(gamut_hip_image_info); F(gamut_hip_image_info, format); F(gamut_hip_image_info, width); F(gamut_hip_image_info, height);
    F(gamut_hip_image_info, channels_in_file); F(gamut_hip_image_info, channels);
    printf("\n");
    return 0;
}
