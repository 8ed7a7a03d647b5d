/* A plain C99 consumer of include/gamut_hip.h + include/gamut_image.h: what a D (or any FFI) binding sees.  Built and run by
 * tests/test_capi_cpu.py without a GPU: host-only calls must work, compute calls must report GAMUT_HIP_ERR_NO_DEVICE. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gamut_hip.h"
#include "gamut_image.h"

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    static unsigned char buf[1 << 20];
    const size_t n = fread(buf, 1, sizeof(buf), f);
    fclose(f);

    gamut_hip_jpeg_frame hdr;
    if (gamut_hip_jpeg_read_header(buf, n, &hdr) != GAMUT_HIP_OK) { printf("header: %s\n", gamut_hip_last_error()); return 4; }
    gamut_hip_jpeg_frame fr;
    if (gamut_hip_jpeg_decode_coeffs(buf, n, &fr) != GAMUT_HIP_OK) { printf("decode: %s\n", gamut_hip_last_error()); return 5; }
    long nonzero = 0;
    const long total = (long)fr.mcus_per_row * fr.mcus_per_col * fr.blocks_per_mcu * 64;
    for (long i = 0; i < total; ++i) nonzero += fr.coeffs[i] != 0;
    printf("%dx%d comps=%d scan_type=%d blocks=%ld nonzero=%ld format=%d\n", fr.width, fr.height, fr.comps, fr.scan_type, total / 64, nonzero,
           gamut_identify_format_from_memory(buf, n));
    gamut_hip_jpeg_frame_free(&fr);

    if (gamut_hip_device_count() == 0) {                       /* no GPU: loud failure, nothing computed on the CPU */
        unsigned char src[16] = { 0 }, dst[64];
        memset(dst, 0xA5, sizeof(dst));
        const int rc = gamut_hip_scanlines_convert(GAMUT_PIXEL_rgba8, src, 16, GAMUT_PIXEL_rgbaf32, dst, 64, 4, 1);
        if (rc != GAMUT_HIP_ERR_NO_DEVICE || dst[0] != 0xA5) return 6;
        gamut_image* img = gamut_image_new();
        if (gamut_image_load_from_memory(img, buf, n, 0) || !gamut_image_is_error(img)) return 7;
        gamut_image_delete(img);
    }
    return hdr.width == fr.width ? 0 : 8;
}
