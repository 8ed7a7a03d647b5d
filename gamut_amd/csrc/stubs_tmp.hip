#include "common.hpp"
using namespace gamut;
extern "C" {
int gamut_hip_png_defilter_device(const gamut_hip_png_desc*, int, void*) { return set_error(GAMUT_HIP_ERR_UNSUPPORTED, "not built yet"); }
int gamut_hip_png_defilter_batch_device(const uint8_t*, int64_t, uint32_t, uint8_t*, int64_t, uint32_t, uint32_t, int, int, int, int, int, void*) { return set_error(GAMUT_HIP_ERR_UNSUPPORTED, "not built yet"); }
uint8_t* gamut_hip_stbi_load_from_memory(const uint8_t*, size_t, int*, int*, int*, int, float*, float*, float*) { return nullptr; }
uint16_t* gamut_hip_stbi_load_16_from_memory(const uint8_t*, size_t, int*, int*, int*, int, float*, float*, float*) { return nullptr; }
int gamut_hip_png_is16(const uint8_t*, size_t) { return 0; }
}
