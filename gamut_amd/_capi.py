"""ctypes binding of libgamut_hip.so (include/gamut_hip.h).

The product path: there is NO CPU fallback.  If the HIP library is missing the
import fails loudly; if no GPU is present every compute entry point returns
GAMUT_HIP_ERR_NO_DEVICE / a HIP error and `check()` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GAMUT_HIP_LIB") or os.path.join(_HERE, "lib", "libgamut_hip.so")

OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_OUT_OF_MEMORY, ERR_HIP, ERR_DECODE, ERR_NO_DEVICE = range(7)


class GamutHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"gamut_hip status {status}: {message}")
        self.status = status


class JpegDesc(C.Structure):
    _fields_ = [("coeffs", C.c_void_p), ("max_zag", C.c_void_p), ("out", C.c_void_p), ("out_pitch", C.c_int64),
                ("width", C.c_int32), ("height", C.c_int32), ("scan_type", C.c_int32), ("out_comps", C.c_int32)]


class PngInfo(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("channels_in_file", C.c_int32), ("channels", C.c_int32), ("bits", C.c_int32),
                ("pixels_per_meter_x", C.c_float), ("pixels_per_meter_y", C.c_float), ("pixel_aspect_ratio", C.c_float)]


class QoiDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint8), ("colorspace", C.c_uint8)]


class JpegFrame(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("comps", C.c_int32), ("scan_type", C.c_int32),
                ("mcus_per_row", C.c_int32), ("mcus_per_col", C.c_int32), ("blocks_per_mcu", C.c_int32),
                ("coeffs", C.POINTER(C.c_int16)), ("max_zag", C.POINTER(C.c_uint8)),
                ("pixel_aspect_ratio", C.c_float), ("dpi_y", C.c_float)]


class ImageInfo(C.Structure):                                  # gamut_hip_image_info
    _fields_ = [("format", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("channels_in_file", C.c_int32), ("channels", C.c_int32)]


class InflateDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("src_len", C.c_uint32), ("dst_cap", C.c_uint32)]


class PngDesc(C.Structure):
    _fields_ = [("raw", C.c_void_p), ("out", C.c_void_p), ("raw_len", C.c_uint32), ("x", C.c_uint32), ("y", C.c_uint32),
                ("img_n", C.c_int32), ("out_n", C.c_int32), ("depth", C.c_int32), ("color", C.c_int32)]


# name -> (restype, argtypes); mirrors include/gamut_hip.h one to one
_vp, _i, _i64, _u32, _sz, _f = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_size_t, C.c_float
_pi, _pf = C.POINTER(C.c_int), C.POINTER(C.c_float)
SIGNATURES = {
    "gamut_hip_version": (C.c_char_p, []),
    "gamut_hip_device_count": (_i, []),
    "gamut_hip_init": (_i, [_i]),
    "gamut_hip_shutdown": (None, []),
    "gamut_hip_last_error": (C.c_char_p, []),
    "gamut_hip_device_malloc": (_vp, [_sz]),
    "gamut_hip_device_free": (None, [_vp]),
    "gamut_hip_host_malloc_pinned": (_vp, [_sz]),
    "gamut_hip_host_free_pinned": (None, [_vp]),
    "gamut_hip_memcpy_h2d": (_i, [_vp, _vp, _sz, _vp]),
    "gamut_hip_memcpy_d2h": (_i, [_vp, _vp, _sz, _vp]),
    "gamut_hip_stream_create": (_vp, []),
    "gamut_hip_stream_destroy": (None, [_vp]),
    "gamut_hip_stream_synchronize": (_i, [_vp]),
    "gamut_hip_pixel_type_size": (_i, [_i]),
    "gamut_hip_scanlines_inter_type": (_i, [_i, _i]),
    "gamut_hip_scanlines_convert": (_i, [_i, _vp, _i, _i, _vp, _i, _i, _i]),
    "gamut_hip_scanlines_copy": (_i, [_i, _vp, _i, _vp, _i, _i, _i]),
    "gamut_hip_scanlines_convert_device": (_i, [_i, _vp, _i64, _i64, _i, _vp, _i64, _i64, _i, _i, _i, _vp]),
    "gamut_hip_jpeg_reconstruct_device": (_i, [C.POINTER(JpegDesc), _i, _vp]),
    "gamut_hip_jpeg_reconstruct_batch_device": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i, _i, _i, _i, _i, _vp]),
    "gamut_hip_jpeg_decode_coeffs": (_i, [_vp, _sz, C.POINTER(JpegFrame)]),
    "gamut_hip_png_read_header": (_i, [_vp, _sz, C.POINTER(PngInfo)]),
    "gamut_hip_png_decode_batch_device": (_i, [C.POINTER(_vp), C.POINTER(_sz), _i, _i, _i, C.POINTER(_i64), _vp, C.POINTER(PngInfo), C.POINTER(_i), _i, _vp]),
    "gamut_hip_qoi_decode": (_vp, [_vp, _i, C.POINTER(QoiDesc), _i]),
    "gamut_hip_qoi_read_header": (_i, [_vp, _i, C.POINTER(QoiDesc)]),
    "gamut_hip_qoi_decode_batch_device": (_i, [C.POINTER(_vp), C.POINTER(_i), _i, _i, C.POINTER(_i64), _vp, C.POINTER(QoiDesc), C.POINTER(_i), _vp]),
    "gamut_hip_qoi_decode_resident_device": (_i, [_vp, _i64, C.POINTER(_i64), C.POINTER(_i), C.POINTER(QoiDesc), _i, _i, C.POINTER(_i64), _vp, _vp]),
    "gamut_hip_flip_device": (_i, [_i, _vp, _i64, _i64, _i, _i, _i, _i, _vp]),
    "gamut_hip_flip": (_i, [_i, _vp, _i, _i, _i, _i]),
    "gamut_hip_jpeg_read_header": (_i, [_vp, _sz, C.POINTER(JpegFrame)]),
    "gamut_hip_jpeg_scan_layout": (_i, [_vp, _sz, C.POINTER(JpegFrame), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    "gamut_hip_jpeg_entropy_decode_device": (_i, [C.POINTER(_vp), C.POINTER(_sz), _i, C.POINTER(_i64), C.POINTER(_i64), _vp, _vp, _vp,
                                              C.POINTER(JpegFrame), C.POINTER(_i), _vp]),
    "gamut_hip_jpeg_decode_batch_device": (_i, [C.POINTER(_vp), C.POINTER(_sz), _i, _i, C.POINTER(_i64), _vp, C.POINTER(JpegFrame), C.POINTER(_i), _vp, _vp]),
    "gamut_hip_jpeg_decode_coeffs_batch": (_i, [C.POINTER(_vp), C.POINTER(_sz), _i, C.POINTER(JpegFrame), C.POINTER(_i), _i]),
    "gamut_hip_jpeg_frame_free": (None, [C.POINTER(JpegFrame)]),
    "gamut_hip_decompress_jpeg_image_from_memory": (_vp, [_vp, _sz, _pi, _pi, _pi, _pf, _pf, _i]),
    "gamut_hip_png_defilter_device": (_i, [C.POINTER(PngDesc), _i, _vp, _vp]),
    "gamut_hip_png_defilter_batch_device": (_i, [_vp, _i64, _u32, _vp, _i64, _u32, _u32, _i, _i, _i, _i, _i, _vp, _vp]),
    "gamut_hip_stbi_load_from_memory": (_vp, [_vp, _sz, _pi, _pi, _pi, _i, _pf, _pf, _pf]),
    "gamut_hip_stbi_load_16_from_memory": (_vp, [_vp, _sz, _pi, _pi, _pi, _i, _pf, _pf, _pf]),
    "gamut_hip_png_is16": (_i, [_vp, _sz]),
    "gamut_hip_decompress_jpeg_image_from_stream": (_vp, [_vp, _vp, _pi, _pi, _pi, _pf, _pf, _i]),
    "gamut_hip_stbi_load_from_callbacks": (_vp, [_vp, _vp, _pi, _pi, _pi, _i, _pf, _pf, _pf]),
    "gamut_hip_stbi_load_16_from_callbacks": (_vp, [_vp, _vp, _pi, _pi, _pi, _i, _pf, _pf, _pf]),
    "gamut_hip_stbi_png_is16_from_callbacks": (_i, [_vp, _vp]),
    "gamut_hip_inflate_batch_device": (_i, [_vp, _i, _vp, _vp, _vp]),
    "gamut_hip_inflate_batch_device_sliced": (_i, [_vp, _i, _vp, _vp, C.c_uint32, _vp]),
    "gamut_hip_identify_format": (_i, [_vp, _sz]),
    "gamut_hip_decode_batch_device": (_i, [C.POINTER(_vp), C.POINTER(_sz), _i, _i, C.POINTER(_i64), _vp, C.POINTER(ImageInfo), C.POINTER(_i), _vp]),
    "gamut_hip_shard_owner": (_i, [_i64, _i]),
    "gamut_hip_shard_count": (_i64, [_i, _i, _i64]),
    "gamut_hip_shard_local_index": (_i64, [_i64, _i]),
    "gamut_hip_shard_global_index": (_i64, [_i64, _i, _i]),
    "gamut_hip_host_threads": (_i, []),
    "gamut_hip_comm_get_unique_id": (_i, [_vp]),
    "gamut_hip_comm_init": (_i, [C.POINTER(_vp), _i, _i, _vp]),
    "gamut_hip_comm_destroy": (None, [_vp]),
    "gamut_hip_comm_rank": (_i, [_vp]),
    "gamut_hip_comm_world": (_i, [_vp]),
    "gamut_hip_gather_outputs_device": (_i, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _i, _vp]),
}

JPEG_STREAM_READ_FUNC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_ubyte), C.c_void_p)      # jpegload.d:70


class StbiIoCallbacks(C.Structure):                                                                       # stbdec.d:408-419
    _fields_ = [("read", C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_char), C.c_int)),
                ("skip", C.CFUNCTYPE(None, C.c_void_p, C.c_int)),
                ("eof", C.CFUNCTYPE(C.c_int, C.c_void_p))]

_lib = None


def lib():
    """Load libgamut_hip.so (raises if it has not been built: no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `make -C gamut_amd/csrc` "
                              f"(or __graft_entry__.build()); gamut_amd has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the library does not export it
            fn.restype, fn.argtypes = res, args
        # wake the HIP runtime the library is linked against NOW: a process that loads it, then lets PyTorch (which brings
        # its own copy of the runtime) find the GPU first, was seen to get "no device" from this one afterwards
        L.gamut_hip_device_count()
        _lib = L
    return _lib


def last_error():
    return lib().gamut_hip_last_error().decode("utf-8", "replace")


def check(status):
    if status != OK:
        raise GamutHipError(status, last_error())
