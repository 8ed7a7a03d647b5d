"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the reference.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module; the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

PIXEL_TYPES = ["l8", "l16", "lf32", "la8", "la16", "laf32", "lap8", "lap16", "lapf32",
               "rgb8", "rgb16", "rgbf32", "rgba8", "rgba16", "rgbaf32",
               "rgbap8", "rgbap16", "rgbapf32"]
PT = {n: i for i, n in enumerate(PIXEL_TYPES)}
PT_SIZE = [1, 2, 4, 2, 4, 8, 2, 4, 8, 3, 6, 12, 4, 8, 16, 4, 8, 16]
PT_CHANNELS = [1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 4, 4, 4]
PT_DTYPE = [np.uint8, np.uint16, np.float32] * 6

JPGD_GRAYSCALE, JPGD_YH1V1, JPGD_YH2V1, JPGD_YH1V2, JPGD_YH2V2 = range(5)


def build():
    """(Re)build liboracle.so with the committed Makefile."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


class JpegFrame(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("comps", C.c_int), ("scan_type", C.c_int),
                ("mcus_per_row", C.c_int), ("mcus_per_col", C.c_int), ("blocks_per_mcu", C.c_int),
                ("coeffs", C.POINTER(C.c_int16)), ("max_zag", C.POINTER(C.c_uint8)),
                ("pixel_aspect_ratio", C.c_float), ("dpi_y", C.c_float)]


class PngInfo(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32),
                ("depth", C.c_int), ("color", C.c_int), ("interlace", C.c_int),
                ("img_n", C.c_int), ("pal_img_n", C.c_int), ("has_trans", C.c_int), ("is_iphone", C.c_int),
                ("palette", C.c_uint8 * 1024), ("pal_len", C.c_uint32),
                ("tc", C.c_uint8 * 3), ("tc16", C.c_uint16 * 3),
                ("raw", C.POINTER(C.c_uint8)), ("raw_len", C.c_uint32),
                ("ppmX", C.c_float), ("ppmY", C.c_float), ("pixelAspectRatio", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    vp, i32, u32, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_size_t
    L.orc_pixel_type_size.argtypes = [i32]
    L.orc_scanlines_inter_type.argtypes = [i32, i32]
    L.orc_scanlines_copy.argtypes = [i32, vp, i32, vp, i32, i32, i32]
    L.orc_scanlines_convert.argtypes = [i32, vp, i32, i32, vp, i32, i32, i32, i32, vp]
    L.orc_jpeg_idct.argtypes = [vp, vp, i32]
    L.orc_jpeg_idct.restype = None
    L.orc_jpeg_idct_4x4.argtypes = [vp, vp]
    L.orc_jpeg_idct_4x4.restype = None
    L.orc_jpeg_idct_colfirst.argtypes = [vp, vp]
    L.orc_jpeg_idct_colfirst.restype = None
    L.orc_jpeg_upsample_block.argtypes = [vp, i32, vp]
    L.orc_jpeg_upsample_block.restype = None
    L.orc_jpeg_decode_coeffs.argtypes = [vp, sz, C.POINTER(JpegFrame)]
    L.orc_jpeg_frame_free.argtypes = [C.POINTER(JpegFrame)]
    L.orc_jpeg_frame_free.restype = None
    L.orc_jpeg_reconstruct.argtypes = [C.POINTER(JpegFrame), i32, vp, i32, i32]
    L.orc_decompress_jpeg_image_from_memory.argtypes = [vp, sz, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32),
                                                        C.POINTER(C.c_float), C.POINTER(C.c_float), i32]
    L.orc_decompress_jpeg_image_from_memory.restype = vp
    L.orc_png_create_image_raw.argtypes = [vp, u32, i32, i32, u32, u32, i32, i32, vp]
    L.orc_png_create_image.argtypes = [vp, u32, i32, i32, u32, u32, i32, i32, i32, vp]
    L.orc_png_parse.argtypes = [vp, sz, C.POINTER(PngInfo)]
    L.orc_png_info_free.argtypes = [C.POINTER(PngInfo)]
    L.orc_png_info_free.restype = None
    L.orc_stbi_load_from_memory.argtypes = [vp, sz, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i32]
    L.orc_stbi_load_from_memory.restype = vp
    L.orc_stbi_load_16_from_memory.argtypes = [vp, sz, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i32]
    L.orc_stbi_load_16_from_memory.restype = vp
    L.orc_png_convert_format8.argtypes = [vp, i32, i32, u32, u32, vp]
    L.orc_png_convert_format8.restype = None
    L.orc_png_convert_format16.argtypes = [vp, i32, i32, u32, u32, vp]
    L.orc_png_convert_format16.restype = None
    _lib = L
    return L


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]
_libc.free.restype = None


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------- convert
def scanlines_convert(src_type, src, dst_type, width, height, src_pitch=None, dst_pitch=None):
    """src: contiguous uint8 array holding height rows of src_pitch bytes.
    Returns uint8 array of height*dst_pitch bytes."""
    L = lib()
    st, dt = PT[src_type] if isinstance(src_type, str) else src_type, PT[dst_type] if isinstance(dst_type, str) else dst_type
    if src_pitch is None:
        src_pitch = width * PT_SIZE[st]
    if dst_pitch is None:
        dst_pitch = width * PT_SIZE[dt]
    src = np.ascontiguousarray(src).view(np.uint8).reshape(-1)
    dst = np.zeros(max(1, abs(dst_pitch) * height), np.uint8)
    inter = L.orc_scanlines_inter_type(st, dt)
    ibuf = np.zeros(max(16, width * PT_SIZE[inter]), np.uint8)
    s_off = (height - 1) * -src_pitch if src_pitch < 0 else 0
    d_off = (height - 1) * -dst_pitch if dst_pitch < 0 else 0
    ok = L.orc_scanlines_convert(st, src.ctypes.data + s_off, src_pitch, dt, dst.ctypes.data + d_off, dst_pitch,
                                 width, height, inter, _ptr(ibuf))
    assert ok
    return dst


# ---------------------------------------------------------------- jpeg
def jpeg_idct(block, max_zag=64):
    b = np.ascontiguousarray(block, np.int16).reshape(64)
    out = np.zeros(64, np.uint8)
    lib().orc_jpeg_idct(_ptr(b), _ptr(out), int(max_zag))
    return out.reshape(8, 8)


def jpeg_idct_4x4(block):
    b = np.ascontiguousarray(block, np.int16).reshape(64)
    out = np.zeros(64, np.uint8)
    lib().orc_jpeg_idct_4x4(_ptr(b), _ptr(out))
    return out.reshape(8, 8)


def jpeg_upsample_block(block, max_zag=64):
    b = np.ascontiguousarray(block, np.int16).reshape(64)
    out = np.zeros(4 * 64, np.int16)
    lib().orc_jpeg_upsample_block(_ptr(b), int(max_zag), _ptr(out))
    return out.reshape(4, 8, 8)


class DecodedJpeg:
    """Dense coefficient form of a baseline JPEG (host feeder output)."""

    def __init__(self, data):
        self._f = JpegFrame()
        buf = np.frombuffer(bytes(data), np.uint8)
        rc = lib().orc_jpeg_decode_coeffs(_ptr(buf), buf.size, C.byref(self._f))
        if rc != 0:
            raise ValueError("oracle: jpeg decode failed")
        f = self._f
        self.width, self.height, self.comps, self.scan_type = f.width, f.height, f.comps, f.scan_type
        self.mcus_per_row, self.mcus_per_col, self.blocks_per_mcu = f.mcus_per_row, f.mcus_per_col, f.blocks_per_mcu
        n = f.mcus_per_row * f.mcus_per_col * f.blocks_per_mcu
        self.coeffs = np.ctypeslib.as_array(f.coeffs, (n, 64)).copy()
        self.max_zag = np.ctypeslib.as_array(f.max_zag, (n,)).copy()
        self.pixel_aspect_ratio, self.dpi_y = f.pixel_aspect_ratio, f.dpi_y
        lib().orc_jpeg_frame_free(C.byref(self._f))


def jpeg_reconstruct(width, height, comps, scan_type, coeffs, max_zag=None, req_comps=4, colfirst=False):
    """coefficients (nblocks,64) int16 in MCU order -> pixels (height, width*req_comps) uint8."""
    f = JpegFrame()
    f.width, f.height, f.comps, f.scan_type = width, height, comps, scan_type
    mx = 16 if scan_type in (JPGD_YH2V1, JPGD_YH2V2) else 8
    my = 16 if scan_type in (JPGD_YH1V2, JPGD_YH2V2) else 8
    f.mcus_per_row, f.mcus_per_col = (width + mx - 1) // mx, (height + my - 1) // my
    f.blocks_per_mcu = {JPGD_GRAYSCALE: 1, JPGD_YH1V1: 3, JPGD_YH2V1: 4, JPGD_YH1V2: 4, JPGD_YH2V2: 6}[scan_type]
    co = np.ascontiguousarray(coeffs, np.int16).reshape(-1)
    assert co.size == f.mcus_per_row * f.mcus_per_col * f.blocks_per_mcu * 64
    f.coeffs = co.ctypes.data_as(C.POINTER(C.c_int16))
    if max_zag is not None:
        mz = np.ascontiguousarray(max_zag, np.uint8).reshape(-1)
        f.max_zag = mz.ctypes.data_as(C.POINTER(C.c_uint8))
    out = np.zeros((height, width * req_comps), np.uint8)
    rc = lib().orc_jpeg_reconstruct(C.byref(f), req_comps, _ptr(out), width * req_comps, int(colfirst))
    assert rc == 0
    return out


def decompress_jpeg(data, req_comps=-1):
    buf = np.frombuffer(bytes(data), np.uint8)
    w, h, ac = C.c_int(), C.c_int(), C.c_int()
    par, dpi = C.c_float(), C.c_float()
    p = lib().orc_decompress_jpeg_image_from_memory(_ptr(buf) if buf.size else None, buf.size, C.byref(w), C.byref(h),
                                                    C.byref(ac), C.byref(par), C.byref(dpi), req_comps)
    if not p:
        return None
    comps = ac.value if req_comps < 0 else req_comps
    n = w.value * h.value * comps
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n,)).copy().reshape(h.value, w.value * comps)
    _libc.free(p)
    return out, ac.value, par.value, dpi.value


def same_density(got, want):
    """pixelAspectRatio / dotsPerInchY pairs, bit for bit as floats except that NaN equals NaN (a file without JFIF / EXIF density: the D struct's
    float members are never assigned, jpegload.d:510-512)"""
    return all((np.isnan(np.float32(g)) and np.isnan(np.float32(w))) or np.float32(g) == np.float32(w) for g, w in zip(got, want))


# ---------------------------------------------------------------- qoi
class QoiDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint8), ("colorspace", C.c_uint8)]


def qoi_decode(data, channels=0):
    """-> (pixels (h, w*channels) uint8, file_channels, colorspace) or None"""
    buf = np.frombuffer(bytes(data), np.uint8) if len(data) else np.zeros(1, np.uint8)
    d = QoiDesc()
    fn = lib().orc_qoi_decode
    fn.restype, fn.argtypes = C.c_void_p, [C.c_void_p, C.c_int, C.POINTER(QoiDesc), C.c_int]
    p = fn(buf.ctypes.data, len(data), C.byref(d), channels)
    if not p:
        return None
    ch = channels or d.channels
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (d.height * d.width * ch,)).copy().reshape(d.height, d.width * ch)
    _libc.free(p)
    return out, d.channels, d.colorspace


# ---------------------------------------------------------------- png
def png_create_image_raw(raw, img_n, out_n, x, y, depth, color=None):
    raw = np.ascontiguousarray(raw, np.uint8)
    if color is None:
        color = {1: 0, 2: 4, 3: 2, 4: 6}[img_n]
    nbytes = 2 if depth == 16 else 1
    out = np.zeros(x * y * out_n * nbytes + 16, np.uint8)
    ok = lib().orc_png_create_image_raw(_ptr(raw), raw.size, img_n, out_n, x, y, depth, color, _ptr(out))
    if not ok:
        return None
    return out[:x * y * out_n * nbytes]


def png_parse(data):
    info = PngInfo()
    buf = np.frombuffer(bytes(data), np.uint8)
    ok = lib().orc_png_parse(_ptr(buf), buf.size, C.byref(info))
    if not ok:
        return None
    d = {k: getattr(info, k) for k in ("width", "height", "depth", "color", "interlace", "img_n", "pal_img_n",
                                        "has_trans", "is_iphone", "pal_len", "ppmX", "ppmY", "pixelAspectRatio")}
    d["raw"] = np.ctypeslib.as_array(info.raw, (info.raw_len,)).copy()
    d["palette"] = np.frombuffer(bytes(info.palette), np.uint8).copy()
    d["tc"] = list(info.tc)
    d["tc16"] = list(info.tc16)
    lib().orc_png_info_free(C.byref(info))
    return d


def stbi_load(data, req_comp=0, sixteen=False):
    buf = np.frombuffer(bytes(data), np.uint8)
    x, y, n = C.c_int(), C.c_int(), C.c_int()
    fn = lib().orc_stbi_load_16_from_memory if sixteen else lib().orc_stbi_load_from_memory
    p = fn(_ptr(buf), buf.size, C.byref(x), C.byref(y), C.byref(n), req_comp)
    if not p:
        return None
    comps = n.value if req_comp == 0 else req_comp
    cnt = x.value * y.value * comps
    ct = C.c_uint16 if sixteen else C.c_uint8
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), (cnt,)).copy().reshape(y.value, x.value, comps)
    _libc.free(p)
    return out, n.value
