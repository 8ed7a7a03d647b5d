"""Python access to the C++ mirror of Gamut's `Image` (include/gamut_image.h) -- used by tests and examples.
Method names follow image.d."""
import ctypes as C

import numpy as np

from . import _capi

_vp, _i, _sz, _f = C.c_void_p, C.c_int, C.c_size_t, C.c_float
IMAGE_SIGNATURES = {
    "gamut_convert_pixel_type": (_i, [_i, _i]), "gamut_apply_load_flags": (_i, [_i, _i]),
    "gamut_compute_requested_image_components": (_i, [_i]), "gamut_valid_load_flags": (_i, [_i]),
    "gamut_layout_constraints_valid": (_i, [_i]), "gamut_layout_constraints_compatible": (_i, [_i, _i]),
    "gamut_identify_format_from_memory": (_i, [_vp, _sz]), "gamut_free_image_data": (None, [_vp]),
    "gamut_image_new": (_vp, []), "gamut_image_delete": (None, [_vp]),
    "gamut_image_create": (_i, [_vp, _i, _i, _i, _i]), "gamut_image_create_layered": (_i, [_vp, _i, _i, _i, _i, _i]),
    "gamut_image_create_no_init": (_i, [_vp, _i, _i, _i, _i]), "gamut_image_create_layered_no_init": (_i, [_vp, _i, _i, _i, _i, _i]),
    "gamut_image_create_with_no_data": (_i, [_vp, _i, _i, _i, _i]), "gamut_image_create_view": (_i, [_vp, _vp, _i, _i, _i, _i]),
    "gamut_image_load_from_memory": (_i, [_vp, _vp, _sz, _i]),
    "gamut_image_convert_to": (_i, [_vp, _i, _i]), "gamut_image_set_layout": (_i, [_vp, _i]), "gamut_image_convert_op": (_i, [_vp, _i, _i]),
    "gamut_image_convert_to_greyscale_alpha": (_i, [_vp, _i]), "gamut_image_convert_to_rgba": (_i, [_vp, _i]),
    "gamut_image_flip_vertical": (_i, [_vp]), "gamut_image_flip_horizontal": (_i, [_vp]),
    "gamut_image_layer_range": (_vp, [_vp, _i, _i]), "gamut_image_create_layered_view": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i]),
    "gamut_image_clone": (_vp, [_vp]), "gamut_image_copy_pixels_to": (_i, [_vp, _vp]),
    "gamut_image_type": (_i, [_vp]), "gamut_image_width": (_i, [_vp]), "gamut_image_height": (_i, [_vp]), "gamut_image_layers": (_i, [_vp]),
    "gamut_image_pitch_in_bytes": (_i, [_vp]), "gamut_image_layer_offset_in_bytes": (_i, [_vp]), "gamut_image_scanline_in_bytes": (_i, [_vp]),
    "gamut_image_layout_constraints": (_i, [_vp]), "gamut_image_is_error": (_i, [_vp]), "gamut_image_is_valid": (_i, [_vp]),
    "gamut_image_error_message": (C.c_char_p, [_vp]), "gamut_image_has_data": (_i, [_vp]), "gamut_image_is_owned": (_i, [_vp]),
    "gamut_image_is_stored_upside_down": (_i, [_vp]), "gamut_image_pixel_aspect_ratio": (_f, [_vp]), "gamut_image_dots_per_inch_y": (_f, [_vp]),
    "gamut_image_scanptr": (_vp, [_vp, _i]), "gamut_image_layerptr": (_vp, [_vp, _i, _i]), "gamut_image_disown_data": (_vp, [_vp]),
    "gamut_image_set_device_storage": (_i, [_vp, _i]), "gamut_image_is_device": (_i, [_vp]),
    "gamut_image_copy_pixels_to_host": (_i, [_vp, _i, _vp, C.c_int64]),
}
_bound = False


def lib():
    global _bound
    L = _capi.lib()
    if not _bound:
        for name, (res, args) in IMAGE_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _bound = True
    return L


TO_GREYSCALE, TO_RGB, TO_ADD_ALPHA, TO_DROP_ALPHA, TO_PREMUL, TO_NO_PREMUL, TO_8BIT, TO_16BIT, TO_FP32 = range(9)
LOAD_GREYSCALE, LOAD_ALPHA, LOAD_NO_ALPHA, LOAD_RGB = 0x10000, 0x20000, 0x40000, 0x80000
LOAD_8BIT, LOAD_16BIT, LOAD_FP32, LOAD_PREMUL, LOAD_NO_PREMUL = 0x100000, 0x200000, 0x400000, 0x1000000, 0x2000000
LAYOUT_DEFAULT, LAYOUT_VERT_FLIPPED, LAYOUT_VERT_STRAIGHT, LAYOUT_GAPLESS = 0, 512, 1024, 2048
LAYOUT_MULTIPLICITY = {1: 0, 2: 1, 4: 2, 8: 3}
LAYOUT_TRAILING = {0: 0, 1: 4, 3: 8, 7: 12}
LAYOUT_ALIGNED = {1: 0, 2: 16, 4: 32, 8: 48, 16: 64, 32: 80, 64: 96, 128: 112}
LAYOUT_BORDER = {0: 0, 1: 128, 2: 256, 3: 384}


class Image:
    """struct Image (image.d:85).  A fresh Image is in the error state ("Uninitialized image")."""

    def __init__(self, device=False, _handle=None, _keep=None):
        """device=True: pixel storage in HBM (an extension, see include/gamut_image.h): loads and convertTo chains stay on the GPU"""
        self.L = lib()
        self._keep = _keep                                     # a view keeps the image whose pixels it borrows alive
        if _handle is not None:
            self.h = _handle
            return
        self.h = self.L.gamut_image_new()
        if device and not self.L.gamut_image_set_device_storage(self.h, 1):
            raise RuntimeError("device storage needs a GPU")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.gamut_image_delete(self.h)
            self.h = None

    # creation / load
    def create(self, w, h, type=12, layout=0): return bool(self.L.gamut_image_create(self.h, w, h, type, layout))
    def createLayered(self, w, h, layers, type=12, layout=0): return bool(self.L.gamut_image_create_layered(self.h, w, h, layers, type, layout))
    def createNoInit(self, w, h, type=12, layout=0): return bool(self.L.gamut_image_create_no_init(self.h, w, h, type, layout))
    def createWithNoData(self, w, h, type=12, layout=0): return bool(self.L.gamut_image_create_with_no_data(self.h, w, h, type, layout))

    def createView(self, array, w, h, type, pitch):
        self._view_keepalive = array
        return bool(self.L.gamut_image_create_view(self.h, array.ctypes.data if pitch >= 0 else array.ctypes.data + (h - 1) * -pitch, w, h, type, pitch))

    def loadFromMemory(self, data, flags=0):
        buf = np.frombuffer(bytes(data), np.uint8)
        return bool(self.L.gamut_image_load_from_memory(self.h, buf.ctypes.data if buf.size else None, buf.size, flags))

    # conversion
    def convertTo(self, type, layout=0): return bool(self.L.gamut_image_convert_to(self.h, type, layout))
    def setLayout(self, layout): return bool(self.L.gamut_image_set_layout(self.h, layout))
    def convertOp(self, op, layout=0): return bool(self.L.gamut_image_convert_op(self.h, op, layout))
    def convertTo8Bit(self, layout=0): return self.convertOp(TO_8BIT, layout)
    def convertTo16Bit(self, layout=0): return self.convertOp(TO_16BIT, layout)
    def convertToFP32(self, layout=0): return self.convertOp(TO_FP32, layout)
    def convertToGreyscale(self, layout=0): return self.convertOp(TO_GREYSCALE, layout)
    def convertToRGB(self, layout=0): return self.convertOp(TO_RGB, layout)
    def convertToRGBA(self, layout=0): return bool(self.L.gamut_image_convert_to_rgba(self.h, layout))
    def convertToGreyscaleAlpha(self, layout=0): return bool(self.L.gamut_image_convert_to_greyscale_alpha(self.h, layout))
    def addAlphaChannel(self, layout=0): return self.convertOp(TO_ADD_ALPHA, layout)
    def dropAlphaChannel(self, layout=0): return self.convertOp(TO_DROP_ALPHA, layout)
    def premultiply(self, layout=0): return self.convertOp(TO_PREMUL, layout)
    def unpremultiply(self, layout=0): return self.convertOp(TO_NO_PREMUL, layout)
    def flipVertical(self): return bool(self.L.gamut_image_flip_vertical(self.h))
    def flipHorizontal(self): return bool(self.L.gamut_image_flip_horizontal(self.h))

    # views and copies
    def layerRange(self, start, end): return Image(_handle=self.L.gamut_image_layer_range(self.h, start, end), _keep=self)
    def layer(self, index): return self.layerRange(index, index + 1)
    def clone(self): return Image(_handle=self.L.gamut_image_clone(self.h))
    def copyPixelsTo(self, other): return bool(self.L.gamut_image_copy_pixels_to(self.h, other.h))

    def createLayeredView(self, array, w, h, layers, type, pitch, layer_offset):
        self._view_keepalive = array
        return bool(self.L.gamut_image_create_layered_view(self.h, array.ctypes.data if pitch >= 0 else array.ctypes.data + (h - 1) * -pitch, w, h, layers, type, pitch, layer_offset))

    # state
    @property
    def type(self): return self.L.gamut_image_type(self.h)
    @property
    def width(self): return self.L.gamut_image_width(self.h)
    @property
    def height(self): return self.L.gamut_image_height(self.h)
    @property
    def layers(self): return self.L.gamut_image_layers(self.h)
    @property
    def pitchInBytes(self): return self.L.gamut_image_pitch_in_bytes(self.h)
    @property
    def layerOffsetInBytes(self): return self.L.gamut_image_layer_offset_in_bytes(self.h)
    @property
    def scanlineInBytes(self): return self.L.gamut_image_scanline_in_bytes(self.h)
    @property
    def layoutConstraints(self): return self.L.gamut_image_layout_constraints(self.h)
    @property
    def isError(self): return bool(self.L.gamut_image_is_error(self.h))
    @property
    def isValid(self): return bool(self.L.gamut_image_is_valid(self.h))
    @property
    def errorMessage(self):
        m = self.L.gamut_image_error_message(self.h)
        return None if m is None else m.decode()
    @property
    def hasData(self): return bool(self.L.gamut_image_has_data(self.h))
    @property
    def isOwned(self): return bool(self.L.gamut_image_is_owned(self.h))
    @property
    def isStoredUpsideDown(self): return bool(self.L.gamut_image_is_stored_upside_down(self.h))
    @property
    def pixelAspectRatio(self): return self.L.gamut_image_pixel_aspect_ratio(self.h)
    @property
    def dotsPerInchY(self): return self.L.gamut_image_dots_per_inch_y(self.h)

    def scanptr(self, y): return self.L.gamut_image_scanptr(self.h, y)
    def layerptr(self, layer, y): return self.L.gamut_image_layerptr(self.h, layer, y)

    @property
    def isDevice(self): return bool(self.L.gamut_image_is_device(self.h))

    def scanline(self, y, layer=0):
        n = self.scanlineInBytes
        if self.isDevice:
            return self.pixels(layer)[y].copy()
        return np.ctypeslib.as_array(C.cast(self.layerptr(layer, y), C.POINTER(C.c_uint8)), (n,)).copy() if n else np.zeros(0, np.uint8)

    def pixels(self, layer=0):
        """(height, scanlineInBytes) uint8 copy of one layer in logical (top-down) order"""
        if self.isDevice:
            out = np.zeros((self.height, self.scanlineInBytes), np.uint8)
            if out.size and not self.L.gamut_image_copy_pixels_to_host(self.h, layer, out.ctypes.data, out.shape[1]):
                raise RuntimeError("copy_pixels_to_host failed")
            return out
        return np.stack([self.scanline(y, layer) for y in range(self.height)]) if self.height else np.zeros((0, self.scanlineInBytes), np.uint8)
