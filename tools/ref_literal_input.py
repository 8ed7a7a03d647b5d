#!/usr/bin/env python3
"""ref_literal_input.py -- the SECOND READING of jpgd's input layer: the byte / bit reader, the marker walk, the Huffman
decoder, restart handling, baseline and progressive coefficient decoding and the driver of
/root/reference/source/gamut/codecs/jpegload.d, written statement for statement from the D text with the decoder's state
KEPT (the 8 KB input buffer with its pad areas, m_pIn_buf_ofs / m_in_buf_left, stuffed-back bytes, m_tem_flag, m_bit_buf /
m_bits_left, the look_up / look_up2 / tree tables of make_huff_table), not modelled.

TEST INFRASTRUCTURE ONLY, build-container only (like tools/ref_literal_jpeg.py): nothing in the product, the oracle or the
-m gpu tests imports it.  tools/fuzz_input.py runs it against oracle/liboracle.so (and the product's host feeder) on mutated
files; every disagreement becomes a committed file under tests/golden/jpeg_fuzz/ with the expected result next to it.

What it returns is what decompress_jpeg_image_from_stream (:3720-3808) hands to transform_mcu / transform_mcu_expand and to
its caller: per MCU the m_pMCU_coefficients blocks and m_mcu_block_max_zag (the pixel arithmetic behind them is
tools/ref_literal_jpeg.py's subject), width / height / components, pixelAspectRatio / dotsPerInchY, or None (null).

D semantics kept: `int` / `uint` are 32 bits with wrap-around, `short` stores keep the low 16 bits, a `float` member of a
struct starts as NaN (jpeg_decoder.m_pixelAspectRatio / m_pixelsPerInch* are never assigned in initit :1971-2078).

Where the reference would read or write outside an array, use memory it never wrote, trip an assert or never return, this
file raises Undefined(reason): there is no result to restate (the repo's decoders reject such files; the fuzzer checks that).
"""
import math
import struct

import numpy as np

JPGD_IN_BUF_SIZE, JPGD_MAX_BLOCKS_PER_MCU, JPGD_MAX_HUFF_TABLES, JPGD_MAX_QUANT_TABLES = 8192, 10, 8, 4      # :95-98
JPGD_MAX_COMPONENTS, JPGD_MAX_COMPS_IN_SCAN, JPGD_MAX_BLOCKS_PER_ROW, JPGD_MAX_HEIGHT, JPGD_MAX_WIDTH = 4, 4, 8192, 16384, 16384
JPGD_SUCCESS, JPGD_FAILED, JPGD_DONE = 0, -1, 1
g_ZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56,
         57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]      # :101
M_SOF0, M_SOF1, M_SOF2, M_SOF3, M_SOF5, M_SOF6, M_SOF7, M_JPG = 0xC0, 0xC1, 0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC8                 # :104-110
M_SOF9, M_SOF10, M_SOF11, M_SOF13, M_SOF14, M_SOF15, M_DHT, M_DAC = 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF, 0xC4, 0xCC
M_RST0, M_RST7, M_SOI, M_EOI, M_SOS, M_DQT, M_DRI, M_APP0, M_TEM = 0xD0, 0xD7, 0xD8, 0xD9, 0xDA, 0xDB, 0xDD, 0xE0, 0x01
JPGD_GRAYSCALE, JPGD_YH1V1, JPGD_YH2V1, JPGD_YH1V2, JPGD_YH2V2 = 0, 1, 2, 3, 4                                                    # :113

U32 = 0xFFFFFFFF
NAN = float("nan")
import collections
EVENTS = collections.Counter()             # what a run exercised (tools/fuzz_input.py sums them up): not part of the reading


class Undefined(Exception):
    """the reference has no defined result on this input (out-of-bounds access, uninitialised memory, assert, endless loop)"""


class Rejected(Exception):
    """set_error (:1097-1101) was called: m_error_code is set and every later check of it (:3733, :532, :546) returns null"""


def f32(x):
    return struct.unpack("<f", struct.pack("<f", x))[0] if not (math.isnan(x) or math.isinf(x)) else x


def to_float(x):                           # a double (or int) assigned to a float
    try:
        return np.float32(x).item()
    except OverflowError:
        return math.copysign(math.inf, x)


def i32(x):
    x &= U32
    return x - (1 << 32) if x & 0x80000000 else x


def i16(x):
    x &= 0xFFFF
    return x - 0x10000 if x & 0x8000 else x


def convertInchesToMeters(x):              # types.d:127-130, float x / 39.37007874f
    return (np.float32(x) / np.float32(39.37007874)).item()


s_extend_test = [0, 0x0001, 0x0002, 0x0004, 0x0008, 0x0010, 0x0020, 0x0040, 0x0080, 0x0100, 0x0200, 0x0400, 0x0800, 0x1000, 0x2000, 0x4000]   # :816
s_extend_offset = [0] + [((-1) << n) + 1 for n in range(1, 16)]                                                                                # :817


def JPGD_HUFF_EXTEND(x, s):                # :819-822
    if not 0 <= s <= 15:
        raise Undefined("JPGD_HUFF_EXTEND: s_extend_test[%d]" % s)
    return x + s_extend_offset[s] if x < s_extend_test[s] else x


class huff_tables:                         # :413-419
    def __init__(self):
        self.ac_table = False
        self.look_up = [0] * 256
        self.look_up2 = [0] * 256
        self.code_size = [0] * 256
        self.tree = [0] * 512


class coeff_buf:                           # :421-426; pData as int16 words
    pass


class MemoryStream:
    """the plug-in's stream_read_jpeg (plugins/jpeg.d:158-169) over gamut's memory IO (io.d:428-471): up to max bytes, eof = offset >= bytes"""
    def __init__(self, data):
        self.data, self.offset = bytes(data), 0

    def __call__(self, max_bytes_to_read):
        n = min(max_bytes_to_read, len(self.data) - self.offset)
        chunk = self.data[self.offset:self.offset + n]
        self.offset += n
        return chunk, self.offset >= len(self.data)


class jpeg_decoder:                        # :401-3714
    PAD = 128                              # m_in_buf_pad_start[128] | m_in_buf[8192 + 128] | m_in_buf_pad_end[128]: one array, m_in_buf at PAD (:481-483)

    def __init__(self, rfn):               # this(...) :517-522
        self.m_pixelsPerInchX = self.m_pixelsPerInchY = self.m_pixelAspectRatio = NAN       # float members: .init (never assigned in initit)
        self.mcus = []                     # what transform_mcu / transform_mcu_expand are handed, in call order: (coefficients[blocks*64], max_zag[blocks])
        self.success = self.decode_init(rfn)

    # ---- :631-696 ------------------------------------------------------------------------------------------------
    def get_char(self):                    # :631-652 (and :655-674; no stream error exists on a memory stream)
        if not self.m_in_buf_left:
            self.prep_in_buffer()
            if not self.m_in_buf_left:
                EVENTS["pad byte read"] += 1
                t = self.m_tem_flag
                self.m_tem_flag ^= 1
                return 0xD9 if t else 0xFF
        c = self.mem[self.m_pIn_buf_ofs]
        self.m_pIn_buf_ofs += 1
        self.m_in_buf_left -= 1
        return c

    def stuff_char(self, q):               # :677-680
        self.m_pIn_buf_ofs -= 1
        if self.m_pIn_buf_ofs < 0:
            raise Undefined("stuff_char in front of m_in_buf_pad_start")
        self.mem[self.m_pIn_buf_ofs] = q
        self.m_in_buf_left += 1

    def get_octet(self):                   # :683-696.  get_char(&padding_flag) binds to get_char(bool* err): padding_flag is the (never set) error flag
        c = self.get_char()
        if c == 0xFF:
            c = self.get_char()
            if c == 0x00:
                return 0xFF
            self.stuff_char(c)
            self.stuff_char(0xFF)
            return 0xFF
        return c

    def get_bits(self, num_bits):          # :699-719
        if not num_bits:
            return 0
        i = self.m_bit_buf >> (32 - num_bits)
        self.m_bits_left -= num_bits
        if self.m_bits_left <= 0:
            num_bits += self.m_bits_left
            self.m_bit_buf = (self.m_bit_buf << num_bits) & U32
            c1 = self.get_char()
            c2 = self.get_char()
            self.m_bit_buf = (self.m_bit_buf & 0xFFFF0000) | (c1 << 8) | c2
            self.m_bit_buf = (self.m_bit_buf << -self.m_bits_left) & U32
            self.m_bits_left += 16
            if self.m_bits_left < 0:
                raise Undefined("get_bits: assert(m_bits_left >= 0)")
        else:
            self.m_bit_buf = (self.m_bit_buf << num_bits) & U32
        return i

    def get_bits_no_markers(self, num_bits):        # :722-743
        if not num_bits:
            return 0
        if num_bits > 32:
            raise Undefined("get_bits_no_markers(%d)" % num_bits)
        i = self.m_bit_buf >> (32 - num_bits)
        self.m_bits_left -= num_bits
        if self.m_bits_left <= 0:
            num_bits += self.m_bits_left
            if num_bits < 0:
                raise Undefined("get_bits_no_markers: more bits than the buffer holds")
            self.m_bit_buf = (self.m_bit_buf << num_bits) & U32
            mem, o = self.mem, self.m_pIn_buf_ofs
            if self.m_in_buf_left < 2 or mem[o] == 0xFF or mem[o + 1] == 0xFF:
                c1 = self.get_octet()
                c2 = self.get_octet()
                self.m_bit_buf |= (c1 << 8) | c2
            else:
                self.m_bit_buf |= (mem[o] << 8) | mem[o + 1]
                self.m_in_buf_left -= 2
                self.m_pIn_buf_ofs += 2
            self.m_bit_buf = (self.m_bit_buf << -self.m_bits_left) & U32
            self.m_bits_left += 16
            if self.m_bits_left < 0:
                raise Undefined("get_bits_no_markers: assert(m_bits_left >= 0)")
        else:
            self.m_bit_buf = (self.m_bit_buf << num_bits) & U32
        return i

    # ---- :746-813 ------------------------------------------------------------------------------------------------
    def _tree_walk(self, pH, symbol):      # the do / while of :752-756 and :775-779
        ofs = 23
        while True:
            idx = -i32(symbol + ((self.m_bit_buf >> ofs) & 1))
            if not 0 <= idx < 512 or ofs < 0:
                raise Undefined("huff_decode: tree[%d] / bit %d" % (idx, ofs))
            symbol = i32(pH.tree[idx])
            ofs -= 1
            if symbol >= 0:
                return symbol, ofs

    def huff_decode(self, pH):             # :746-766
        symbol = i32(pH.look_up[self.m_bit_buf >> 24])
        if symbol < 0:
            symbol, ofs = self._tree_walk(pH, symbol)
            if symbol == 0 and pH.code_size[0] != 8 + (23 - ofs):
                EVENTS["two-argument huff_decode: empty tree slot"] += 1
            self.get_bits_no_markers(8 + (23 - ofs))
        else:
            if symbol == 0 and pH.look_up2[self.m_bit_buf >> 24] == 0:
                EVENTS["two-argument huff_decode: empty look_up entry"] += 1
            elif pH.look_up2[self.m_bit_buf >> 24] >> 8 & 31 != pH.code_size[symbol] + (symbol & 15 if pH.look_up2[self.m_bit_buf >> 24] & 0x8000 else 0):
                EVENTS["two-argument huff_decode: code_size[symbol] is another code word's length"] += 1
            self.get_bits_no_markers(pH.code_size[symbol])
        return symbol

    def huff_decode2(self, pH):            # :769-813, -> (symbol, extra_bits); extra_bits is None where the D leaves its `ref` untouched... it never does
        symbol = i32(pH.look_up2[self.m_bit_buf >> 24])
        if symbol == 0:
            EVENTS["three-argument huff_decode: empty look_up2 entry"] += 1
        if symbol < 0:
            symbol, ofs = self._tree_walk(pH, symbol)
            if symbol == 0 and pH.code_size[0] != 8 + (23 - ofs):
                EVENTS["three-argument huff_decode: empty tree slot"] += 1
            self.get_bits_no_markers(8 + (23 - ofs))
            extra_bits = self.get_bits_no_markers(symbol & 0xF)
        else:
            if symbol & 0x8000:
                self.get_bits_no_markers((symbol >> 8) & 31)
                extra_bits = symbol >> 16
            else:
                code_size = (symbol >> 8) & 31
                num_extra_bits = symbol & 0xF
                bits = code_size + num_extra_bits
                if bits <= self.m_bits_left + 16:
                    extra_bits = self.get_bits_no_markers(bits) & ((1 << num_extra_bits) - 1)
                else:
                    self.get_bits_no_markers(code_size)
                    extra_bits = self.get_bits_no_markers(num_extra_bits)
            symbol &= 0xFF
        return symbol, extra_bits

    # ---- :1097-1174 ----------------------------------------------------------------------------------------------
    def set_error(self, status):           # :1097-1101
        self.m_error_code = status
        raise Rejected(status)

    def prep_in_buffer(self):              # :1149-1174
        self.m_in_buf_left = 0
        self.m_pIn_buf_ofs = self.PAD
        if self.m_eof_flag:
            return True
        while True:
            chunk, self.m_eof_flag = self.readfn(JPGD_IN_BUF_SIZE - self.m_in_buf_left)
            self.mem[self.PAD + self.m_in_buf_left:self.PAD + self.m_in_buf_left + len(chunk)] = chunk
            self.m_in_buf_left += len(chunk)
            if not (self.m_in_buf_left < JPGD_IN_BUF_SIZE and not self.m_eof_flag):
                break
        self.m_total_bytes_read += self.m_in_buf_left
        p = self.m_pIn_buf_ofs + self.m_in_buf_left          # word_clear(..., 0xD9FF, 64) :1136-1144, :1172
        for n in range(64):
            self.mem[p + 2 * n] = 0xFF
            self.mem[p + 2 * n + 1] = 0xD9
        return True

    # ---- :1177-1543 ----------------------------------------------------------------------------------------------
    def read_dht_marker(self):             # :1177-1269
        huff_num = [0] * 17                  # D clears its locals: ubyte[17] huff_num; ubyte[256] huff_val; (:1179-1180)
        huff_val = [0] * 256
        num_left = self.get_bits(16)
        if num_left < 2:
            self.set_error("JPGD_BAD_DHT_MARKER")
        num_left -= 2
        while num_left:
            index = self.get_bits(8)
            huff_num[0] = 0
            count = 0
            for i in range(1, 17):
                huff_num[i] = self.get_bits(8)
                count += huff_num[i]
            if count > 255:
                self.set_error("JPGD_BAD_DHT_COUNTS")
            for i in range(count):
                huff_val[i] = self.get_bits(8)
            i = 1 + 16 + count
            if num_left < i:
                self.set_error("JPGD_BAD_DHT_MARKER")
            num_left -= i
            if (index & 0x10) > 0x10:
                self.set_error("JPGD_BAD_DHT_INDEX")
            index = (index & 0x0F) + ((index & 0x10) >> 4) * (JPGD_MAX_HUFF_TABLES >> 1)
            if index >= JPGD_MAX_HUFF_TABLES:
                self.set_error("JPGD_BAD_DHT_INDEX")
            self.m_huff_ac[index] = (index & 0x10) != 0
            self.m_huff_num[index] = list(huff_num)
            self.m_huff_val[index] = list(huff_val)
        return True

    def read_dqt_marker(self):             # :1272-1343
        num_left = self.get_bits(16)
        if num_left < 2:
            self.set_error("JPGD_BAD_DQT_MARKER")
        num_left -= 2
        while num_left:
            n = self.get_bits(8)
            prec = n >> 4
            n &= 0x0F
            if n >= JPGD_MAX_QUANT_TABLES:
                self.set_error("JPGD_BAD_DQT_TABLE")
            if self.m_quant[n] is None:
                self.m_quant[n] = [None] * 64
            for i in range(64):
                temp = self.get_bits(8)
                if prec:
                    temp = (temp << 8) + self.get_bits(8)
                self.m_quant[n][i] = i16(temp)
            i = 64 + 1
            if prec:
                i += 64
            if num_left < i:
                self.set_error("JPGD_BAD_DQT_LENGTH")
            num_left -= i
        return True

    def read_sof_marker(self):             # :1346-1415
        num_left = self.get_bits(16)
        if self.get_bits(8) != 8:
            self.set_error("JPGD_BAD_PRECISION")
        self.m_image_y_size = self.get_bits(16)
        if self.m_image_y_size < 1 or self.m_image_y_size > JPGD_MAX_HEIGHT:
            self.set_error("JPGD_BAD_HEIGHT")
        self.m_image_x_size = self.get_bits(16)
        if self.m_image_x_size < 1 or self.m_image_x_size > JPGD_MAX_WIDTH:
            self.set_error("JPGD_BAD_WIDTH")
        self.m_comps_in_frame = self.get_bits(8)
        if self.m_comps_in_frame > JPGD_MAX_COMPONENTS:
            self.set_error("JPGD_TOO_MANY_COMPONENTS")
        if num_left != self.m_comps_in_frame * 3 + 8:
            self.set_error("JPGD_BAD_SOF_LENGTH")
        for i in range(self.m_comps_in_frame):
            self.m_comp_ident[i] = self.get_bits(8)
            self.m_comp_h_samp[i] = self.get_bits(4)
            self.m_comp_v_samp[i] = self.get_bits(4)
            self.m_comp_quant[i] = self.get_bits(8)
        return True

    def skip_variable_marker(self):        # :1418-1442
        num_left = self.get_bits(16)
        if num_left < 2:
            self.set_error("JPGD_BAD_VARIABLE_MARKER")
        num_left -= 2
        while num_left:
            self.get_bits(8)
            num_left -= 1
        return True

    def read_dri_marker(self):             # :1445-1462
        drilen = self.get_bits(16)
        if drilen != 4:
            self.set_error("JPGD_BAD_DRI_LENGTH")
        self.m_restart_interval = self.get_bits(16)
        return True

    def read_sos_marker(self):             # :1466-1543
        num_left = self.get_bits(16)
        n = self.get_bits(8)
        self.m_comps_in_scan = n
        num_left = (num_left - 3) & U32
        if num_left != n * 2 + 3 or n < 1 or n > JPGD_MAX_COMPS_IN_SCAN:
            self.set_error("JPGD_BAD_SOS_LENGTH")
        for i in range(n):
            cc = self.get_bits(8)
            c = self.get_bits(8)
            num_left -= 2
            ci = 0
            while ci < self.m_comps_in_frame:
                if cc == self.m_comp_ident[ci]:
                    break
                ci += 1
            if ci >= self.m_comps_in_frame:
                self.set_error("JPGD_BAD_SOS_COMP_ID")
            self.m_comp_list[i] = ci
            self.m_comp_dc_tab[ci] = (c >> 4) & 15
            self.m_comp_ac_tab[ci] = (c & 15) + (JPGD_MAX_HUFF_TABLES >> 1)
        self.m_spectral_start = self.get_bits(8)
        self.m_spectral_end = self.get_bits(8)
        self.m_successive_high = self.get_bits(4)
        self.m_successive_low = self.get_bits(4)
        if not self.m_progressive_flag:
            self.m_spectral_start = 0
            self.m_spectral_end = 63
        num_left -= 3
        while num_left:
            self.get_bits(8)
            num_left -= 1
        return True

    # ---- :1546-1848 ----------------------------------------------------------------------------------------------
    def next_marker(self):                 # :1546-1573
        while True:
            while True:
                c = self.get_bits(8)
                if c == 0xFF:
                    break
            while True:
                c = self.get_bits(8)
                if c != 0xFF:
                    break
            if c != 0:
                return c

    def process_markers(self, allow_restarts=False):       # :1578-1848 -> (c, err)
        while True:
            c = self.next_marker()
            if c in (M_SOF0, M_SOF1, M_SOF2, M_SOF3, M_SOF5, M_SOF6, M_SOF7, M_SOF9, M_SOF10, M_SOF11, M_SOF13, M_SOF14, M_SOF15, M_SOI, M_EOI, M_SOS):
                return c, False
            if c == M_DHT:
                self.read_dht_marker()
            elif c == M_DAC:
                self.set_error("JPGD_NO_ARITHMITIC_SUPPORT")
            elif c == M_DQT:
                self.read_dqt_marker()
            elif c == M_DRI:
                self.read_dri_marker()
            elif c == M_APP0:              # :1634-1702
                num_left = self.get_bits(16)
                if num_left < 7:
                    self.set_error("JPGD_BAD_VARIABLE_MARKER")       # (the D carries on -- through 2^32 get_bits(8) -- with m_error_code set: null in the end)
                num_left -= 2
                jfif_id = [self.get_bits(8) for _ in range(5)]
                num_left -= 5
                if jfif_id == [0x4A, 0x46, 0x49, 0x46, 0x00] and num_left >= 7:
                    self.get_bits(16)
                    units = self.get_bits(8)
                    Xdensity = self.get_bits(16)
                    Ydensity = self.get_bits(16)
                    num_left -= 7
                    if Ydensity:
                        self.m_pixelAspectRatio = to_float(Xdensity / Ydensity)
                    else:
                        self.m_pixelAspectRatio = math.inf if Xdensity else NAN
                    if units == 0:
                        self.m_pixelsPerInchX = -1.0
                        self.m_pixelsPerInchY = -1.0
                    elif units == 1:
                        self.m_pixelsPerInchX = float(Xdensity)
                        self.m_pixelsPerInchY = float(Ydensity)
                    elif units == 2:
                        self.m_pixelsPerInchX = convertInchesToMeters(np.float32(Xdensity) * np.float32(100.0))
                        self.m_pixelsPerInchY = convertInchesToMeters(np.float32(Ydensity) * np.float32(100.0))
                while num_left:
                    self.get_bits(8)
                    num_left -= 1
            elif c == M_APP0 + 1:          # :1704-1816
                num_left = self.get_bits(16)
                if num_left < 2:
                    self.set_error("JPGD_BAD_VARIABLE_MARKER")
                num_left -= 2
                exifData = [self.get_bits(8) for _ in range(num_left)]
                self._exif(exifData)
            elif M_RST0 <= c <= M_RST7:    # :1818-1832
                if allow_restarts:
                    continue
                return 0, True
            elif c in (M_JPG, M_TEM):      # :1833-1838
                return 0, True
            else:
                self.skip_variable_marker()

    def _exif(self, exifData):             # the body of `case M_APP0+1` behind the read loop, :1728-1815
        pos = [0]

        def rd(n, at=None):                # read_ubyte / read_ushort_* / read_uint_* of internals/binop.d on a pointer into exifData
            p = pos if at is None else at
            if p[0] < 0 or p[0] + n > len(exifData):
                raise Undefined("EXIF: read of %d bytes at %d of a %d-byte segment" % (n, p[0], len(exifData)))
            v = exifData[p[0]:p[0] + n]
            p[0] += n
            return v

        def u16(le, at=None):
            b = rd(2, at)
            return b[0] | b[1] << 8 if le else b[0] << 8 | b[1]

        def u32(le, at=None):
            b = rd(4, at)
            return (b[0] | b[1] << 8 | b[2] << 16 | b[3] << 24) if le else (b[0] << 24 | b[1] << 16 | b[2] << 8 | b[3])

        exif_id = rd(6)
        if exif_id != [0x45, 0x78, 0x69, 0x66, 0x00, 0x00]:
            return
        EVENTS["EXIF segment"] += 1
        tiffFile = pos[0]
        byteOrder = u16(False)
        if byteOrder != 0x4949 and byteOrder != 0x4D4D:
            self.set_error("JPGD_DECODE_ERROR")
        littleEndian = byteOrder == 0x4949
        version_ = u16(littleEndian)
        if version_ != 42:
            self.set_error("JPGD_DECODE_ERROR")
        offset = u32(littleEndian)
        resolutionX = 72.0
        resolutionY = 72.0
        unit = 2
        seen = set()
        while offset != 0:
            if offset > len(exifData):
                self.set_error("JPGD_DECODE_ERROR")
            if offset in seen:
                raise Undefined("EXIF: the IFD chain loops (the reference never returns)")
            seen.add(offset)
            pIFD = [tiffFile + offset]
            numEntries = u16(littleEndian, pIFD)
            for _ in range(numEntries):
                tag = u16(littleEndian, pIFD)
                u16(littleEndian, pIFD)                          # type
                u32(littleEndian, pIFD)                          # count
                valueOffset = u32(littleEndian, pIFD)
                if tag == 282 or tag == 283:
                    tagData = [tiffFile + valueOffset]
                    num = float(u32(littleEndian, tagData))
                    denom = float(u32(littleEndian, tagData))
                    if denom:
                        frac = num / denom
                    else:
                        frac = math.inf if num else NAN
                    if tag == 282:
                        resolutionX = frac
                    else:
                        resolutionY = frac
                if tag == 296:
                    unit = i32(valueOffset)
            offset = u32(littleEndian, pIFD)

        def ddiv(a, b):                    # double / double
            if math.isnan(a) or math.isnan(b):
                return NAN
            if math.isinf(a) and math.isinf(b):
                return NAN
            if b == 0:
                return NAN if a == 0 else math.copysign(math.inf, a)
            if math.isinf(b):
                return 0.0
            return a / b

        if unit == 2:
            self.m_pixelsPerInchX = to_float(resolutionX)
            self.m_pixelsPerInchY = to_float(resolutionY)
            self.m_pixelAspectRatio = to_float(ddiv(resolutionX, resolutionY))
        elif unit == 3:
            self.m_pixelsPerInchX = convertInchesToMeters(to_float(resolutionX * 100))
            self.m_pixelsPerInchY = convertInchesToMeters(to_float(resolutionY * 100))
            self.m_pixelAspectRatio = to_float(ddiv(resolutionX, resolutionY))

    # ---- :1854-1967 ----------------------------------------------------------------------------------------------
    def locate_soi_marker(self):           # :1854-1908
        lastchar = self.get_bits(8)
        thischar = self.get_bits(8)
        if lastchar == 0xFF and thischar == M_SOI:
            return True
        bytesleft = 4096
        while True:
            bytesleft -= 1
            if bytesleft == 0:
                self.set_error("JPGD_NOT_JPEG")
            lastchar = thischar
            thischar = self.get_bits(8)
            if lastchar == 0xFF:
                if thischar == M_SOI:
                    break
                elif thischar == M_EOI:
                    self.set_error("JPGD_NOT_JPEG")
        thischar = (self.m_bit_buf >> 24) & 0xFF
        if thischar != 0xFF:
            self.set_error("JPGD_NOT_JPEG")
        return True

    def locate_sof_marker(self):           # :1911-1941
        self.locate_soi_marker()
        c, err = self.process_markers()
        if err:
            return False
        if c == M_SOF2:
            self.m_progressive_flag = True
            self.read_sof_marker()
        elif c in (M_SOF0, M_SOF1):
            self.read_sof_marker()
        elif c == M_SOF9:
            self.set_error("JPGD_NO_ARITHMITIC_SUPPORT")
        else:
            self.set_error("JPGD_UNSUPPORTED_MARKER")
        return True

    def locate_sos_marker(self):           # :1944-1967 -> (found, err)
        c, err = self.process_markers()
        if err:
            return False, True
        if c == M_EOI:
            return False, False
        elif c != M_SOS:
            self.set_error("JPGD_UNEXPECTED_MARKER")
        self.read_sos_marker()
        return True, False

    # ---- :1971-2118 ----------------------------------------------------------------------------------------------
    def initit(self, rfn):                 # :1971-2078
        self.m_error_code = JPGD_SUCCESS
        self.m_ready_flag = False
        self.m_image_x_size = self.m_image_y_size = 0
        self.readfn = rfn
        self.m_progressive_flag = False
        self.m_huff_ac = [0] * JPGD_MAX_HUFF_TABLES
        self.m_huff_num = [None] * JPGD_MAX_HUFF_TABLES
        self.m_huff_val = [None] * JPGD_MAX_HUFF_TABLES
        self.m_quant = [None] * JPGD_MAX_QUANT_TABLES
        self.m_scan_type = 0
        self.m_comps_in_frame = 0
        self.m_comp_h_samp = [0] * JPGD_MAX_COMPONENTS
        self.m_comp_v_samp = [0] * JPGD_MAX_COMPONENTS
        self.m_comp_quant = [0] * JPGD_MAX_COMPONENTS
        self.m_comp_ident = [0] * JPGD_MAX_COMPONENTS
        self.m_comp_h_blocks = [0] * JPGD_MAX_COMPONENTS
        self.m_comp_v_blocks = [0] * JPGD_MAX_COMPONENTS
        self.m_comps_in_scan = 0
        self.m_comp_list = [0] * JPGD_MAX_COMPS_IN_SCAN
        self.m_comp_dc_tab = [0] * JPGD_MAX_COMPONENTS
        self.m_comp_ac_tab = [0] * JPGD_MAX_COMPONENTS
        self.m_spectral_start = self.m_spectral_end = self.m_successive_low = self.m_successive_high = 0
        self.m_max_mcu_x_size = self.m_max_mcu_y_size = 0
        self.m_blocks_per_mcu = self.m_max_blocks_per_row = self.m_mcus_per_row = self.m_mcus_per_col = 0
        self.m_expanded_blocks_per_component = self.m_expanded_blocks_per_mcu = self.m_expanded_blocks_per_row = 0
        self.m_freq_domain_chroma_upsample = False
        self.m_mcu_org = [0] * JPGD_MAX_BLOCKS_PER_MCU
        self.m_total_lines_left = self.m_mcu_lines_left = 0
        self.m_pHuff_tabs = [None] * JPGD_MAX_HUFF_TABLES
        self.m_dc_coeffs = [None] * JPGD_MAX_COMPONENTS
        self.m_ac_coeffs = [None] * JPGD_MAX_COMPONENTS
        self.m_block_y_mcu = [0] * JPGD_MAX_COMPONENTS
        self.m_eob_run = 0
        self.m_pIn_buf_ofs = self.PAD
        self.m_in_buf_left = 0
        self.m_eof_flag = False
        self.m_tem_flag = 0
        self.mem = bytearray(128 + JPGD_IN_BUF_SIZE + 128 + 128)
        self.m_restart_interval = self.m_restarts_left = self.m_next_restart_num = 0
        self.m_max_mcus_per_row = self.m_max_blocks_per_mcu = self.m_max_mcus_per_col = 0
        self.m_last_dc_val = [0] * JPGD_MAX_COMPONENTS
        self.m_pMCU_coefficients = None
        self.m_total_bytes_read = 0
        self.prep_in_buffer()
        self.m_bits_left = 16
        self.m_bit_buf = 0
        self.get_bits(16)
        self.get_bits(16)
        self.m_mcu_block_max_zag = [64] * JPGD_MAX_BLOCKS_PER_MCU
        return True

    def fix_in_buffer(self):               # :2098-2118
        if self.m_bits_left & 7:
            raise Undefined("fix_in_buffer: assert((m_bits_left & 7) == 0)")
        if self.m_bits_left == 16:
            self.stuff_char(self.m_bit_buf & 0xFF)
        if self.m_bits_left >= 8:
            self.stuff_char((self.m_bit_buf >> 8) & 0xFF)
        self.stuff_char((self.m_bit_buf >> 16) & 0xFF)
        self.stuff_char((self.m_bit_buf >> 24) & 0xFF)
        self.m_bits_left = 16
        self.get_bits_no_markers(16)
        self.get_bits_no_markers(16)
        return True

    def transform_mcu(self, mcu_row):      # :2120-2130 / :2139-2255: what they are handed
        n = self.m_blocks_per_mcu
        if n != self.m_max_blocks_per_mcu or self.m_mcus_per_row != self.m_max_mcus_per_row:
            raise Undefined("the scan's MCU (%d blocks x %d) is not the frame's (%d x %d): m_pSample_buf is written in another layout than the "
                            "*Convert functions read" % (n, self.m_mcus_per_row, self.m_max_blocks_per_mcu, self.m_max_mcus_per_row))
        if any(v is None for v in self.m_pMCU_coefficients[:n * 64]):
            raise Undefined("m_pMCU_coefficients holds words nobody wrote")
        self.mcus.append((list(self.m_pMCU_coefficients[:n * 64]), list(self.m_mcu_block_max_zag[:n])))

    # ---- :2259-2525 ----------------------------------------------------------------------------------------------
    def load_next_row(self):               # :2259-2332
        block_x_mcu = [0] * JPGD_MAX_COMPONENTS
        for mcu_row in range(self.m_mcus_per_row):
            block_x_mcu_ofs = block_y_mcu_ofs = 0
            for mcu_block in range(self.m_blocks_per_mcu):
                component_id = self.m_mcu_org[mcu_block]
                q = self._quant(self.m_comp_quant[component_id])
                p = 64 * mcu_block
                if p + 64 > len(self.m_pMCU_coefficients):
                    raise Undefined("load_next_row: block %d of an MCU buffer of %d" % (mcu_block, len(self.m_pMCU_coefficients) // 64))
                pAC = self.coeff_buf_getp(self.m_ac_coeffs[component_id], block_x_mcu[component_id] + block_x_mcu_ofs, self.m_block_y_mcu[component_id] + block_y_mcu_ofs)
                pDC = self.coeff_buf_getp(self.m_dc_coeffs[component_id], block_x_mcu[component_id] + block_x_mcu_ofs, self.m_block_y_mcu[component_id] + block_y_mcu_ofs)
                co = self.m_pMCU_coefficients
                co[p] = self.m_dc_coeffs[component_id].pData[pDC]
                co[p + 1:p + 64] = self.m_ac_coeffs[component_id].pData[pAC + 1:pAC + 64]
                i = 63
                while i > 0:
                    if co[p + g_ZAG[i]]:
                        break
                    i -= 1
                self.m_mcu_block_max_zag[mcu_block] = i + 1
                while i >= 0:
                    if co[p + g_ZAG[i]]:
                        co[p + g_ZAG[i]] = i16(co[p + g_ZAG[i]] * q[i])
                    i -= 1
                if self.m_comps_in_scan == 1:
                    block_x_mcu[component_id] += 1
                else:
                    block_x_mcu_ofs += 1
                    if block_x_mcu_ofs == self.m_comp_h_samp[component_id]:
                        block_x_mcu_ofs = 0
                        block_y_mcu_ofs += 1
                        if block_y_mcu_ofs == self.m_comp_v_samp[component_id]:
                            block_y_mcu_ofs = 0
                            block_x_mcu[component_id] += self.m_comp_h_samp[component_id]
            self.transform_mcu(mcu_row)
        if self.m_comps_in_scan == 1:
            self.m_block_y_mcu[self.m_comp_list[0]] += 1
        else:
            for component_num in range(self.m_comps_in_scan):
                component_id = self.m_comp_list[component_num]
                self.m_block_y_mcu[component_id] += self.m_comp_v_samp[component_id]

    def process_restart(self):             # :2335-2402
        c = 0
        i = 1536
        while i > 0:
            if self.get_char() == 0xFF:
                break
            i -= 1
        if 0 < i < 1536:
            EVENTS["process_restart: bytes skipped in front of the marker"] += 1
        if i == 0:
            self.set_error("JPGD_BAD_RESTART_MARKER")
        while i > 0:
            c = self.get_char()
            if c != 0xFF:
                break
            i -= 1
        if i == 0:
            self.set_error("JPGD_BAD_RESTART_MARKER")
        if c != self.m_next_restart_num + M_RST0:
            self.set_error("JPGD_BAD_RESTART_MARKER")
        for k in range(self.m_comps_in_frame):
            self.m_last_dc_val[k] = 0
        self.m_eob_run = 0
        self.m_restarts_left = self.m_restart_interval
        self.m_next_restart_num = (self.m_next_restart_num + 1) & 7
        self.m_bits_left = 16
        self.get_bits_no_markers(16)
        self.get_bits_no_markers(16)
        return True

    def _quant(self, n):
        if not 0 <= n < JPGD_MAX_QUANT_TABLES:
            raise Undefined("m_quant[%d]" % n)
        q = self.m_quant[n]
        if q is None:
            raise Undefined("m_quant[%d] is null" % n)
        return q

    def _huff(self, n):
        if not 0 <= n < JPGD_MAX_HUFF_TABLES:
            raise Undefined("m_pHuff_tabs[%d]" % n)
        pH = self.m_pHuff_tabs[n]
        if pH is None:
            raise Undefined("m_pHuff_tabs[%d] is null" % n)
        return pH

    def decode_next_row(self):             # :2405-2525
        co = self.m_pMCU_coefficients
        for mcu_row in range(self.m_mcus_per_row):
            if self.m_restart_interval and self.m_restarts_left == 0:
                self.process_restart()
            p = 0
            for mcu_block in range(self.m_blocks_per_mcu):
                if p + 64 > len(co):
                    raise Undefined("decode_next_row: block %d of an MCU buffer of %d" % (mcu_block, len(co) // 64))
                component_id = self.m_mcu_org[mcu_block]
                q = self._quant(self.m_comp_quant[component_id])
                s, r = self.huff_decode2(self._huff(self.m_comp_dc_tab[component_id]))
                s = JPGD_HUFF_EXTEND(r, s)
                s = i32(s + self.m_last_dc_val[component_id])
                self.m_last_dc_val[component_id] = s & U32
                co[p] = i16(s * q[0])
                prev_num_set = self.m_mcu_block_max_zag[mcu_block]
                pH = self._huff(self.m_comp_ac_tab[component_id])
                k = 1
                while k < 64:
                    s, extra_bits = self.huff_decode2(pH)
                    r = s >> 4
                    s &= 15
                    if s:
                        if r:
                            if k + r > 63:
                                self.set_error("JPGD_DECODE_ERROR")
                            if k < prev_num_set:
                                n = min(r, prev_num_set - k)
                                kt = k
                                while n:
                                    n -= 1
                                    co[p + g_ZAG[kt]] = 0
                                    kt += 1
                            k += r
                        s = JPGD_HUFF_EXTEND(extra_bits, s)
                        co[p + g_ZAG[k]] = i16(s * q[k])
                    else:
                        if r == 15:
                            if k + 16 > 64:
                                self.set_error("JPGD_DECODE_ERROR")
                            if k < prev_num_set:
                                n = min(16, prev_num_set - k)
                                kt = k
                                while n:
                                    n -= 1
                                    co[p + g_ZAG[kt]] = 0
                                    kt += 1
                            k += 16 - 1
                        else:
                            break
                    k += 1
                if k < prev_num_set:
                    kt = k
                    while kt < prev_num_set:
                        co[p + g_ZAG[kt]] = 0
                        kt += 1
                self.m_mcu_block_max_zag[mcu_block] = k
                p += 64
            self.transform_mcu(mcu_row)
            self.m_restarts_left -= 1
        return True

    # ---- :2826-3125 ----------------------------------------------------------------------------------------------
    def find_eoi(self):                    # :2826-2848
        if not self.m_progressive_flag:
            self.m_bits_left = 16
            self.get_bits(16)
            self.get_bits(16)
            before = self.m_total_bytes_read - self.m_in_buf_left
            c, err = self.process_markers(True)
            if err:
                EVENTS["find_eoi: RSTn-tolerant walk fails on TEM / JPG"] += 1
                return False
            if c != M_EOI:
                EVENTS["find_eoi: ends at a marker other than EOI"] += 1
        self.m_total_bytes_read -= self.m_in_buf_left
        return True

    def make_huff_table(self, index, pH):  # :2851-2987
        huffsize = [0] * 257
        huffcode = [0] * 257
        pH.ac_table = self.m_huff_ac[index] != 0
        p = 0
        for l in range(1, 17):
            for i in range(1, self.m_huff_num[index][l] + 1):
                if p > 256:
                    raise Undefined("make_huff_table: huffsize[%d]" % p)
                huffsize[p] = l
                p += 1
        if p > 256:
            raise Undefined("make_huff_table: huffsize[%d]" % p)
        huffsize[p] = 0
        lastp = p
        if sum(self.m_huff_num[index][l] << (16 - l) for l in range(1, 17)) > 1 << 16:
            # codes run past 2^length: for lengths <= 8 `pH.look_up.ptr[code]` is written behind the array (:2917), longer ones alias other
            # prefixes through `& 0xFF` (:2946) and overwrite tree nodes -- the table is whatever those writes leave
            raise Undefined("make_huff_table: the code lengths over-subscribe the code space")
        code = 0
        si = huffsize[0]
        p = 0
        while huffsize[p]:
            while huffsize[p] == si:
                huffcode[p] = code
                p += 1
                code += 1
            code = (code << 1) & U32
            si += 1
        pH.look_up = [0] * 256
        pH.look_up2 = [0] * 256
        pH.tree = [0] * 512
        pH.code_size = [0] * 256
        nextfreeentry = -1
        p = 0
        while p < lastp:
            i = self.m_huff_val[index][p]
            code = huffcode[p]
            code_size = huffsize[p]
            pH.code_size[i] = code_size
            if code_size <= 8:
                code = (code << (8 - code_size)) & U32
                l = 1 << (8 - code_size)
                while l > 0:
                    if code > 255:
                        raise Undefined("make_huff_table: look_up[%d] (the code lengths over-subscribe the code space)" % code)
                    pH.look_up[code] = i
                    has_extrabits = False
                    extra_bits = 0
                    num_extra_bits = i & 15
                    bits_to_fetch = code_size
                    if num_extra_bits:
                        total_codesize = code_size + num_extra_bits
                        if total_codesize <= 8:
                            has_extrabits = True
                            extra_bits = ((1 << num_extra_bits) - 1) & (code >> (8 - total_codesize))
                            bits_to_fetch += num_extra_bits
                    if not has_extrabits:
                        pH.look_up2[code] = i | (bits_to_fetch << 8)
                    else:
                        pH.look_up2[code] = i | 0x8000 | (extra_bits << 16) | (bits_to_fetch << 8)
                    code += 1
                    l -= 1
            else:
                subtree = (code >> (code_size - 8)) & 0xFF
                currententry = i32(pH.look_up[subtree])
                if currententry == 0:
                    pH.look_up[subtree] = currententry = nextfreeentry
                    pH.look_up2[subtree] = currententry = nextfreeentry
                    nextfreeentry -= 2
                elif currententry > 0:
                    raise Undefined("make_huff_table: a code longer than 8 bits under the prefix of a shorter one (not a prefix code)")
                code = (code << (16 - (code_size - 8))) & U32
                l = code_size
                while l > 9:
                    if (code & 0x8000) == 0:
                        currententry -= 1
                    if not 0 <= -currententry - 1 < 512:
                        raise Undefined("make_huff_table: tree[%d]" % (-currententry - 1))
                    if pH.tree[-currententry - 1] == 0:
                        pH.tree[-currententry - 1] = nextfreeentry
                        currententry = nextfreeentry
                        nextfreeentry -= 2
                    else:
                        currententry = i32(pH.tree[-currententry - 1])
                        if currententry > 0:
                            raise Undefined("make_huff_table: walks through a leaf (not a prefix code)")
                    code = (code << 1) & U32
                    l -= 1
                if (code & 0x8000) == 0:
                    currententry -= 1
                if not 0 <= -currententry - 1 < 512:
                    raise Undefined("make_huff_table: tree[%d]" % (-currententry - 1))
                pH.tree[-currententry - 1] = i
            p += 1

    def check_quant_tables(self):          # :2990-3000
        for i in range(self.m_comps_in_scan):
            n = self.m_comp_quant[self.m_comp_list[i]]
            if not 0 <= n < JPGD_MAX_QUANT_TABLES:
                raise Undefined("m_quant[%d]" % n)
            if self.m_quant[n] is None:
                self.set_error("JPGD_UNDEFINED_QUANT_TABLE")
        return True

    def check_huff_tables(self):           # :3003-3034
        for i in range(self.m_comps_in_scan):
            d, a = self.m_comp_dc_tab[self.m_comp_list[i]], self.m_comp_ac_tab[self.m_comp_list[i]]
            if self.m_spectral_start == 0:
                if d >= JPGD_MAX_HUFF_TABLES:
                    raise Undefined("m_huff_num[%d]" % d)
                if self.m_huff_num[d] is None:
                    self.set_error("JPGD_UNDEFINED_HUFF_TABLE")
            if self.m_spectral_end > 0:
                if a >= JPGD_MAX_HUFF_TABLES:
                    raise Undefined("m_huff_num[%d]" % a)
                if self.m_huff_num[a] is None:
                    self.set_error("JPGD_UNDEFINED_HUFF_TABLE")
        for i in range(JPGD_MAX_HUFF_TABLES):
            if self.m_huff_num[i] is not None:
                if self.m_pHuff_tabs[i] is None:
                    self.m_pHuff_tabs[i] = huff_tables()
                self.make_huff_table(i, self.m_pHuff_tabs[i])
        return True

    def calc_mcu_block_order(self):        # :3038-3090
        max_h_samp = max_v_samp = 0
        for component_id in range(self.m_comps_in_frame):
            if self.m_comp_h_samp[component_id] > max_h_samp:
                max_h_samp = self.m_comp_h_samp[component_id]
            if self.m_comp_v_samp[component_id] > max_v_samp:
                max_v_samp = self.m_comp_v_samp[component_id]
        for component_id in range(self.m_comps_in_frame):
            self.m_comp_h_blocks[component_id] = ((((self.m_image_x_size * self.m_comp_h_samp[component_id]) + (max_h_samp - 1)) // max_h_samp) + 7) // 8
            self.m_comp_v_blocks[component_id] = ((((self.m_image_y_size * self.m_comp_v_samp[component_id]) + (max_v_samp - 1)) // max_v_samp) + 7) // 8
        if self.m_comps_in_scan == 1:
            self.m_mcus_per_row = self.m_comp_h_blocks[self.m_comp_list[0]]
            self.m_mcus_per_col = self.m_comp_v_blocks[self.m_comp_list[0]]
        else:
            self.m_mcus_per_row = (((self.m_image_x_size + 7) // 8) + (max_h_samp - 1)) // max_h_samp
            self.m_mcus_per_col = (((self.m_image_y_size + 7) // 8) + (max_v_samp - 1)) // max_v_samp
        if self.m_comps_in_scan == 1:
            self.m_mcu_org[0] = self.m_comp_list[0]
            self.m_blocks_per_mcu = 1
        else:
            self.m_blocks_per_mcu = 0
            for component_num in range(self.m_comps_in_scan):
                component_id = self.m_comp_list[component_num]
                num_blocks = self.m_comp_h_samp[component_id] * self.m_comp_v_samp[component_id]
                while num_blocks:
                    num_blocks -= 1
                    if self.m_blocks_per_mcu >= JPGD_MAX_BLOCKS_PER_MCU:
                        raise Undefined("m_mcu_org[%d]" % self.m_blocks_per_mcu)
                    self.m_mcu_org[self.m_blocks_per_mcu] = component_id
                    self.m_blocks_per_mcu += 1

    def init_scan(self):                   # :3093-3125 -> (found, err)
        found, err = self.locate_sos_marker()
        if not found:
            return False, err
        self.calc_mcu_block_order()
        self.check_huff_tables()
        self.check_quant_tables()
        for k in range(self.m_comps_in_frame):
            self.m_last_dc_val[k] = 0
        self.m_eob_run = 0
        if self.m_restart_interval:
            self.m_restarts_left = self.m_restart_interval
            self.m_next_restart_num = 0
        self.fix_in_buffer()
        return True, False

    # ---- :3130-3295 ----------------------------------------------------------------------------------------------
    def init_frame(self):                  # :3130-3268
        hs, vs = self.m_comp_h_samp, self.m_comp_v_samp
        if self.m_comps_in_frame == 1:
            if hs[0] != 1 or vs[0] != 1:
                self.set_error("JPGD_UNSUPPORTED_SAMP_FACTORS")
            self.m_scan_type = JPGD_GRAYSCALE
            self.m_max_blocks_per_mcu = 1
            self.m_max_mcu_x_size = 8
            self.m_max_mcu_y_size = 8
        elif self.m_comps_in_frame == 3:
            if (hs[1] != 1 or vs[1] != 1) or (hs[2] != 1 or vs[2] != 1):
                self.set_error("JPGD_UNSUPPORTED_SAMP_FACTORS")
            if hs[0] == 1 and vs[0] == 1:
                self.m_scan_type, self.m_max_blocks_per_mcu, self.m_max_mcu_x_size, self.m_max_mcu_y_size = JPGD_YH1V1, 3, 8, 8
            elif hs[0] == 2 and vs[0] == 1:
                self.m_scan_type, self.m_max_blocks_per_mcu, self.m_max_mcu_x_size, self.m_max_mcu_y_size = JPGD_YH2V1, 4, 16, 8
            elif hs[0] == 1 and vs[0] == 2:
                self.m_scan_type, self.m_max_blocks_per_mcu, self.m_max_mcu_x_size, self.m_max_mcu_y_size = JPGD_YH1V2, 4, 8, 16
            elif hs[0] == 2 and vs[0] == 2:
                self.m_scan_type, self.m_max_blocks_per_mcu, self.m_max_mcu_x_size, self.m_max_mcu_y_size = JPGD_YH2V2, 6, 16, 16
            else:
                self.set_error("JPGD_UNSUPPORTED_SAMP_FACTORS")
        else:
            self.set_error("JPGD_UNSUPPORTED_COLORSPACE")
        self.m_max_mcus_per_row = (self.m_image_x_size + (self.m_max_mcu_x_size - 1)) // self.m_max_mcu_x_size
        self.m_max_mcus_per_col = (self.m_image_y_size + (self.m_max_mcu_y_size - 1)) // self.m_max_mcu_y_size
        self.m_dest_bytes_per_pixel = 1 if self.m_scan_type == JPGD_GRAYSCALE else 4
        self.m_max_blocks_per_row = self.m_max_mcus_per_row * self.m_max_blocks_per_mcu
        if self.m_max_blocks_per_row > JPGD_MAX_BLOCKS_PER_ROW:
            self.set_error("JPGD_ASSERTION_ERROR")
        self.m_pMCU_coefficients = [None] * (self.m_max_blocks_per_mcu * 64)       # alloc(..., false): not cleared
        for i in range(self.m_max_blocks_per_mcu):
            self.m_mcu_block_max_zag[i] = 64
        self.m_expanded_blocks_per_component = hs[0] * vs[0]
        self.m_expanded_blocks_per_mcu = self.m_expanded_blocks_per_component * self.m_comps_in_frame
        self.m_expanded_blocks_per_row = self.m_max_mcus_per_row * self.m_expanded_blocks_per_mcu
        self.m_freq_domain_chroma_upsample = self.m_expanded_blocks_per_mcu == 4 * 3
        self.m_total_lines_left = self.m_image_y_size
        self.m_mcu_lines_left = 0
        return True

    def coeff_buf_open(self, block_num_x, block_num_y, block_len_x, block_len_y):      # :3274-3290
        cb = coeff_buf()
        cb.block_num_x, cb.block_num_y, cb.block_len_x, cb.block_len_y = block_num_x, block_num_y, block_len_x, block_len_y
        cb.block_size = block_len_x * block_len_y                                        # in jpgd_block_t words
        cb.pData = [0] * (cb.block_size * block_num_x * block_num_y)
        return cb

    def coeff_buf_getp(self, cb, block_x, block_y):        # :3292-3295 -> word offset into cb.pData
        if not (0 <= block_x < cb.block_num_x and 0 <= block_y < cb.block_num_y):
            raise Undefined("coeff_buf_getp: assert((block_x < cb.block_num_x) && (block_y < cb.block_num_y)): (%d, %d) of (%d, %d)"
                            % (block_x, block_y, cb.block_num_x, cb.block_num_y))
        return block_x * cb.block_size + block_y * (cb.block_size * cb.block_num_x)

    # ---- :3299-3518 ----------------------------------------------------------------------------------------------
    def decode_block_dc_first(self, component_id, block_x, block_y):       # :3299-3320
        cb = self.m_dc_coeffs[component_id]
        p = self.coeff_buf_getp(cb, block_x, block_y)
        s = self.huff_decode(self._huff(self.m_comp_dc_tab[component_id]))
        if s != 0:
            if s > 15:
                raise Undefined("decode_block_dc_first: a DC category of %d" % s)
            r = self.get_bits_no_markers(s)
            s = JPGD_HUFF_EXTEND(r, s)
        s = i32(s + self.m_last_dc_val[component_id])
        self.m_last_dc_val[component_id] = s & U32
        cb.pData[p] = i16(s << self.m_successive_low)
        return True

    def decode_block_dc_refine(self, component_id, block_x, block_y):      # :3322-3333
        if self.get_bits_no_markers(1):
            cb = self.m_dc_coeffs[component_id]
            p = self.coeff_buf_getp(cb, block_x, block_y)
            cb.pData[p] = i16(cb.pData[p] | (1 << self.m_successive_low))
        return True

    def decode_block_ac_first(self, component_id, block_x, block_y):       # :3335-3398
        if self.m_eob_run:
            self.m_eob_run -= 1
            return True
        cb = self.m_ac_coeffs[component_id]
        p = self.coeff_buf_getp(cb, block_x, block_y)
        pH = self._huff(self.m_comp_ac_tab[component_id])
        k = self.m_spectral_start
        while k <= self.m_spectral_end:
            s = self.huff_decode(pH)
            r = s >> 4
            s &= 15
            if s:
                k += r
                if k > 63:
                    self.set_error("JPGD_DECODE_ERROR")
                r = self.get_bits_no_markers(s)
                s = JPGD_HUFF_EXTEND(r, s)
                cb.pData[p + g_ZAG[k]] = i16(s << self.m_successive_low)
            else:
                if r == 15:
                    k += 15
                    if k > 63:
                        self.set_error("JPGD_DECODE_ERROR")
                else:
                    self.m_eob_run = 1 << r
                    if r:
                        self.m_eob_run += self.get_bits_no_markers(r)
                    self.m_eob_run -= 1
                    break
            k += 1
        return True

    def decode_block_ac_refine(self, component_id, block_x, block_y):      # :3400-3518
        p1 = 1 << self.m_successive_low
        m1 = i32((-1) << self.m_successive_low)
        cb = self.m_ac_coeffs[component_id]
        p = self.coeff_buf_getp(cb, block_x, block_y)
        d = cb.pData
        if self.m_spectral_end > 63:
            raise Undefined("decode_block_ac_refine: assert(pD.m_spectral_end <= 63)")
        k = self.m_spectral_start
        pH = self._huff(self.m_comp_ac_tab[component_id])

        def correct(at):
            if self.get_bits_no_markers(1):
                if (d[at] & p1) == 0:
                    if d[at] >= 0:
                        d[at] = i16(d[at] + p1)
                    else:
                        d[at] = i16(d[at] + m1)

        if self.m_eob_run == 0:
            while k <= self.m_spectral_end:
                s = self.huff_decode(pH)
                r = s >> 4
                s &= 15
                brk = False
                if s:
                    if s != 1:
                        self.set_error("JPGD_DECODE_ERROR")
                    if self.get_bits_no_markers(1):
                        s = p1
                    else:
                        s = m1
                else:
                    if r != 15:
                        self.m_eob_run = 1 << r
                        if r:
                            self.m_eob_run += self.get_bits_no_markers(r)
                        brk = True
                if brk:
                    break
                while True:
                    this_coef = p + g_ZAG[k & 63]
                    if d[this_coef] != 0:
                        correct(this_coef)
                    else:
                        r -= 1
                        if r < 0:
                            break
                    k += 1
                    if not k <= self.m_spectral_end:
                        break
                if s and k < 64:
                    d[p + g_ZAG[k]] = i16(s)
                k += 1
        if self.m_eob_run > 0:
            while k <= self.m_spectral_end:
                this_coef = p + g_ZAG[k & 63]
                if d[this_coef] != 0:
                    correct(this_coef)
                k += 1
            self.m_eob_run -= 1
        return True

    # ---- :3521-3713 ----------------------------------------------------------------------------------------------
    def decode_scan(self, decode_block_func):              # :3521-3584
        m_block_y_mcu = [0] * JPGD_MAX_COMPONENTS            # the local that shadows the member
        for mcu_col in range(self.m_mcus_per_col):
            block_x_mcu = [0] * JPGD_MAX_COMPONENTS
            for mcu_row in range(self.m_mcus_per_row):
                block_x_mcu_ofs = block_y_mcu_ofs = 0
                if self.m_restart_interval and self.m_restarts_left == 0:
                    self.process_restart()
                for mcu_block in range(self.m_blocks_per_mcu):
                    component_id = self.m_mcu_org[mcu_block]
                    decode_block_func(component_id, block_x_mcu[component_id] + block_x_mcu_ofs, m_block_y_mcu[component_id] + block_y_mcu_ofs)
                    if self.m_comps_in_scan == 1:
                        block_x_mcu[component_id] += 1
                    else:
                        block_x_mcu_ofs += 1
                        if block_x_mcu_ofs == self.m_comp_h_samp[component_id]:
                            block_x_mcu_ofs = 0
                            block_y_mcu_ofs += 1
                            if block_y_mcu_ofs == self.m_comp_v_samp[component_id]:
                                block_y_mcu_ofs = 0
                                block_x_mcu[component_id] += self.m_comp_h_samp[component_id]
                self.m_restarts_left -= 1
            if self.m_comps_in_scan == 1:
                m_block_y_mcu[self.m_comp_list[0]] += 1
            else:
                for component_num in range(self.m_comps_in_scan):
                    component_id = self.m_comp_list[component_num]
                    m_block_y_mcu[component_id] += self.m_comp_v_samp[component_id]
        return True

    def init_progressive(self):            # :3587-3683
        if self.m_comps_in_frame == 4:
            self.set_error("JPGD_UNSUPPORTED_COLORSPACE")
        for i in range(self.m_comps_in_frame):
            self.m_dc_coeffs[i] = self.coeff_buf_open(self.m_max_mcus_per_row * self.m_comp_h_samp[i], self.m_max_mcus_per_col * self.m_comp_v_samp[i], 1, 1)
            self.m_ac_coeffs[i] = self.coeff_buf_open(self.m_max_mcus_per_row * self.m_comp_h_samp[i], self.m_max_mcus_per_col * self.m_comp_v_samp[i], 8, 8)
        while True:
            scanInit, err = self.init_scan()
            if err:
                return False
            if not scanInit:
                break
            dc_only_scan = self.m_spectral_start == 0
            refinement_scan = self.m_successive_high != 0
            if self.m_spectral_start > self.m_spectral_end or self.m_spectral_end > 63:
                self.set_error("JPGD_BAD_SOS_SPECTRAL")
            if dc_only_scan:
                if self.m_spectral_end:
                    self.set_error("JPGD_BAD_SOS_SPECTRAL")
            elif self.m_comps_in_scan != 1:
                self.set_error("JPGD_BAD_SOS_SPECTRAL")
            if refinement_scan and self.m_successive_low != self.m_successive_high - 1:
                self.set_error("JPGD_BAD_SOS_SUCCESSIVE")
            if dc_only_scan:
                decode_block_func = self.decode_block_dc_refine if refinement_scan else self.decode_block_dc_first
            else:
                decode_block_func = self.decode_block_ac_refine if refinement_scan else self.decode_block_ac_first
            self.decode_scan(decode_block_func)
            self.m_bits_left = 16
            self.get_bits(16)
            self.get_bits(16)
        self.m_comps_in_scan = self.m_comps_in_frame
        for i in range(self.m_comps_in_frame):
            self.m_comp_list[i] = i
        self.calc_mcu_block_order()
        return True

    def init_sequential(self):             # :3685-3695
        found, err = self.init_scan()
        if not found:
            self.set_error("JPGD_UNEXPECTED_MARKER")
        return True

    def decode_start(self):                # :3697-3706
        self.init_frame()
        if self.m_progressive_flag:
            return self.init_progressive()
        return self.init_sequential()

    def decode_init(self, rfn):            # :3708-3713
        self.initit(rfn)
        return self.locate_sof_marker()

    # ---- :530-612 ------------------------------------------------------------------------------------------------
    def begin_decoding(self):              # :530-537
        if self.m_ready_flag:
            return JPGD_SUCCESS
        if self.m_error_code:
            return JPGD_FAILED
        ok = self.decode_start()           # the D drops the result
        if not ok:
            # init_progressive left through `if (err) return false` (:3615) with no error code: RSTn / TEM / JPG between two scans (:1818-1838).
            # decode() then runs load_next_row on whatever the scans so far left behind, in the block order of the LAST scan
            raise Undefined("init_progressive gave up without an error code: decode() goes on with the state of the last scan")
        self.m_ready_flag = True
        return JPGD_SUCCESS

    def decode(self):                      # :545-612, up to the colour conversion
        if self.m_error_code or not self.m_ready_flag:
            return JPGD_FAILED
        if self.m_total_lines_left == 0:
            return JPGD_DONE
        if self.m_mcu_lines_left == 0:
            if self.m_progressive_flag:
                self.load_next_row()
            else:
                self.decode_next_row()
            if self.m_total_lines_left <= self.m_max_mcu_y_size:
                if not self.find_eoi():
                    return JPGD_FAILED
            self.m_mcu_lines_left = self.m_max_mcu_y_size
        self.m_mcu_lines_left -= 1
        self.m_total_lines_left -= 1
        return JPGD_SUCCESS


def decompress_jpeg_image_from_stream(data, req_comps=-1):      # :3720-3808 without the pixels
    """-> None (null) or dict(width, height, actual_comps, pixelAspectRatio, dotsPerInchY, scan_type, mcus=[(coefficients, max_zag), ...] in
    transform order); raises Undefined where the reference has no defined result"""
    if req_comps not in (-1, 1, 3, 4):
        return None
    try:
        decoder = jpeg_decoder(MemoryStream(data))
        if not decoder.success:
            return None
        if decoder.m_error_code != JPGD_SUCCESS:
            return None
        image_height = decoder.m_image_y_size
        if decoder.begin_decoding() != JPGD_SUCCESS:
            return None
        for y in range(image_height):
            if decoder.decode() != JPGD_SUCCESS:
                return None
    except Rejected:
        return None
    return dict(width=decoder.m_image_x_size, height=image_height, actual_comps=decoder.m_comps_in_frame,
                pixelAspectRatio=decoder.m_pixelAspectRatio, dotsPerInchY=decoder.m_pixelsPerInchY,
                scan_type=decoder.m_scan_type, progressive=bool(decoder.m_progressive_flag), mcus=decoder.mcus,
                total_bytes_read=decoder.m_total_bytes_read)


if __name__ == "__main__":
    import sys
    for path in sys.argv[1:]:
        try:
            r = decompress_jpeg_image_from_stream(open(path, "rb").read())
        except Undefined as e:
            print(path, "UNDEFINED:", e)
            continue
        if r is None:
            print(path, "null")
        else:
            print(path, {k: v for k, v in r.items() if k != "mcus"}, len(r["mcus"]), "MCUs")
