// jpeg_host.hip -- host side of the JPEG path: the baseline entropy-decode feeder that
// produces the dense coefficient form the kernels consume, and the C-ABI entry points.
//
// The feeder does what jpeg_decoder::decode_next_row does on the CPU in the reference
// (jpegload.d:2405-2525: Huffman decode, DC prediction, de-quantise `s * q[k]` into natural
// order through g_ZAG, track m_mcu_block_max_zag), for a whole image at once, after the
// marker pass of :1160-1848 (DQT tables kept in zig-zag order as int16, :1313-1329; DHT;
// SOF0/SOF1 8-bit; DRI; SOS).  It stays on the host: bit-serial work.
// Not supported here (the call fails like the reference fails on a stream it rejects):
// progressive / arithmetic / lossless frames, multi-scan baseline files.
#include "common.hpp"
#include <new>

namespace gamut {
namespace {

const uint8_t kZag[64] = { 0,1,8,16,9,2,3,10,17,24,32,25,18,11,4,5,12,19,26,33,40,48,41,34,27,20,13,6,7,14,21,28,35,42,49,56,57,50,43,36,29,22,15,23,30,37,44,51,58,59,52,45,38,31,39,46,53,60,61,54,47,55,62,63 };

struct HuffTable {
    bool    defined = false;
    uint8_t bits[17] = {};        // number of codes per length
    uint8_t vals[256] = {};
    // fast path: 10-bit window -> (code length << 8) | symbol, 0 when the code is longer
    uint16_t fast[1024];
    // slow path: canonical ranges
    int32_t  maxcode[18];
    int32_t  delta[17];           // valptr - mincode
    void build()
    {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            delta[len] = k - code;
            code += bits[len]; k += bits[len];
            maxcode[len] = bits[len] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = INT32_MAX;
        memset(fast, 0, sizeof(fast));
        code = 0; k = 0;
        for (int len = 1; len <= 10; ++len) {
            for (int i = 0; i < bits[len]; ++i, ++k, ++code) {
                const int first = code << (10 - len), n = 1 << (10 - len);
                for (int f = 0; f < n; ++f) fast[first + f] = (uint16_t)((len << 8) | vals[k]);
            }
            code <<= 1;
        }
    }
};

// MSB-first bit reader over the entropy-coded segment; handles FF00 stuffing and stops
// feeding real data at a marker (subsequent bits read as 1s, like get_octet :683-696).
struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint64_t acc = 0; int nbits = 0; bool at_marker = false;
    BitReader(const uint8_t* b, const uint8_t* e) : p(b), end(e) {}
    inline void refill()
    {
        while (nbits <= 56) {
            uint32_t c = 0xFF;
            if (!at_marker && p < end) {
                c = *p;
                if (c == 0xFF) {
                    if (p + 1 < end && p[1] == 0) p += 2;
                    else { at_marker = true; }
                } else ++p;
            }
            acc = (acc << 8) | c; nbits += 8;
        }
    }
    inline uint32_t peek(int n) const { return (uint32_t)(acc >> (nbits - n)) & ((1u << n) - 1); }
    inline void drop(int n) { nbits -= n; }
    inline int receive_extend(int s)          // JPGD_HUFF_EXTEND :816-822
    {
        if (!s) return 0;
        if (nbits < s) refill();
        const int v = (int)peek(s); drop(s);
        return v < (1 << (s - 1)) ? v + (int)(0xFFFFFFFFu << s) + 1 : v;
    }
    inline int decode(const HuffTable& h)
    {
        if (nbits < 16) refill();
        const uint16_t e = h.fast[peek(10)];
        if (e) { drop(e >> 8); return e & 0xFF; }
        int32_t code = (int32_t)peek(10); int len = 10;
        while (len < 17 && code > h.maxcode[len]) { ++len; code = (int32_t)peek(len); }
        if (len > 16) return -1;
        drop(len);
        return h.vals[(code + h.delta[len]) & 0xFF];
    }
    void restart(const uint8_t* np) { p = np; acc = 0; nbits = 0; at_marker = false; }
};

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

struct Parser {
    int16_t   quant[4][64]; bool quant_def[4] = { false, false, false, false };
    HuffTable huff[8];                 // 0-3 DC, 4-7 AC (index mapping of read_dht_marker :1247)
    int comp_id[3] = {}, hs[3] = {}, vs[3] = {}, tq[3] = {}, td[3] = {}, ta[3] = {};
    int restart_interval = 0;
};

int fail(gamut_hip_jpeg_frame* f, const char* why)
{
    free(f->coeffs); free(f->max_zag); f->coeffs = nullptr; f->max_zag = nullptr;
    return set_error(GAMUT_HIP_ERR_DECODE, "jpeg: %s", why);
}

int decode_coeffs(const uint8_t* data, size_t len, gamut_hip_jpeg_frame* f)
{
    memset(f, 0, sizeof(*f));
    f->pixel_aspect_ratio = -1; f->dpi_y = -1;
    if (!data || len < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail(f, "not a JPEG (no SOI)");

    Parser* ps = new (std::nothrow) Parser();
    if (!ps) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: out of memory");
    struct Guard { Parser* p; ~Guard() { delete p; } } guard{ ps };
    Parser& P = *ps;

    size_t pos = 2; bool have_sof = false; const uint8_t* scan = nullptr;
    while (!scan) {
        while (pos < len && data[pos] != 0xFF) ++pos;          // next_marker :1544-1572
        while (pos < len && data[pos] == 0xFF) ++pos;
        if (pos >= len) return fail(f, "no SOS marker");
        const int m = data[pos++];
        if (m == 0x00 || m == 0x01 || m == 0xD8 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) return fail(f, "EOI before SOS");
        if (pos + 2 > len) return fail(f, "truncated marker");
        const int seg = be16(data + pos);
        if (seg < 2 || pos + (size_t)seg > len) return fail(f, "bad marker length");
        const uint8_t* s = data + pos + 2; int n = seg - 2;
        switch (m) {
        case 0xDB:                                             // DQT :1274-1346
            while (n > 0) {
                const int prec = s[0] >> 4, id = s[0] & 15; ++s; --n;
                if (id >= 4 || n < (prec ? 128 : 64)) return fail(f, "bad DQT");
                for (int i = 0; i < 64; ++i) {
                    uint32_t v = *s++; if (prec) v = (v << 8) + *s++;
                    P.quant[id][i] = (int16_t)v;               // stored as `short`, zig-zag order
                }
                n -= prec ? 128 : 64; P.quant_def[id] = true;
            }
            break;
        case 0xC4:                                             // DHT :1173-1270
            while (n > 0) {
                if (n < 17) return fail(f, "bad DHT");
                const int idx = (s[0] & 0x0F) + ((s[0] & 0x10) >> 4) * 4;
                if (idx >= 8) return fail(f, "bad DHT index");
                HuffTable& h = P.huff[idx];
                int cnt = 0; h.bits[0] = 0;
                for (int i = 1; i <= 16; ++i) { h.bits[i] = s[i]; cnt += s[i]; }
                s += 17; n -= 17;
                if (cnt > 255 || n < cnt) return fail(f, "bad DHT counts");
                memset(h.vals, 0, sizeof(h.vals)); memcpy(h.vals, s, (size_t)cnt);
                s += cnt; n -= cnt; h.defined = true; h.build();
            }
            break;
        case 0xC0: case 0xC1: {                                // SOF0 / SOF1 :1349-1417
            if (n < 6) return fail(f, "bad SOF");
            if (s[0] != 8) return fail(f, "only 8-bit precision is supported");
            f->height = be16(s + 1); f->width = be16(s + 3); f->comps = s[5];
            if (f->height < 1 || f->height > 16384) return fail(f, "bad height");
            if (f->width < 1 || f->width > 16384) return fail(f, "bad width");
            if (f->comps != 1 && f->comps != 3) return fail(f, "unsupported colorspace");       // init_frame :3191-3195
            if (n != f->comps * 3 + 6) return fail(f, "bad SOF length");
            for (int i = 0; i < f->comps; ++i) {
                P.comp_id[i] = s[6 + 3 * i]; P.hs[i] = s[7 + 3 * i] >> 4; P.vs[i] = s[7 + 3 * i] & 15; P.tq[i] = s[8 + 3 * i];
                if (P.tq[i] >= 4) return fail(f, "bad quant table selector");
            }
            // init_frame :3134-3190
            if (f->comps == 1) {
                if (P.hs[0] != 1 || P.vs[0] != 1) return fail(f, "unsupported sampling factors");
                f->scan_type = GAMUT_JPGD_GRAYSCALE; f->blocks_per_mcu = 1;
            } else {
                if (P.hs[1] != 1 || P.vs[1] != 1 || P.hs[2] != 1 || P.vs[2] != 1) return fail(f, "unsupported sampling factors");
                if      (P.hs[0] == 1 && P.vs[0] == 1) { f->scan_type = GAMUT_JPGD_YH1V1; f->blocks_per_mcu = 3; }
                else if (P.hs[0] == 2 && P.vs[0] == 1) { f->scan_type = GAMUT_JPGD_YH2V1; f->blocks_per_mcu = 4; }
                else if (P.hs[0] == 1 && P.vs[0] == 2) { f->scan_type = GAMUT_JPGD_YH1V2; f->blocks_per_mcu = 4; }
                else if (P.hs[0] == 2 && P.vs[0] == 2) { f->scan_type = GAMUT_JPGD_YH2V2; f->blocks_per_mcu = 6; }
                else return fail(f, "unsupported sampling factors");
            }
            const int mw = 8 * (f->comps == 3 ? P.hs[0] : 1), mh = 8 * (f->comps == 3 ? P.vs[0] : 1);
            f->mcus_per_row = (f->width + mw - 1) / mw; f->mcus_per_col = (f->height + mh - 1) / mh;
            have_sof = true;
        } break;
        case 0xC2: case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            return fail(f, "progressive / lossless / arithmetic frames are not supported by the GPU feeder");
        case 0xCC: return fail(f, "arithmetic coding is not supported");
        case 0xDD:                                             // DRI :1445-1462
            if (seg != 4) return fail(f, "bad DRI length");
            P.restart_interval = be16(s);
            break;
        case 0xE0:                                             // APP0 / JFIF density :1632-1690
            if (n >= 12 && !memcmp(s, "JFIF\0", 5)) {
                const int units = s[7], xd = be16(s + 8), yd = be16(s + 10);
                f->pixel_aspect_ratio = (float)(xd / (double)yd);
                if (units == 0) f->dpi_y = -1;
                else if (units == 1) f->dpi_y = (float)yd;
                else if (units == 2) f->dpi_y = (yd * 100.0f) / 39.37007874f;
            }
            break;
        case 0xDA: {                                           // SOS :1466-1540
            if (!have_sof) return fail(f, "SOS before SOF");
            if (n < 1 || s[0] != f->comps || n != f->comps * 2 + 4) return fail(f, "only single-scan baseline files are supported");
            for (int i = 0; i < f->comps; ++i) {
                int ci = 0; while (ci < f->comps && P.comp_id[ci] != s[1 + 2 * i]) ++ci;
                if (ci >= f->comps) return fail(f, "bad SOS component id");
                P.td[ci] = (s[2 + 2 * i] >> 4) & 15; P.ta[ci] = (s[2 + 2 * i] & 15) + 4;
                if (P.td[ci] >= 4 || P.ta[ci] >= 8) return fail(f, "bad Huffman table selector");
            }
            scan = data + pos + seg;
        } break;
        default: break;                                        // APPn / COM / unknown: skipped (:1826-1846)
        }
        pos += (size_t)seg;
    }

    for (int c = 0; c < f->comps; ++c) {                       // check_quant_tables / check_huff_tables :2990-3034
        if (!P.quant_def[P.tq[c]]) return fail(f, "undefined quant table");
        if (!P.huff[P.td[c]].defined || !P.huff[P.ta[c]].defined) return fail(f, "undefined Huffman table");
    }

    int order[6], nb = 0;                                      // calc_mcu_block_order :3076-3088
    if (f->comps == 1) order[nb++] = 0;
    else for (int c = 0; c < 3; ++c) for (int i = 0; i < P.hs[c] * P.vs[c]; ++i) order[nb++] = c;

    const size_t nmcu = (size_t)f->mcus_per_row * f->mcus_per_col, nblk = nmcu * nb;
    f->coeffs  = (int16_t*)calloc(nblk * 64, sizeof(int16_t));
    f->max_zag = (uint8_t*)malloc(nblk ? nblk : 1);
    if (!f->coeffs || !f->max_zag) { fail(f, "out of memory"); return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: out of memory"); }

    BitReader br(scan, data + len);
    uint32_t pred[3] = { 0, 0, 0 };
    int until_restart = P.restart_interval, expect_rst = 0;
    int16_t* blk = f->coeffs; uint8_t* mz = f->max_zag;
    for (size_t mcu = 0; mcu < nmcu; ++mcu) {
        if (P.restart_interval && until_restart == 0) {        // process_restart :2335-2402
            const uint8_t* q = br.p;
            while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
            if (q + 1 >= br.end || q[1] != 0xD0 + expect_rst) return fail(f, "bad restart marker");
            br.restart(q + 2);
            pred[0] = pred[1] = pred[2] = 0;
            until_restart = P.restart_interval; expect_rst = (expect_rst + 1) & 7;
        }
        for (int b = 0; b < nb; ++b, blk += 64, ++mz) {
            const int c = order[b];
            const int16_t* q = P.quant[P.tq[c]];
            const HuffTable& dc = P.huff[P.td[c]]; const HuffTable& ac = P.huff[P.ta[c]];
            int s = br.decode(dc);
            if (s < 0) return fail(f, "bad Huffman code");
            int v = br.receive_extend(s & 15);
            pred[c] = (uint32_t)(v += (int)pred[c]);
            blk[0] = (int16_t)((uint32_t)v * (uint32_t)(int32_t)q[0]);
            int kk = 1;
            for (; kk < 64; ++kk) {
                const int rs = br.decode(ac);
                if (rs < 0) return fail(f, "bad Huffman code");
                const int run = rs >> 4, size = rs & 15;
                if (size) {
                    if (run) { if (kk + run > 63) return fail(f, "decode error"); kk += run; }
                    const int e = br.receive_extend(size);
                    blk[kZag[kk]] = (int16_t)((uint32_t)e * (uint32_t)(int32_t)q[kk]);
                } else if (run == 15) {
                    if (kk + 16 > 64) return fail(f, "decode error");
                    kk += 15;
                } else break;
            }
            *mz = (uint8_t)kk;                                   // m_mcu_block_max_zag :2512
        }
        --until_restart;
    }
    return GAMUT_HIP_OK;
}

} // namespace
} // namespace gamut

using namespace gamut;

extern "C" {

int gamut_hip_jpeg_decode_coeffs(const uint8_t* data, size_t len, gamut_hip_jpeg_frame* out)
{
    clear_error();
    if (!out) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_decode_coeffs: null frame");
    return decode_coeffs(data, len, out);
}

void gamut_hip_jpeg_frame_free(gamut_hip_jpeg_frame* f)
{
    if (!f) return;
    free(f->coeffs); free(f->max_zag);
    f->coeffs = nullptr; f->max_zag = nullptr;
}

int gamut_hip_jpeg_reconstruct_batch_device(const int16_t* coeffs, int64_t coeff_stride,
                                            const uint8_t* max_zag, int64_t zag_stride,
                                            uint8_t* out, int64_t out_pitch, int64_t out_stride,
                                            int width, int height, int scan_type, int out_comps,
                                            int count, void* stream)
{
    clear_error();
    return jpeg_reconstruct_launch(coeffs, coeff_stride, max_zag, zag_stride, out, out_pitch, out_stride,
                                   width, height, scan_type, out_comps, count, pick_stream(stream));
}

int gamut_hip_jpeg_reconstruct_device(const gamut_hip_jpeg_desc* descs, int count, void* stream)
{
    clear_error();
    if (count < 0 || (count > 0 && !descs)) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_reconstruct: bad descriptor array");
    hipStream_t st = pick_stream(stream);
    for (int i = 0; i < count; ++i) {
        const gamut_hip_jpeg_desc& d = descs[i];
        if (int rc = jpeg_reconstruct_launch(d.coeffs, 0, d.max_zag, 0, d.out, d.out_pitch, 0,
                                             d.width, d.height, d.scan_type, d.out_comps, 1, st))
            return rc;
    }
    return GAMUT_HIP_OK;
}

// decompress_jpeg_image_from_stream (jpegload.d:3720-3808) on a memory buffer
uint8_t* gamut_hip_decompress_jpeg_image_from_memory(const uint8_t* data, size_t len,
        int* width, int* height, int* actual_comps, float* pixelAspectRatio, float* dotsPerInchY, int req_comps)
{
    clear_error();
    if (req_comps != -1 && req_comps != 1 && req_comps != 3 && req_comps != 4) {          // :3727
        set_error(GAMUT_HIP_ERR_INVALID_ARG, "decompress_jpeg: req_comps must be -1, 1, 3 or 4");
        return nullptr;
    }
    gamut_hip_jpeg_frame f;
    if (decode_coeffs(data, len, &f)) return nullptr;
    if (width) *width = f.width;
    if (height) *height = f.height;
    if (actual_comps) *actual_comps = f.comps;
    if (pixelAspectRatio) *pixelAspectRatio = -1;
    if (dotsPerInchY) *dotsPerInchY = -1;
    if (req_comps < 0) req_comps = f.comps;

    const size_t nblk = (size_t)f.mcus_per_row * f.mcus_per_col * f.blocks_per_mcu;
    const size_t dst_bpl = (size_t)f.width * req_comps, out_bytes = dst_bpl * f.height;
    uint8_t* result = (uint8_t*)malloc(out_bytes ? out_bytes : 1);                          // :3749, free()-compatible
    void *dco = nullptr, *dzz = nullptr, *dout = nullptr;
    hipStream_t st = thread_stream();
    bool ok = result != nullptr;
    if (!ok) set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "decompress_jpeg: out of memory");
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess) { set_error(GAMUT_HIP_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e)); return false; }
        return true;
    };
    ok = ok && hip_ok(hipMalloc(&dco, nblk * 128), "hipMalloc") && hip_ok(hipMalloc(&dzz, nblk ? nblk : 1), "hipMalloc") &&
         hip_ok(hipMalloc(&dout, out_bytes ? out_bytes : 1), "hipMalloc");
    ok = ok && hip_ok(hipMemcpyAsync(dco, f.coeffs, nblk * 128, hipMemcpyHostToDevice, st), "hipMemcpyAsync") &&
         hip_ok(hipMemcpyAsync(dzz, f.max_zag, nblk, hipMemcpyHostToDevice, st), "hipMemcpyAsync");
    ok = ok && jpeg_reconstruct_launch((const int16_t*)dco, 0, (const uint8_t*)dzz, 0, (uint8_t*)dout, (int64_t)dst_bpl, 0,
                                       f.width, f.height, f.scan_type, req_comps, 1, st) == GAMUT_HIP_OK;
    ok = ok && hip_ok(hipMemcpyAsync(result, dout, out_bytes, hipMemcpyDeviceToHost, st), "hipMemcpyAsync") &&
         hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize");
    if (dco) (void)hipFree(dco);
    if (dzz) (void)hipFree(dzz);
    if (dout) (void)hipFree(dout);
    if (ok) {
        if (pixelAspectRatio) *pixelAspectRatio = f.pixel_aspect_ratio;                     // :3804-3805
        if (dotsPerInchY) *dotsPerInchY = f.dpi_y;
    } else { free(result); result = nullptr; }
    gamut_hip_jpeg_frame_free(&f);
    return result;
}

} // extern "C"
