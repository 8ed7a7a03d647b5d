// jpeg_host.hip -- host side of the JPEG path: the baseline entropy-decode feeder that
// produces the dense coefficient form the kernels consume, and the C-ABI entry points.
//
// The feeder does what jpeg_decoder::decode_next_row does on the CPU in the reference
// (jpegload.d:2405-2525: Huffman decode, DC prediction, de-quantise `s * q[k]` into natural
// order through g_ZAG, track m_mcu_block_max_zag), for a whole image at once, after the
// marker pass of :1160-1848 (DQT tables kept in zig-zag order as int16, :1313-1329; DHT;
// SOF0/SOF1 8-bit; DRI; SOS).  It stays on the host: bit-serial work.
// Progressive frames (SOF2, jpegload.d:3296-3664) accumulate their scans into per-component coefficient planes on the
// host and come out in the same dense form, so the kernels do not know the difference.
// Not supported (the call fails like the reference fails on a stream it rejects): arithmetic / lossless /
// hierarchical frames, 12-bit precision, CMYK, multi-scan *baseline* files.
#include "common.hpp"
#include <functional>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace gamut {
namespace {

const uint8_t kZag[64] = { 0,1,8,16,9,2,3,10,17,24,32,25,18,11,4,5,12,19,26,33,40,48,41,34,27,20,13,6,7,14,21,28,35,42,49,56,57,50,43,36,29,22,15,23,30,37,44,51,58,59,52,45,38,31,39,46,53,60,61,54,47,55,62,63 };

// One DHT table.  jpgd does not decode with canonical ranges but with make_huff_table's look-up arrays and bit tree (:2851-2987), whose
// zero entries double as "no such code": a bit pattern no code word begins is NOT an error there, it decodes as symbol 0 (huff_decode
// :746-813).  The canonical form below reproduces that: the code words, left-aligned in 16 bits, tile [0, span) in table order, so a
// window w >= span is exactly "unassigned", and how many bits such a step takes follows from where the tree walk meets its first empty slot:
//   * no code word shares w's first 8 bits (look_up / look_up2 entry 0): the three-argument huff_decode (baseline, :769-813) takes NO bits,
//     the two-argument one (progressive, :746-766) takes code_size[0] = the length of the last code word whose symbol value is 0 (or 0);
//   * longer code words do (a subtree): the walk stops at the first prefix of w, 9 bits or more, under which no code word lies; that many bits.
// The two-argument variant has one more habit: a code word of up to 8 bits takes code_size[symbol] bits -- the length of the LAST code word
// carrying that symbol value (a table may list a value twice) -- not its own length.
struct HuffTable {
    bool    defined = false;
    bool    oversubscribed = false;  // the lengths claim more than the code space: make_huff_table writes behind look_up[] / aliases tree nodes -- no table to match
    uint8_t bits[17] = {};        // number of codes per length
    uint8_t vals[256] = {};
    // fast path: 10-bit window -> (code length << 8) | symbol, 0 when the code is longer (or the window begins no code word)
    uint16_t fast[1024];
    // slow path: canonical ranges
    int32_t  maxcode[18];
    int32_t  delta[17];           // valptr - mincode
    uint32_t span = 0;            // first 16-bit window no code word covers (65536: the code is complete)
    uint8_t  size_of[256];        // code_size[] of huff_tables (:417): per symbol value, the length of the last code word that carries it
    uint8_t  dc_limit = 0;        // largest symbol value in the table (a DC category above 15 indexes s_extend_test[] out of bounds, :819-822)
    bool     twice = false;       // some symbol value is carried by two code words
    bool build()                   // false: the code-length counts over-subscribe the code space (not a prefix code)
    {
        int code = 0, k = 0;
        span = 0; oversubscribed = false;
        for (int len = 1; len <= 16; ++len) span += (uint32_t)bits[len] << (16 - len);
        memset(size_of, 0, sizeof(size_of)); dc_limit = 0; twice = false;
        for (int len = 1; len <= 16; ++len) for (int i = 0; i < bits[len]; ++i, ++k) {
            if (size_of[vals[k]]) twice = true;
            size_of[vals[k]] = (uint8_t)len;
            if (vals[k] > dc_limit) dc_limit = vals[k];
        }
        if (span > 65536u) { oversubscribed = true; return false; }
        code = 0; k = 0;
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
        return true;
    }
    // bits taken by a window no code word covers (w >= span); `two_arg`: the progressive variant of huff_decode
    inline int unassigned_bits(uint32_t w, bool two_arg) const
    {
        if ((w & 0xFF00u) >= span) return two_arg ? size_of[0] : 0;
        int len = 9;
        while (((w >> (16 - len)) << (16 - len)) < span) ++len;
        return len;
    }
};

// MSB-first bit reader over the entropy-coded segment; handles FF00 stuffing and stops
// feeding real data at a marker (subsequent bits read as 1s, like get_octet :683-696).
struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint64_t acc = 0; int nbits = 0; bool at_marker = false;
    const uint8_t* seg; uint64_t used = 0;     // where the reader was (re)started, the bits taken since (raw_position / resync need the reference's read position)
    BitReader(const uint8_t* b, const uint8_t* e) : p(b), end(e), seg(b) {}
    // the entropy-coded data of a scan whose header ends at `pos`.  A header that runs past the end of the file took its last bytes out of the
    // FF D9 FF D9 ... padding; if it stopped between an FF and its D9, the scan's data begins with that 0xD9 (and ends there: FF D9 follows).
    BitReader(const uint8_t* data, size_t len, size_t pos) : p(data + (pos < len ? pos : len)), end(data + len), seg(p)
    {
        if (pos > len && ((pos - len) & 1)) { acc = 0xD9; nbits = 8; }
    }
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
    inline void drop(int n) { nbits -= n; used += (uint64_t)n; }
    inline int receive_extend(int s)          // JPGD_HUFF_EXTEND :816-822
    {
        if (!s) return 0;
        if (nbits < s) refill();
        const int v = (int)peek(s); drop(s);
        return v < (1 << (s - 1)) ? v + (int)(0xFFFFFFFFu << s) + 1 : v;
    }
    // huff_decode: TWO_ARG = the progressive variant (:746-766), otherwise the baseline one (:769-813).  Always a symbol (see HuffTable).
    template <bool TWO_ARG> inline int decode(const HuffTable& h)
    {
        if (nbits < 16) refill();
        const uint16_t e = h.fast[peek(10)];
        if (e) {
            int len = e >> 8;
            if (TWO_ARG && len <= 8) len = h.size_of[e & 0xFF];
            drop(len); return e & 0xFF;
        }
        const uint32_t w = peek(16);
        if (w >= h.span) { drop(h.unassigned_bits(w, TWO_ARG)); return 0; }
        int32_t code = (int32_t)peek(10); int len = 10;
        while (code > h.maxcode[len]) { ++len; code = (int32_t)peek(len); }      // w < span: some code word of 11 .. 16 bits covers it
        drop(len);
        return h.vals[(code + h.delta[len]) & 0xFF];
    }
    void restart(const uint8_t* np) { p = np; acc = 0; nbits = 0; at_marker = false; seg = np; used = 0; }
};

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// One SOS header (read_sos_marker :1466-1540)
struct Scan {
    int ncomp = 0, comp[4] = {};                // frame component indices, in scan order (JPGD_MAX_COMPS_IN_SCAN = 4: a scan may list a component twice)
    int ss = 0, se = 63, ah = 0, al = 0;        // spectral selection, successive approximation (progressive only)
};

// Where the walk over the marker segments stands in the reference's call tree: it decides what SOI / EOI / SOFn / SOS / RSTn mean.
enum class Walk {
    kFirstScan,    // locate_sof_marker (:1911-1941), then the first locate_sos_marker (:1944-1967): SOF0-2 once, then SOS
    kNextScan,     // locate_sos_marker of a later scan of a progressive frame: EOI ends the frame
    kTrailer       // find_eoi (:2826-2848) behind the last MCU row of a sequential frame: process_markers(allow_restarts) up to ANY of SOFn / SOI / EOI / SOS
};

struct Parser {
    const uint8_t* data = nullptr; size_t len = 0, pos = 0;
    int16_t   quant[4][64]; bool quant_def[4] = { false, false, false, false };
    HuffTable huff[8];                 // 0-3 DC, 4-7 AC (index mapping of read_dht_marker :1247)
    int comp_id[4] = {}, hs[4] = {}, vs[4] = {}, tq[4] = {}, td[4] = {}, ta[4] = {};
    int restart_interval = 0;
    bool have_sof = false, progressive = false;
    Scan scan;
    std::vector<uint8_t> spill;       // a segment that runs past the end of the file, completed with the reference's padding
    // The input as jpgd reads it: the file, then FF D9 FF D9 ... without end (get_char :631-652 pads the end of the stream with EOI markers;
    // stuff_char :677-680 puts read-ahead back in order, so the padding is a property of the position, not of the reader's history)
    inline uint8_t at(size_t p) const { return p < len ? data[p] : (((p - len) & 1) ? 0xD9 : 0xFF); }
    // `n` bytes from `from` on, readable `slack` bytes past their end (a table whose counts run over the segment is read before it is refused)
    const uint8_t* bytes(size_t from, size_t n, size_t slack = 0)
    {
        if (from + n + slack <= len) return data + from;
        spill.resize(n + slack);
        for (size_t i = 0; i < n + slack; ++i) spill[i] = at(from + i);
        return spill.data();
    }
};

int fail(gamut_hip_jpeg_frame* f, const char* why)
{
    free(f->coeffs); free(f->max_zag); f->coeffs = nullptr; f->max_zag = nullptr;
    return set_error(GAMUT_HIP_ERR_DECODE, "jpeg: %s", why);
}

// locate_soi_marker :1854-1908: FF D8 at once, or within the first 4096 bytes and then followed by 0xFF; an FF D9 on the way is the end of the
// search (reading past the end yields FF D9).  -> position behind the SOI, or 0.
size_t locate_soi(const Parser& P)
{
    uint32_t last = P.at(0), cur = P.at(1);
    size_t pos = 2;
    if (last == 0xFF && cur == 0xD8) return pos;
    for (uint32_t left = 4096;;) {
        if (--left == 0) return 0;
        last = cur; cur = P.at(pos++);
        if (last == 0xFF) {
            if (cur == 0xD8) break;
            if (cur == 0xD9) return 0;
        }
    }
    return P.at(pos) == 0xFF ? pos : 0;
}

// `case M_APP0+1` of process_markers, :1704-1816: an "Exif\0\0" segment carries a TIFF file; every IFD of its chain is walked for XResolution
// (282), YResolution (283) -- RATIONALs behind a value offset -- and ResolutionUnit (296, in the offset field itself), defaults 72 / 72 / inches;
// inches and centimetres set the density and the aspect ratio, any other unit leaves them alone.  A byte order other than II / MM, a version other
// than 42, an IFD offset behind the segment: JPGD_DECODE_ERROR.  The reference checks nothing else: a read outside the segment (an entry, a value
// offset, a chain that loops) is outside its malloc block or never ends -- refused here.  A segment shorter than the identifier is not EXIF.
// -> false: the file is refused.
bool exif_density(const uint8_t* s, uint32_t n, gamut_hip_jpeg_frame* f)
{
    if (n < 6 || memcmp(s, "Exif\0\0", 6)) return true;
    const uint8_t* tiff = s + 6; const uint64_t room = n - 6;                      // reads are relative to the TIFF header, bounded by the segment
    bool ok = true;
    auto rd = [&](uint64_t at, int k, bool le) -> uint32_t {
        if (at + (uint64_t)k > room) { ok = false; return 0; }
        uint32_t v = 0;
        for (int i = 0; i < k; ++i) v = le ? v | (uint32_t)tiff[at + i] << (8 * i) : (v << 8) | tiff[at + i];
        return v;
    };
    const uint32_t order = rd(0, 2, false);
    if (!ok || (order != 0x4949 && order != 0x4D4D)) return false;
    const bool le = order == 0x4949;
    if (rd(2, 2, le) != 42 || !ok) return false;
    uint32_t offset = rd(4, 4, le);
    double rx = 72, ry = 72; int unit = 2;
    for (uint32_t hops = 0; offset != 0 && ok; ) {
        if (offset > n || ++hops > n) return false;                                // `offset > exifData.length` (:1769); more IFDs than bytes: a loop
        uint64_t at = offset;
        const uint32_t entries = rd(at, 2, le); at += 2;
        for (uint32_t e = 0; e < entries && ok; ++e, at += 12) {
            const uint32_t tag = rd(at, 2, le), value = rd(at + 8, 4, le);
            if (!ok) break;
            if (tag == 282 || tag == 283) {
                const double num = rd(value, 4, le), den = rd((uint64_t)value + 4, 4, le);
                (tag == 282 ? rx : ry) = num / den;
            }
            if (tag == 296) unit = (int)value;
        }
        offset = rd(at, 4, le);
    }
    if (!ok) return false;
    if (unit == 2)      { f->dpi_y = (float)ry;                                   f->pixel_aspect_ratio = (float)(rx / ry); }
    else if (unit == 3) { f->dpi_y = (float)(ry * 100) / 39.37007874f;            f->pixel_aspect_ratio = (float)(rx / ry); }   // convertInchesToMeters, types.d:127
    return true;
}

// init_frame :3130-3195: the sampling modes jpgd decodes
bool frame_layout(const Parser& P, gamut_hip_jpeg_frame* f)
{
    if (f->comps == 1) {
        if (P.hs[0] != 1 || P.vs[0] != 1) return false;
        f->scan_type = GAMUT_JPGD_GRAYSCALE; f->blocks_per_mcu = 1;
    } else if (f->comps == 3) {
        if (P.hs[1] != 1 || P.vs[1] != 1 || P.hs[2] != 1 || P.vs[2] != 1) return false;
        if      (P.hs[0] == 1 && P.vs[0] == 1) { f->scan_type = GAMUT_JPGD_YH1V1; f->blocks_per_mcu = 3; }
        else if (P.hs[0] == 2 && P.vs[0] == 1) { f->scan_type = GAMUT_JPGD_YH2V1; f->blocks_per_mcu = 4; }
        else if (P.hs[0] == 1 && P.vs[0] == 2) { f->scan_type = GAMUT_JPGD_YH1V2; f->blocks_per_mcu = 4; }
        else if (P.hs[0] == 2 && P.vs[0] == 2) { f->scan_type = GAMUT_JPGD_YH2V2; f->blocks_per_mcu = 6; }
        else return false;
    } else return false;
    const int mw = 8 * (f->comps == 3 ? P.hs[0] : 1), mh = 8 * (f->comps == 3 ? P.vs[0] : 1);
    f->mcus_per_row = (f->width + mw - 1) / mw; f->mcus_per_col = (f->height + mh - 1) / mh;
    return true;
}

// Walks the marker segments from P.pos the way process_markers does (:1578-1848) -- tables may be (re)defined anywhere, JFIF / EXIF density is
// taken wherever it stands, and the verdicts are the reference's: RSTn outside a scan, TEM (FF 01) and JPG (FF C8) are errors, DAC is, a segment
// shorter than its length field is, and a segment LONGER than the file is read out of the FF D9 padding.  Returns
//   0xDA  an SOS has been read (P.scan filled, P.pos at the first entropy-coded byte),
//   0xD9  Walk::kNextScan: EOI -- the frame has no further scan;  Walk::kTrailer: the marker that ends find_eoi's search (whichever it is),
//   -1    after fail(): the reference returns null for this file.
int next_scan(Parser& P, gamut_hip_jpeg_frame* f, Walk walk)
{
    size_t& pos = P.pos;
    for (;;) {
        int m;
        do {                                                   // next_marker :1546-1573
            while (P.at(pos) != 0xFF) ++pos;
            while (P.at(pos) == 0xFF) ++pos;
            m = P.at(pos++);
        } while (m == 0);
        switch (m) {                                           // the markers process_markers hands back to its caller (:1588-1605)
        case 0xC0: case 0xC1: case 0xC2:
            if (walk == Walk::kTrailer) return 0xD9;
            if (walk != Walk::kFirstScan || P.have_sof) { fail(f, "unexpected marker (a second frame header)"); return -1; }       // locate_sos_marker :1953-1958
            break;                                             // read_sof_marker below
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            if (walk == Walk::kTrailer) return 0xD9;
            fail(f, m == 0xC9 && !P.have_sof ? "arithmetic coding is not supported" : "lossless / hierarchical / arithmetic frames are not supported"); return -1;   // :1932-1938
        case 0xD8:
            if (walk == Walk::kTrailer) return 0xD9;
            fail(f, "unexpected SOI marker"); return -1;
        case 0xD9:
            if (walk == Walk::kFirstScan && !P.have_sof) { fail(f, "unsupported marker (EOI in front of the frame header)"); return -1; }
            return 0xD9;                                       // (a frame without a scan: the caller's business)
        case 0xDA:
            if (walk == Walk::kTrailer) return 0xD9;
            if (!P.have_sof) { fail(f, "unsupported marker (SOS in front of the frame header)"); return -1; }
            break;
        case 0xD0: case 0xD1: case 0xD2: case 0xD3: case 0xD4: case 0xD5: case 0xD6: case 0xD7:
            if (walk == Walk::kTrailer) continue;              // allow_restarts :1826-1827 (Issue #93)
            // (between two scans of a progressive frame the reference gives up WITHOUT an error code, :3615, and decode() goes on with the
            //  scans so far in the block order of the last one: parts of every row are memory nobody wrote.  Refused like the rest.)
            fail(f, "restart marker outside a scan"); return -1;
        case 0x01: case 0xC8:
            fail(f, "TEM / JPG marker"); return -1;            // :1833-1838
        case 0xCC:
            fail(f, "arithmetic coding is not supported"); return -1;
        default: break;
        }
        // a segment: two bytes of length, which count themselves
        const uint32_t seg = (uint32_t)P.at(pos) << 8 | P.at(pos + 1);
        const uint8_t* s = P.bytes(pos + 2, seg >= 2 ? seg - 2 : 0, 288);
        uint32_t n = seg - 2;                                  // num_left (uint: a length below 2 wraps, and every reader refuses it first)
        switch (m) {
        case 0xDB:                                             // DQT :1272-1343
            if (seg < 2) { fail(f, "bad DQT marker"); return -1; }
            while (n) {
                const int prec = s[0] >> 4, id = s[0] & 15;
                if (id >= 4) { fail(f, "bad DQT table"); return -1; }
                const uint32_t need = prec ? 129 : 65;
                if (n < need) { fail(f, "bad DQT length"); return -1; }       // (the reference reads the 64 entries first: out of the next segment or the padding)
                ++s;
                for (int i = 0; i < 64; ++i) {
                    uint32_t v = *s++; if (prec) v = (v << 8) + *s++;
                    P.quant[id][i] = (int16_t)v;               // stored as `short`, zig-zag order
                }
                n -= need; P.quant_def[id] = true;
            }
            break;
        case 0xC4:                                             // DHT :1177-1269
            if (seg < 2) { fail(f, "bad DHT marker"); return -1; }
            while (n) {
                int cnt = 0;
                for (int i = 1; i <= 16; ++i) cnt += s[i];
                if (cnt > 255) { fail(f, "bad DHT counts"); return -1; }
                if (n < (uint32_t)(17 + cnt)) { fail(f, "bad DHT marker"); return -1; }
                const int idx = (s[0] & 0x0F) + ((s[0] & 0x10) >> 4) * 4;
                if (idx >= 8) { fail(f, "bad DHT index"); return -1; }
                HuffTable& h = P.huff[idx];
                h.bits[0] = 0;
                for (int i = 1; i <= 16; ++i) h.bits[i] = s[i];
                s += 17; n -= 17;
                memset(h.vals, 0, sizeof(h.vals)); memcpy(h.vals, s, (size_t)cnt);
                s += cnt; n -= cnt;
                // The files of a batch mostly carry the same few tables (an encoder's standard set): a host thread keeps the last tables it
                // built, found again by their bits[] / vals[] -- a copy instead of the 1024-entry look-up table's construction (a header walk
                // 16 -> 3 us per file; 1024 files on 16 threads: 1.1 ms of every batch call in front of the first upload)
                {
                    // (32 of them: a progressive file defines a table per scan, the same dozen in every file of an encoder's; on the heap, not in
                    //  the library's thread-local segment, which every thread that touches the library pays for)
                    struct Built { bool used = false; HuffTable t; };
                    constexpr int kKept = 32;
                    static thread_local std::unique_ptr<Built[]> cache; static thread_local unsigned victim = 0;
                    if (!cache) cache.reset(new Built[kKept]);
                    int hit = -1;
                    for (int k = 0; k < kKept && hit < 0; ++k) if (cache[k].used && !memcmp(cache[k].t.bits, h.bits, sizeof(h.bits)) && !memcmp(cache[k].t.vals, h.vals, sizeof(h.vals))) hit = k;
                    if (hit >= 0) h = cache[hit].t;
                    else {
                        // an over-subscribed table is not refused HERE: the reference builds its tables at the start of a scan (check_huff_tables
                        // :3019-3031, every table defined by then), so one that is replaced before, or arrives behind the last scan, does no harm
                        h.build();
                        h.defined = true;
                        Built& b = cache[victim++ % kKept]; b.t = h; b.used = true;
                    }
                }
            }
            break;
        case 0xC0: case 0xC1: case 0xC2: {                     // SOF0 / SOF1 / SOF2: read_sof_marker :1346-1415, then init_frame :3130-3195
            if (s[0] != 8) { fail(f, "only 8-bit precision is supported"); return -1; }
            f->height = be16(s + 1); f->width = be16(s + 3); f->comps = s[5];
            if (f->height < 1 || f->height > 16384) { fail(f, "bad height"); return -1; }
            if (f->width < 1 || f->width > 16384) { fail(f, "bad width"); return -1; }
            if (f->comps > 4) { fail(f, "too many components"); return -1; }
            if (seg != (uint32_t)(f->comps * 3 + 8)) { fail(f, "bad SOF length"); return -1; }
            for (int i = 0; i < f->comps; ++i) { P.comp_id[i] = s[6 + 3 * i]; P.hs[i] = s[7 + 3 * i] >> 4; P.vs[i] = s[7 + 3 * i] & 15; P.tq[i] = s[8 + 3 * i]; }
            if (f->comps != 1 && f->comps != 3) { fail(f, "unsupported colorspace"); return -1; }
            if (!frame_layout(P, f)) { fail(f, "unsupported sampling factors"); return -1; }
            for (int i = 0; i < f->comps; ++i) if (P.tq[i] >= 4) { fail(f, "bad quant table selector"); return -1; }   // m_quant[4 ..]: outside the array
            P.have_sof = true; P.progressive = (m == 0xC2);
        } break;
        case 0xDD:                                             // DRI :1445-1462
            if (seg != 4) { fail(f, "bad DRI length"); return -1; }
            P.restart_interval = be16(s);
            break;
        case 0xE0:                                             // APP0 :1634-1702: the density of a JFIF header of 14 bytes or more
            if (seg < 7) { fail(f, "bad variable marker (APP0)"); return -1; }
            if (seg >= 14 && !memcmp(s, "JFIF\0", 5)) {
                const int units = s[7], xd = be16(s + 8), yd = be16(s + 10);
                f->pixel_aspect_ratio = (float)(xd / (double)yd);
                if (units == 0) f->dpi_y = -1;
                else if (units == 1) f->dpi_y = (float)yd;
                else if (units == 2) f->dpi_y = (yd * 100.0f) / 39.37007874f;
            }
            break;
        case 0xE1:                                             // APP1 :1704-1816
            if (seg < 2) { fail(f, "bad variable marker (APP1)"); return -1; }
            if (!exif_density(s, n, f)) { fail(f, "bad EXIF segment"); return -1; }
            break;
        case 0xDA: {                                           // SOS: read_sos_marker :1466-1543
            Scan& sc = P.scan;
            const int ns = s[0];
            if (seg != (uint32_t)(ns * 2 + 6) || ns < 1 || ns > 4) { fail(f, "bad SOS length"); return -1; }
            sc.ncomp = ns;
            for (int i = 0; i < ns; ++i) {
                int ci = 0; while (ci < f->comps && P.comp_id[ci] != s[1 + 2 * i]) ++ci;
                if (ci >= f->comps) { fail(f, "bad SOS component id"); return -1; }
                sc.comp[i] = ci;
                P.td[ci] = (s[2 + 2 * i] >> 4) & 15; P.ta[ci] = (s[2 + 2 * i] & 15) + 4;     // (a component listed twice keeps its last pair; range: scan_tables_ok)
            }
            const uint8_t* t = s + 1 + 2 * ns;
            sc.ss = t[0]; sc.se = t[1]; sc.ah = t[2] >> 4; sc.al = t[2] & 15;
            if (!P.progressive) { sc.ss = 0; sc.se = 63; }
            pos += (size_t)seg;
            return 0xDA;
        }
        default:                                               // DNL / DHP / EXP / APPn / JPGn / COM / RESn: skip_variable_marker :1418-1442
            if (seg < 2) { fail(f, "bad variable marker"); return -1; }
            break;
        }
        pos += (size_t)seg;
    }
}

// What init_scan checks before a scan's first bit (:3101-3106): check_huff_tables wants a DC table for every listed component when the scan
// starts at coefficient 0 and an AC table when it ends behind it -- whether or not the scan will use them (a DC refinement scan reads raw
// bits) -- then BUILDS every table defined so far; check_quant_tables wants the listed components' quantisation tables.
bool scan_tables_ok(const Parser& P, gamut_hip_jpeg_frame* f)
{
    const Scan& sc = P.scan;
    for (int i = 0; i < sc.ncomp; ++i) {
        const int c = sc.comp[i];
        // m_huff_num / m_pHuff_tabs have 8 slots, DC in 0-3, AC in 4-7: nothing keeps a DC selector of 4-7 from naming an AC table (the scan is decoded
        // with it all the same); a selector above that indexes past the arrays -- where the scan looks at it at all (a DC selector in a scan that starts
        // behind coefficient 0, an AC selector in one that ends at it, is never used)
        if (sc.ss == 0 && (P.td[c] >= 8 || !P.huff[P.td[c]].defined)) { fail(f, P.td[c] >= 8 ? "bad Huffman table selector" : "undefined Huffman table"); return false; }
        if (sc.se > 0 && (P.ta[c] >= 8 || !P.huff[P.ta[c]].defined)) { fail(f, P.ta[c] >= 8 ? "bad Huffman table selector" : "undefined Huffman table"); return false; }
    }
    for (int t = 0; t < 8; ++t) if (P.huff[t].defined && P.huff[t].oversubscribed) { fail(f, "bad DHT counts (not a prefix code)"); return false; }
    for (int i = 0; i < sc.ncomp; ++i) if (!P.quant_def[P.tq[sc.comp[i]]]) { fail(f, "undefined quant table"); return false; }
    return true;
}

// Where the REFERENCE's input stands while the bit reader is `used` bits into a segment: it buffers 16 .. 32 bits, two octets per refill
// (get_bits_no_markers :722-743; four at a (re)start :2111-2116 / :2394-2400), an octet being a data byte or an FF 00 pair, and it never steps
// over a marker (get_octet :683-696 puts it back) -- so it is 4 + 2 * (used / 16) octets behind the (re)start, or at the marker that stopped
// it.  process_restart, find_eoi and the search for a progressive frame's next scan all continue from THERE, not from the last bit decoded.
const uint8_t* raw_position(const BitReader& br)
{
    const uint8_t* q = br.seg;
    for (uint64_t n = 4 + 2 * (br.used / 16); n > 0 && q < br.end; --n) {
        if (*q == 0xFF) { if (q + 1 < br.end && q[1] == 0x00) q += 2; else break; }
        else ++q;
    }
    return q;
}

// process_restart :2335-2402: raw bytes from raw_position() on -- up to 1536 of them to the next 0xFF, its fill bytes, and then the expected
// RSTn or JPGD_BAD_RESTART_MARKER (a stray marker, stuffed data left over in front of the marker, more than 1536 bytes of it).
bool resync(BitReader& br, int& expect_rst)
{
    const uint8_t* q = raw_position(br);
    int tem = 0, i, c = 0;
    auto get_char = [&]() -> int { return q < br.end ? (int)*q++ : ((tem ^= 1) ? 0xFF : 0xD9); };      // :631-652: FF D9 FF D9 ... past the end
    for (i = 1536; i > 0; --i) if (get_char() == 0xFF) break;
    if (i == 0) return false;
    for (; i > 0; --i) { c = get_char(); if (c != 0xFF) break; }
    if (i == 0 || c != 0xD0 + expect_rst) return false;
    br.restart(q);
    expect_rst = (expect_rst + 1) & 7;
    return true;
}

// calc_mcu_block_order :3068-3088: the blocks of an MCU belong to the components IN THE ORDER THE SOS LISTS THEM (m_comp_list), hs x vs blocks each
// (one block for a single-component scan).  A conforming file lists them in frame order; the reference does not check, and decodes block b
// with the tables / the predictor of order[b] while everything behind the entropy decoder goes by the block's position.  -> blocks per MCU.
inline int scan_block_order(const Parser& P, int* order, int cap)
{
    const Scan& sc = P.scan;
    int nb = 0;
    if (sc.ncomp == 1) { if (cap > 0) order[0] = sc.comp[0]; return 1; }
    for (int i = 0; i < sc.ncomp; ++i) for (int k = 0; k < P.hs[sc.comp[i]] * P.vs[sc.comp[i]]; ++k) { if (nb < cap) order[nb] = sc.comp[i]; ++nb; }
    return nb;
}
// A sequential frame whose scan lists a component twice has another number of blocks per MCU than init_frame (:3136-3260) sized
// m_pMCU_coefficients / the sample buffer for: more overruns them, fewer leaves the rest of every MCU as the allocator left it.  No result to match.
inline bool baseline_order_ok(const Parser& P, const gamut_hip_jpeg_frame* f, int* order)
{
    return scan_block_order(P, order, 6) == f->blocks_per_mcu;
}
// decode_scan :3520-3583 steps block_x_mcu / m_block_y_mcu of a component once per time the interleaved scan lists it: listed twice, the walk
// leaves the component's plane inside the first MCU row (coeff_buf_getp's assert :3293).
inline bool scan_lists_a_component_twice(const Scan& sc)
{
    for (int a = 0; a < sc.ncomp; ++a) for (int b = a + 1; b < sc.ncomp; ++b) if (sc.comp[a] == sc.comp[b]) return true;
    return false;
}

// ---- sequential frames: one interleaved scan (decode_next_row :2405-2525) --------------------------------------
int decode_baseline(Parser& P, gamut_hip_jpeg_frame* f, const int* order, int nb, size_t nmcu)
{
    if (P.scan.ncomp != f->comps) return fail(f, "only single-scan baseline files are supported");
    if (!scan_tables_ok(P, f)) return GAMUT_HIP_ERR_DECODE;
    BitReader br(P.data, P.len, P.pos);
    uint32_t pred[3] = { 0, 0, 0 };
    int until_restart = P.restart_interval, expect_rst = 0;
    int16_t* blk = f->coeffs; uint8_t* mz = f->max_zag;
    for (size_t mcu = 0; mcu < nmcu; ++mcu) {
        if (P.restart_interval && until_restart == 0) {
            if (!resync(br, expect_rst)) return fail(f, "bad restart marker");
            pred[0] = pred[1] = pred[2] = 0;
            until_restart = P.restart_interval;
        }
        for (int b = 0; b < nb; ++b, blk += 64, ++mz) {
            const int c = order[b];
            const int16_t* q = P.quant[P.tq[c]];
            const HuffTable& dc = P.huff[P.td[c]]; const HuffTable& ac = P.huff[P.ta[c]];
            const int s = br.decode<false>(dc);
            if (s > 15) return fail(f, "a DC category above 15");   // the extra bits are s & 15, the sign test is s_extend_test[s] (:816-822): outside the table
            int v = br.receive_extend(s);
            pred[c] = (uint32_t)(v += (int)pred[c]);
            blk[0] = (int16_t)((uint32_t)v * (uint32_t)(int32_t)q[0]);
            int kk = 1;
            for (; kk < 64; ++kk) {
                const int rs = br.decode<false>(ac);
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
    // find_eoi :2826-2848, called with the last MCU row (:564-569): the markers behind the scan are processed like those in front of it --
    // a table segment that is malformed, a TEM / JPG marker or a bad EXIF segment there makes decode() fail, and a JFIF / EXIF segment there
    // still sets the density the caller gets
    P.pos = (size_t)(raw_position(br) - P.data);
    if (next_scan(P, f, Walk::kTrailer) < 0) return GAMUT_HIP_ERR_DECODE;
    return GAMUT_HIP_OK;
}

// ---- progressive frames (init_progressive :3585-3664) ---------------------------------------------------------
// Every component owns a plane of 64-coefficient blocks (natural order, not yet de-quantised) covering the padded
// MCU grid; DC and AC share a block (the reference's separate 1x1 DC buffer is merged back in load_next_row
// :2280-2284, and AC scans never touch coefficient 0).  Scans accumulate into the planes; afterwards the planes are
// gathered into the same MCU-ordered dense form the baseline path produces.
struct Plane { int16_t* blk = nullptr; int bw = 0, bh = 0; int16_t* at(int bx, int by) const { return blk + ((size_t)by * bw + bx) * 64; } };

struct Progressive {
    Parser& P; gamut_hip_jpeg_frame* f;
    Plane plane[3];
    BitReader br; uint32_t pred[3] = { 0, 0, 0 }; int eobrun = 0;
    Progressive(Parser& p, gamut_hip_jpeg_frame* fr) : P(p), f(fr), br(nullptr, nullptr) {}
    ~Progressive() { for (Plane& pl : plane) free(pl.blk); }

    inline int bit() { if (br.nbits < 1) br.refill(); const int v = (int)br.peek(1); br.drop(1); return v; }
    inline int bits(int n) { if (!n) return 0; if (br.nbits < n) br.refill(); const int v = (int)br.peek(n); br.drop(n); return v; }

    bool dc_first(int c, int16_t* b)                     // decode_block_dc_first :3298-3319
    {
        const int s = br.decode<true>(P.huff[P.td[c]]);
        if (s > 15) return false;                        // get_bits_no_markers(s) / s_extend_test[s] with a category above 15
        const int v = br.receive_extend(s) + (int)pred[c];
        pred[c] = (uint32_t)v;
        b[0] = (int16_t)((uint32_t)v << P.scan.al);
        return true;
    }
    bool dc_refine(int, int16_t* b)                      // decode_block_dc_refine :3321-3333
    {
        if (bit()) b[0] = (int16_t)(b[0] | (1 << P.scan.al));
        return true;
    }
    bool ac_first(int c, int16_t* b)                     // decode_block_ac_first :3335-3398
    {
        if (eobrun) { --eobrun; return true; }
        const HuffTable& ac = P.huff[P.ta[c]];
        for (int k = P.scan.ss; k <= P.scan.se; ++k) {
            const int rs = br.decode<true>(ac);
            const int run = rs >> 4, size = rs & 15;
            if (size) {
                if ((k += run) > 63) return false;
                b[kZag[k]] = (int16_t)((uint32_t)br.receive_extend(size) << P.scan.al);
            } else if (run == 15) {
                if ((k += 15) > 63) return false;
            } else {                                     // EOBn: this block and the next eobrun blocks end here
                eobrun = (1 << run) + bits(run) - 1;
                break;
            }
        }
        return true;
    }
    inline void correct(int16_t& coef, int plus, int minus)      // one correction bit for a coefficient with history
    {
        if (bit() && (coef & plus) == 0) coef = (int16_t)(coef + (coef >= 0 ? plus : minus));
    }
    bool ac_refine(int c, int16_t* b)                    // decode_block_ac_refine :3400-3518
    {
        const int plus = 1 << P.scan.al, minus = (int)(0xFFFFFFFFu << P.scan.al);
        const HuffTable& ac = P.huff[P.ta[c]];
        int k = P.scan.ss;
        if (eobrun == 0) {
            while (k <= P.scan.se) {
                const int rs = br.decode<true>(ac);
                int run = rs >> 4; const int size = rs & 15;
                int fresh = 0;                           // value of the newly non-zero coefficient, if any
                if (size) {
                    if (size != 1) return false;
                    fresh = bit() ? plus : minus;
                } else if (run != 15) {
                    eobrun = (1 << run) + bits(run);
                    break;
                }
                // skip `run` coefficients that are still zero, correcting every non-zero one passed on the way
                for (; k <= P.scan.se; ++k) {
                    int16_t& coef = b[kZag[k]];
                    if (coef) correct(coef, plus, minus);
                    else { if (run == 0) break; --run; }
                }
                if (fresh && k < 64) b[kZag[k]] = (int16_t)fresh;
                ++k;
            }
        }
        if (eobrun > 0) {
            for (; k <= P.scan.se; ++k) { int16_t& coef = b[kZag[k]]; if (coef) correct(coef, plus, minus); }
            --eobrun;
        }
        return true;
    }

    // decode_scan :3521-3582.  Interleaved scans walk the frame's MCU grid; a single-component scan walks that
    // component's own blocks (ceil of its sampled size, calc_mcu_block_order :3052-3066), row by row.
    template <class Fn> bool run_scan(Fn fn)
    {
        const Scan& sc = P.scan;
        int until_restart = P.restart_interval, expect_rst = 0;
        auto boundary = [&]() -> bool {
            if (P.restart_interval && until_restart == 0) {
                if (!resync(br, expect_rst)) return false;
                pred[0] = pred[1] = pred[2] = 0; eobrun = 0;
                until_restart = P.restart_interval;
            }
            return true;
        };
        if (sc.ncomp == 1) {
            const int c = sc.comp[0];
            int max_h = 1, max_v = 1;
            for (int i = 0; i < f->comps; ++i) { if (P.hs[i] > max_h) max_h = P.hs[i]; if (P.vs[i] > max_v) max_v = P.vs[i]; }
            const int nbx = ((f->width  * P.hs[c] + max_h - 1) / max_h + 7) / 8;
            const int nby = ((f->height * P.vs[c] + max_v - 1) / max_v + 7) / 8;
            if (nbx > plane[c].bw || nby > plane[c].bh) return false;
            for (int by = 0; by < nby; ++by)
                for (int bx = 0; bx < nbx; ++bx) {
                    if (!boundary()) return false;
                    if (!fn(c, plane[c].at(bx, by))) return false;
                    --until_restart;
                }
        } else {
            for (int my = 0; my < f->mcus_per_col; ++my)
                for (int mx = 0; mx < f->mcus_per_row; ++mx) {
                    if (!boundary()) return false;
                    for (int i = 0; i < sc.ncomp; ++i) {
                        const int c = sc.comp[i];
                        for (int v = 0; v < P.vs[c]; ++v)
                            for (int h = 0; h < P.hs[c]; ++h)
                                if (!fn(c, plane[c].at(mx * P.hs[c] + h, my * P.vs[c] + v))) return false;
                    }
                    --until_restart;
                }
        }
        return true;
    }

    // `marker`: what the walk behind the frame header found -- 0xDA with P.scan read, or 0xD9: a frame without a scan (init_progressive's loop
    // ends at once and load_next_row hands out the cleared coefficient buffers: a grey image, not an error)
    int run(const int* order, int nb, int marker)
    {
        for (int c = 0; c < f->comps; ++c) {
            plane[c].bw = f->mcus_per_row * P.hs[c]; plane[c].bh = f->mcus_per_col * P.vs[c];
            plane[c].blk = (int16_t*)calloc((size_t)plane[c].bw * plane[c].bh * 64, sizeof(int16_t));
            if (!plane[c].blk) { fail(f, "out of memory"); return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: out of memory"); }
        }
        while (marker == 0xDA) {
            const Scan& sc = P.scan;
            const bool dc_scan = sc.ss == 0, refine = sc.ah != 0;
            if (scan_lists_a_component_twice(sc)) return fail(f, "the scan lists a component twice");
            if (!scan_tables_ok(P, f)) return GAMUT_HIP_ERR_DECODE;
            if (sc.ss > sc.se || sc.se > 63 || (dc_scan && sc.se != 0)) return fail(f, "bad SOS spectral selection");
            if (!dc_scan && sc.ncomp != 1) return fail(f, "AC scans can only contain one component");
            if (refine && sc.al != sc.ah - 1) return fail(f, "bad SOS successive approximation");
            br = BitReader(P.data, P.len, P.pos);
            pred[0] = pred[1] = pred[2] = 0; eobrun = 0;
            bool ok;
            if (dc_scan) ok = refine ? run_scan([&](int c, int16_t* b) { return dc_refine(c, b); }) : run_scan([&](int c, int16_t* b) { return dc_first(c, b); });
            else         ok = refine ? run_scan([&](int c, int16_t* b) { return ac_refine(c, b); }) : run_scan([&](int c, int16_t* b) { return ac_first(c, b); });
            if (!ok) return fail(f, "decode error in a progressive scan");
            P.pos = (size_t)(raw_position(br) - P.data);       // :3667-3673: the bit buffer is thrown away, the search for the next SOS starts where the input stands
            marker = next_scan(P, f, Walk::kNextScan);
            if (marker < 0) return GAMUT_HIP_ERR_DECODE;
        }
        // load_next_row :2259-2333: per block, last non-zero coefficient in zig-zag order + 1, then de-quantise -- with the table of EVERY component
        // of the frame, which no scan may have asked for (check_quant_tables :2990-3000 looks at the components a scan lists): a null pointer there
        for (int c = 0; c < f->comps; ++c) if (!P.quant_def[P.tq[c]]) return fail(f, "undefined quant table");
        int16_t* dst = f->coeffs; uint8_t* mz = f->max_zag;
        for (int my = 0; my < f->mcus_per_col; ++my)
            for (int mx = 0; mx < f->mcus_per_row; ++mx)
                for (int b = 0, within = 0; b < nb; ++b, dst += 64, ++mz) {
                    const int c = order[b];
                    within = (b > 0 && order[b - 1] == c) ? within + 1 : 0;
                    const int16_t* src = plane[c].at(mx * P.hs[c] + within % P.hs[c], my * P.vs[c] + within / P.hs[c]);
                    const int16_t* q = P.quant[P.tq[c]];
                    int last = 63;
                    while (last > 0 && src[kZag[last]] == 0) --last;
                    *mz = (uint8_t)(last + 1);
                    for (int k = 0; k <= last; ++k) {
                        const int16_t v = src[kZag[k]];
                        if (v) dst[kZag[k]] = (int16_t)((uint32_t)(int32_t)v * (uint32_t)(int32_t)q[k]);
                    }
                }
        return GAMUT_HIP_OK;
    }
};

// decode_init (:3708-3713): the SOI, the markers up to the frame header, the frame header; then what decode_start (:3697-3706) reads up to
// the first scan.  -> 0xDA / 0xD9 (a progressive frame without a scan) / -1 after fail().
// pixelAspectRatio / dotsPerInchY start as NaN: they are `float` members of the D struct, which initit (:1971-2078) never assigns -- a file
// without JFIF / EXIF density hands NaN to its caller, not the -1 the comment at :3719 promises.
int open_frame(Parser& P, const uint8_t* data, size_t len, gamut_hip_jpeg_frame* f)
{
    memset(f, 0, sizeof(*f));
    f->pixel_aspect_ratio = NAN; f->dpi_y = NAN;
    P.data = data; P.len = data ? len : 0;
    P.pos = locate_soi(P);
    if (!P.pos) { fail(f, "not a JPEG (no SOI)"); return -1; }
    const int first = next_scan(P, f, Walk::kFirstScan);
    if (first == 0xD9 && !P.progressive) { fail(f, "no SOS marker"); return -1; }
    return first;
}

int decode_coeffs(const uint8_t* data, size_t len, gamut_hip_jpeg_frame* f)
{
    Parser* ps = new (std::nothrow) Parser();
    if (!ps) { memset(f, 0, sizeof(*f)); return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: out of memory"); }
    struct Guard { Parser* p; ~Guard() { delete p; } } guard{ ps };
    Parser& P = *ps;

    const int first = open_frame(P, data, len, f);
    if (first < 0) return GAMUT_HIP_ERR_DECODE;

    int order[6], nb = 0;                                      // calc_mcu_block_order :3076-3088 (frame-interleaved)
    if (f->comps == 1) order[nb++] = 0;
    else for (int c = 0; c < 3; ++c) for (int i = 0; i < P.hs[c] * P.vs[c]; ++i) order[nb++] = c;

    const size_t nmcu = (size_t)f->mcus_per_row * f->mcus_per_col, nblk = nmcu * nb;
    f->coeffs  = (int16_t*)calloc(nblk * 64, sizeof(int16_t));
    f->max_zag = (uint8_t*)malloc(nblk ? nblk : 1);
    if (!f->coeffs || !f->max_zag) { fail(f, "out of memory"); return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: out of memory"); }

    if (!P.progressive) {
        int scan_order[6] = { 0, 0, 0, 0, 0, 0 };              // tables and predictors by the scan's list, positions by the frame's (scan_block_order)
        if (P.scan.ncomp == f->comps && !baseline_order_ok(P, f, scan_order)) return fail(f, "the scan lists a component twice");
        return decode_baseline(P, f, scan_order, nb, nmcu);
    }
    Progressive* pg = new (std::nothrow) Progressive(P, f);
    if (!pg) { fail(f, "out of memory"); return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: out of memory"); }
    const int rc = pg->run(order, nb, first);
    delete pg;
    return rc;
}


// =====================================================================================================================
// Entropy decode ON THE GPU: baseline scans only.
//
// Huffman decoding is serial inside a stream, but a batch holds many independent streams: every image, and inside an
// image every restart interval (process_restart :2335-2402 resets the DC predictors and byte-aligns the stream), can be
// decoded by its own lane.  The host only walks the markers (tables, geometry) and locates the RSTn boundaries; the
// compressed bytes go to the device as they are (230 kB instead of 6.3 MB of coefficients per 1080p image over PCIe) and
// k_jpeg_entropy writes the same dense de-quantised coefficient form decode_next_row (:2405-2525) produces, straight
// into HBM, ready for k_jpeg_h2v2 / k_jpeg_generic.  Same arithmetic as decode_baseline() above, which is its oracle.
#ifndef JPEG_HUFF_SUB_ENTRIES      // (tools/variant.sh knob; the standard tables need 144 / 150 entries)
#define JPEG_HUFF_SUB_ENTRIES 312      // sizeof(DevHuff) = 2048: a table's address is a shift of its number
#endif
struct DevHuff {                       // one Huffman table (shared by every image that uses the same table)
    uint16_t fast[512];                // 9-bit lookahead -> (length << 8) | symbol, 0 = longer code
    int32_t  maxcode[18];
    int32_t  delta[17];
    uint8_t  vals[256];
    uint8_t  pad[4];
    // codes of more than 9 bits, second level: fast[p] of a 9-bit prefix p that only longer codes begin with = 0x8000 | x << 12 | offset: the next
    // x bits (x = the longest such code's length - 9, 1 .. 7) index sub[offset ..] -> (length << 8) | symbol, 0 = no such code.  A table whose
    // second levels do not fit in sub[] keeps fast[p] = 0 for the prefixes that did not (and the canonical search over maxcode[] finds the code,
    // as it did for every long code before round 4: up to eight dependent LDS reads, and with 64 lanes per wave SOME lane has a long code in most steps).
    uint16_t sub[JPEG_HUFF_SUB_ENTRIES];
};
constexpr uint32_t kHuffLong = 0x8000u;
static_assert(JPEG_HUFF_SUB_ENTRIES != 312 || sizeof(DevHuff) == 2048, "DevHuff: a power of two");
inline void to_dev_huff_build(const HuffTable& h, DevHuff& d);
inline void to_dev_huff(const HuffTable& h, DevHuff& d)      // (the same few tables file after file: the last ones a thread converted are kept, as in the DHT walk)
{
    struct Made { bool used = false; uint8_t bits[17]; uint8_t vals[256]; DevHuff d; };
    constexpr int kKept = 32;
    static thread_local std::unique_ptr<Made[]> cache; static thread_local unsigned victim = 0;
    if (!cache) cache.reset(new Made[kKept]);
    for (int k = 0; k < kKept; ++k) if (cache[k].used && !memcmp(cache[k].bits, h.bits, sizeof(h.bits)) && !memcmp(cache[k].vals, h.vals, sizeof(h.vals))) { d = cache[k].d; return; }
    to_dev_huff_build(h, d);
    Made& m = cache[victim++ % kKept];
    memcpy(m.bits, h.bits, sizeof(m.bits)); memcpy(m.vals, h.vals, sizeof(m.vals)); m.d = d; m.used = true;
}
inline void to_dev_huff_build(const HuffTable& h, DevHuff& d)
{
    memset(&d, 0, sizeof(d));
    for (int w = 0; w < 512; ++w) { const uint16_t e = h.fast[w << 1]; d.fast[w] = (e >> 8) <= 9 ? e : 0; }   // 10-bit table -> 9-bit
    memcpy(d.maxcode, h.maxcode, sizeof(d.maxcode)); memcpy(d.delta, h.delta, sizeof(d.delta)); memcpy(d.vals, h.vals, sizeof(d.vals));
    // second level: the longest code under every 9-bit prefix, then the codes themselves (canonical order: bits[] / vals[])
    uint8_t longest[512] = { 0 };
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
        for (int i = 0; i < h.bits[len]; ++i, ++code) if (len > 9) { uint8_t& m = longest[(code >> (len - 9)) & 511]; if (len > m) m = (uint8_t)len; }
        code <<= 1;
    }
    int used = 0;
    uint16_t off[512];
    for (int p = 0; p < 512; ++p) {
        off[p] = 0xFFFF;
        if (!longest[p] || d.fast[p]) continue;                  // (a prefix shorter codes cover cannot begin a longer one: the table is a prefix code)
        const int n = 1 << (longest[p] - 9);
        if (used + n > (int)(sizeof(d.sub) / sizeof(d.sub[0]))) continue;
        off[p] = (uint16_t)used; used += n;
        d.fast[p] = (uint16_t)(kHuffLong | (uint32_t)(longest[p] - 9) << 12 | off[p]);
    }
    code = 0; k = 0;
    for (int len = 1; len <= 16; ++len) {
        for (int i = 0; i < h.bits[len]; ++i, ++code, ++k) {
            if (len <= 9) continue;
            const int p = (code >> (len - 9)) & 511;
            if (off[p] == 0xFFFF) continue;
            const int x = longest[p] - 9, have = len - 9;         // the code's bits behind the prefix, left-aligned in the x index bits
            const int first = (code & ((1 << have) - 1)) << (x - have), n = 1 << (x - have);
            for (int j = 0; j < n; ++j) d.sub[off[p] + first + j] = (uint16_t)((len << 8) | h.vals[k]);
        }
        code <<= 1;
    }
}
struct DevImage {
    int64_t coeff_off, zag_off;        // int16 elements / bytes from the start of the caller's buffers
    int32_t nb, org;                   // blocks per MCU; component of block b = (org >> 2 * b) & 3: the order the SOS lists them in (scan_block_order)
    int32_t quant[3], dc[3], ac[3];    // table indices per component
    int32_t tok;                       // 1: the coefficients leave as a token stream (compact hand-off, below) instead of dense 128-byte blocks
    int64_t tok_off, strip_off;        // the image's tokens / strip table: elements from the start of the token buffer / the strip-start buffer
    int32_t sb, row_blocks;            // blocks per strip of the reconstruction kernel (k_jpeg_h2v2: 8 MCUs = 48), blocks per MCU row
    int32_t n_strips, tok_cap;         // strips of the image (its table has n_strips + 1 entries), tokens its slot holds
};
constexpr uint32_t kStatusBadRestart = 8u;
struct DevItem { int32_t image, first_mcu, n_mcus, pad; uint64_t begin, end; int32_t tail, pad2; };    // one restart interval (or whole scan)
// tail: -1 = nothing follows the segment but the end of the scan; >= 0 = an RSTn marker follows it, behind `tail` 0xFF fill bytes (restart_leftover_bad)

// process_restart (jpegload.d:2335-2402) does not look for the marker where the interval's data ENDS but from where the decoder's input
// stands: the bit reader holds 16 .. 32 bits, fetched two octets at a time (four at a (re)start) and never past a marker, so after
// `used_bits` bits of an interval the input is 4 + 2 * (used_bits / 16) octets behind the interval's start.  From there at most 1536 raw
// bytes are read up to a 0xFF, then its fill bytes, and what follows has to be the expected RSTn.  For an intact file that position IS the
// marker (an encoder pads the last byte and nothing more).  A damaged one may leave whole octets between the two: they are skipped if none
// of them is 0xFF (a stuffed FF 00 is "FF, then not RSTn": JPGD_BAD_RESTART_MARKER) and if they, the marker's 0xFF and its fill bytes are
// within the 1536 reads.  seg = the UNSTUFFED interval (every 0xFF in it was FF 00), seg_bytes its length, fill = 0xFF bytes between the
// marker's first 0xFF and its code.  (The markers themselves -- the wrong RSTn, a missing one -- are checked where the scan is unstuffed.)
__device__ inline bool restart_leftover_bad(const uint8_t* seg, uint32_t seg_bytes, uint32_t used_bits, int fill)
{
    const uint64_t oct = 4 + 2 * (uint64_t)(used_bits / 16);
    const uint32_t left = oct < seg_bytes ? seg_bytes - (uint32_t)oct : 0u;
    if (left + (uint32_t)fill > 1535u) return true;
    for (uint32_t i = 0; i < left; ++i) if (seg[oct + i] == 0xFF) return true;
    return false;
}

__constant__ uint8_t kZagDev[64] = { 0,1,8,16,9,2,3,10,17,24,32,25,18,11,4,5,12,19,26,33,40,48,41,34,27,20,13,6,7,14,21,28,35,42,49,56,57,50,43,36,29,22,15,23,30,37,44,51,58,59,52,45,38,31,39,46,53,60,61,54,47,55,62,63 };

// MSB-first reader over an UNSTUFFED segment (the host drops the 0x00 after every 0xFF while it copies the scan, and
// appends 64 bytes of 0xFF after each segment: reading past the end yields 1-bits, as get_octet :683-696 does, and an
// all-ones window is no valid Huffman code, so a runaway decode of corrupt data stops within two bytes of the padding).
struct DevBits {
    const uint8_t* next;               // next four bytes to fetch
    uint64_t acc; int nbits; uint32_t pos;     // pos = bit index (from the segment start) of the next unread bit
    uint32_t pre;                      // bytes at `next`, requested one refill ahead: a lane is a serial chain of dependent
                                       // operations, so the global-memory latency has to overlap the symbols in between
    __device__ __forceinline__ void open(const uint8_t* seg, uint32_t bit)
    {
        next = seg + (bit >> 3); acc = 0; nbits = 0; pos = bit;
        __builtin_memcpy(&pre, next, 4);
        refill();
        nbits -= (int)(bit & 7);                              // drop the bits before `bit` in its byte
    }
    __device__ __forceinline__ void refill()
    {
        if (nbits > 32) return;
        acc = (acc << 32) | __builtin_bswap32(pre); nbits += 32; next += 4;
        __builtin_memcpy(&pre, next, 4);
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)(acc >> (nbits - n)) & ((1u << n) - 1); }
    __device__ __forceinline__ void drop(int n) { nbits -= n; pos += (uint32_t)n; }
    __device__ __forceinline__ int decode(const DevHuff* h)
    {
        refill();
        const uint32_t e = h->fast[peek(9)];
        if (e && e < kHuffLong) { drop((int)(e >> 8)); return (int)(e & 0xFF); }
        if (e) {                                                // a longer code: the second level (DevHuff.sub)
            const int xb = (int)(e >> 12) & 7;
            const uint32_t e2 = h->sub[(e & 0xFFFu) + (peek(9 + xb) & ((1u << xb) - 1u))];
            if (e2) { drop((int)(e2 >> 8)); return (int)(e2 & 0xFF); }
        }
        int32_t code = (int32_t)peek(9); int len = 9;
        while (code > h->maxcode[len]) { if (++len > 16) return -1; code = (int32_t)peek(len); }
        drop(len);
        return h->vals[(code + h->delta[len]) & 0xFF];
    }
    __device__ __forceinline__ int receive_extend(int s)       // JPGD_HUFF_EXTEND :816-822
    {
        if (!s) return 0;
        refill();
        const int v = (int)peek(s); drop(s);
        return v < (1 << (s - 1)) ? v + (int)(0xFFFFFFFFu << s) + 1 : v;
    }
};

constexpr size_t kBlobSlack = 1024;                          // bytes a lane may read past the last segment: one block + look-ahead in k_jpeg_entropy; in k_prog_scan the 64-byte limit check + a block's worth of a corrupt AC scan (244 bytes) + the 256-byte window
#ifndef JPEG_HUFF_SUB              // A/B knob (tools/variant.sh jpeg_host:nosub:-DJPEG_HUFF_SUB=0): 0 = no second-level Huffman look-up in sub_decode
#define JPEG_HUFF_SUB 1
#endif
// Lanes per long segment.  A group of 256 files is 256 workgroups -- one per compute unit: with 256 lanes that is ONE wave per SIMD, a chain of
// dependent instructions at 8 clocks each (tools/microbench/chain_latency.hip) over 880 bytes per lane and pass.  512 lanes: half the bytes per
// lane, two waves per SIMD to alternate between; the passes to the fixed point grow by less.  Round 4, one box, rocprofv3: the compact kernel
// 3.49 -> 2.80 ms per group of 256 files (1024 lanes: 2.99), files -> pixels 256 files 5.9-6.7 -> 5.0-5.1 ms; 1024 files (inside the run-to-run
// spread), 4096 files and the mixed call unchanged (there the GPU is full either way) -- profiles/r04_jpeg_sync_lanes.txt.
#ifndef JPEG_SYNC_THREADS          // tuning knob (tools/variant.sh)
#define JPEG_SYNC_THREADS 512
#endif
constexpr int kEntropyThreads = 64;                          // one wave per workgroup: lanes spread over CUs, each with its own L1
constexpr int kLdsHuff = JPEG_SYNC_THREADS >= 1024 ? 4 : 8, kLdsQuant = kLdsHuff;                     // tables a workgroup keeps in LDS (15 KB + 1 KB: two encoders' sets; more distinct tables in a batch are read from global memory)
constexpr int kSyncThreads = JPEG_SYNC_THREADS;                // lanes cooperating on one large segment
constexpr uint32_t kSyncMinBytes = 4096;                       // shorter segments take one lane each

// Tables into LDS (IN_LDS: the batch uses few distinct tables -- the usual case: encoders write the Annex K tables or one
// optimised set per quality -- so a symbol costs LDS latencies instead of L2 ones; the look-up sits on the lane's
// critical path twice per symbol).
template <bool IN_LDS, int NT>
__device__ __forceinline__ void load_tables(DevHuff* sh_huff, int16_t* sh_quant, uint8_t* sh_zag, const DevHuff* huff_g, int n_huff,
                                            const int16_t* quant_g, int n_quant)
{
    if (threadIdx.x < 64) sh_zag[threadIdx.x] = kZagDev[threadIdx.x];
    if constexpr (IN_LDS) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(huff_g); uint32_t* dst = reinterpret_cast<uint32_t*>(sh_huff);
        for (int k = threadIdx.x; k < n_huff * (int)(sizeof(DevHuff) / 4); k += NT) dst[k] = src[k];
        for (int k = threadIdx.x; k < n_quant * 64; k += NT) sh_quant[k] = quant_g[k];
    }
}

// ---- one lane per (short) segment -------------------------------------------------------------------------------
template <bool IN_LDS>
__global__ __launch_bounds__(kEntropyThreads) void k_jpeg_entropy(const DevItem* items, int n_items, const DevImage* images,
                                                                  const DevHuff* huff_g, int n_huff, const int16_t* quant_g /* [t][64], zig-zag order */,
                                                                  int n_quant, const uint8_t* blob, int16_t* coeffs, uint8_t* max_zag, uint32_t* status)
{
    __shared__ DevHuff sh_huff[IN_LDS ? kLdsHuff : 1];
    __shared__ int16_t sh_quant[IN_LDS ? kLdsQuant * 64 : 1];
    __shared__ uint8_t sh_zag[64];
    // The block being decoded lives in LDS (one 128-byte row per lane, 144-byte pitch): the sparse coefficient writes are
    // LDS writes, and a finished block leaves as eight 16-byte stores -- a whole line, zeros included, so the destination
    // needs no clearing and the lane's bit-stream loads never queue behind a trail of 2-byte global stores.
    __shared__ __attribute__((aligned(16))) uint8_t sh_blk[kEntropyThreads * 144];
    load_tables<IN_LDS, kEntropyThreads>(sh_huff, sh_quant, sh_zag, huff_g, n_huff, quant_g, n_quant);
    {
        uint4* z = reinterpret_cast<uint4*>(sh_blk + threadIdx.x * 144);
        #pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    const DevHuff* huff = IN_LDS ? sh_huff : huff_g;
    const int16_t* quant = IN_LDS ? sh_quant : quant_g;

    const int i = blockIdx.x * kEntropyThreads + threadIdx.x;
    if (i >= n_items) return;
    const DevItem it = items[i];
    const DevImage im = images[it.image];
    DevBits br; br.open(blob + it.begin, 0);
    // A lane may run past its segment only by the 64 bytes of 0xFF padding: with the standard (incomplete) tables an
    // all-ones window is no code and the decode stops by itself, but a DHT may define a COMPLETE code (all-ones included),
    // and a truncated scan under a large SOF would then keep "decoding" padding, the next segments and whatever follows the
    // blob.  Checked once per block: a block consumes < 256 bytes, and the blob is allocated with that much slack.
    const uint32_t limit_bit = (uint32_t)(it.end - it.begin) * 8u + 64u * 8u;
    int pred0 = 0, pred1 = 0, pred2 = 0;
    int16_t* out = coeffs + im.coeff_off + (int64_t)it.first_mcu * im.nb * 64;
    uint8_t* mz = max_zag + im.zag_off + (int64_t)it.first_mcu * im.nb;
    int16_t* blk = reinterpret_cast<int16_t*>(sh_blk + threadIdx.x * 144);
    // damaged data: flag the image and clear the blocks of the segment that were not reached (the caller's buffer is not cleared beforehand)
    int16_t* const seg_end = out + (int64_t)it.n_mcus * im.nb * 64;
    auto bail = [&](uint32_t flag) {
        atomicOr(status + it.image, flag);
        for (int16_t* p = out; p < seg_end; p += 64) {
            uint4* dst = reinterpret_cast<uint4*>(p);
            #pragma unroll
            for (int k = 0; k < 8; ++k) dst[k] = make_uint4(0, 0, 0, 0);
        }
    };
    for (int mcu = 0; mcu < it.n_mcus; ++mcu) {
        for (int b = 0; b < im.nb; ++b, out += 64, ++mz) {
            const int c = (int)((uint32_t)im.org >> (2 * b)) & 3;
            const int16_t* q = quant + (c == 0 ? im.quant[0] : c == 1 ? im.quant[1] : im.quant[2]) * 64;
            const DevHuff* dc = huff + (c == 0 ? im.dc[0] : c == 1 ? im.dc[1] : im.dc[2]);
            const DevHuff* ac = huff + (c == 0 ? im.ac[0] : c == 1 ? im.ac[1] : im.ac[2]);
            const int s = br.decode(dc);
            if (s < 0) { bail(1u); return; }
            const int pred = c == 0 ? pred0 : c == 1 ? pred1 : pred2;
            const int v = br.receive_extend(s & 15) + pred;
            if (c == 0) pred0 = v; else if (c == 1) pred1 = v; else pred2 = v;
            blk[0] = (int16_t)((uint32_t)v * (uint32_t)(int32_t)q[0]);
            int kk = 1;
            for (; kk < 64; ++kk) {
                const int rs = br.decode(ac);
                if (rs < 0) { bail(1u); return; }
                const int run = rs >> 4, size = rs & 15;
                if (size) {
                    if (run) { if (kk + run > 63) { bail(2u); return; } kk += run; }
                    const int e = br.receive_extend(size);
                    blk[sh_zag[kk]] = (int16_t)((uint32_t)e * (uint32_t)(int32_t)q[kk]);
                } else if (run == 15) {
                    if (kk + 16 > 64) { bail(2u); return; }
                    kk += 15;
                } else break;
            }
            if (br.pos > limit_bit) { bail(4u); return; }                               // ran off the end of the segment
            *mz = (uint8_t)kk;                                   // m_mcu_block_max_zag :2512
            uint4* src = reinterpret_cast<uint4*>(blk);
            uint4* dst = reinterpret_cast<uint4*>(out);          // 128-byte blocks at 128-byte-aligned offsets (checked by the host)
            #pragma unroll
            for (int k = 0; k < 8; ++k) { dst[k] = src[k]; src[k] = make_uint4(0, 0, 0, 0); }
        }
    }
    if (it.tail >= 0 && restart_leftover_bad(blob + it.begin, (uint32_t)(it.end - it.begin), br.pos, it.tail)) atomicOr(status + it.image, kStatusBadRestart);
}

// ---- many lanes per (long) segment: self-synchronising decode --------------------------------------------------------
// A Huffman stream can only be entered at a codeword boundary in a known state -- here (bit position, block of the MCU,
// zig-zag index) -- which only a decoder that came from the start knows.  But decoders started at the WRONG place fall
// into step with the right one after a few symbols (Huffman codes self-synchronise; cf. Weissenberger & Schmidt's
// parallel JPEG decoding).  A workgroup cuts its segment into one sub-sequence per lane; lane t decodes from the state
// lane t-1 LEFT its sub-sequence with and records the state it leaves its own with; this is repeated until no recorded
// exit state changes.  Lane 0 starts from the true state, so the fixed point is the true decode (induction over t).  Lanes
// whose entry state did not change keep their result, so after the first two sweeps only the few unsettled ones work.
// Then block counts and DC differences are prefix-summed across lanes and one last sweep writes the coefficients.
struct SubState { uint32_t pos; uint16_t c, z; };
__device__ __forceinline__ bool same(const SubState& a, const SubState& b) { return a.pos == b.pos && a.c == b.c && a.z == b.z; }

struct SubCtx {
    const DevHuff* huff; const int16_t* quant; const uint8_t* zag;
    const uint8_t* seg; uint32_t end_bit;
    const int* par;                     // LDS: [component] -> quant table, DC table, AC table (3 x 3 ints), the segment's image
    uint32_t org; int nb;              // org: DevImage.org
};

// Decode from state `s` while the position is inside this lane's sub-sequence.  WRITE: coefficients / max_zag go out,
// starting in block `b` with DC predictors pred[]; otherwise only the exit state, the number of blocks finished and the
// sums of the DC differences are produced.  Returns false on a stream error (only meaningful on the true path).
// The 64 lanes of a wave are at 64 different places of the token grammar (DC or AC symbol, short or long code, coefficient / ZRL /
// EOB, end of a block or not), so every branch of a symbol step is taken by SOME lane and a branchy step costs the sum of its
// paths (430 instructions as first written, half of them exec-mask bookkeeping).  The step is therefore straight-line: one refill
// (33 bits cover the longest code plus the longest value), the table picked by index, every outcome a predicate, the new state a
// handful of selects; only the rare long code (> 9 bits) and the stores are under a mask.
// MODE 0: only the exit state, the blocks finished and the DC sums.  MODE 1 (the write sweep): a block this lane STARTS is
// assembled in the lane's 128 bytes of LDS (`blk`, zero on entry) and leaves as one whole line when it ends -- or, unfinished, when
// the sub-sequence ends (its remaining coefficients belong to the lanes after this one); a block the lane finds half-decoded at its
// entry (z > 0) is only counted here.  MODE 2 (after a barrier): exactly that first, inherited block again, its coefficients stored
// one by one on top of the line its starter wrote.  Scattered 2-byte stores for everything (the first version, into a buffer
// cleared beforehand) cost 13 GB of HBM traffic per 512 images for 3 GB of coefficients: every store a read-modify-write of a line.
enum { SUB_COUNT = 0, SUB_WRITE = 1, SUB_FIRST = 2, SUB_TOKENS = 3 };
// SUB_TOKENS (round 4, the compact hand-off): the write sweep emits one 4-byte TOKEN per coefficient instead of assembling 128-byte
// blocks -- [31:26] block of the reconstruction kernel's strip, [25:20] natural position, [15:0] the de-quantised value -- every DC and
// every non-zero AC coefficient, in stream order, so the tokens of consecutive lanes are consecutive (a lane's first token index comes
// from a fifth prefix sum) and a block split between lanes needs no second pass.  The strips of the reconstruction kernel are runs of
// consecutive blocks of the stream: the lane that starts a strip's first block records where the strip's tokens begin, and the kernel
// (k_jpeg_h2v2<.., TOK>) scatters the tokens of its strip into the LDS tile it used to load from memory.  A lane collects its tokens
// in 64 bytes of LDS and ships them every 16 steps, all lanes at once (four 16-byte stores at most): shipping whole 128-byte lines the
// moment a block ended was 1.6 M of the write sweep's 4.5 M cycles per file, and a q 90 block is ~12 tokens = 48 bytes.
struct TokCtx {
    uint32_t* out;                      // this lane's region of the image's token stream
    uint32_t* lds;                      // 16 dwords of staging
    uint32_t* strip_start;              // the image's strip table
    uint32_t base, total;               // index of the lane's first token in the image's stream; tokens the lane emits (from the counting pass)
    uint32_t emitted, nl;               // shipped so far; waiting in LDS
    int sb, SB, left_in_row, row_blocks;    // block of the current strip, blocks per strip, blocks left in the MCU row
    uint32_t strip, n_strips;
    __device__ __forceinline__ void flush()
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        struct __attribute__((packed, aligned(4))) V4 { u32x4 v; };
        #pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (nl > (uint32_t)(4 * p)) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(lds + 4 * p);
                uint32_t* dst = out + emitted + 4 * p;
                if (emitted + 4 * p + 4 <= total) reinterpret_cast<V4*>(dst)->v = v;        // (the tail past nl is this lane's own: the next flush rewrites it)
                else { for (uint32_t k = 0; k < 4 && emitted + 4 * p + k < total && 4 * p + k < nl; ++k) dst[k] = v[k]; }   // the lane's last dwords: exactly
            }
        }
        emitted += nl; nl = 0;
    }
};
// Checkpoints of a counting pass: the decoder's state and running totals the first time it stands at or behind byte j * ck_step of
// its sub-sequence.  A lane that decodes its sub-sequence AGAIN (its entry state changed) falls into step with its previous pass
// after a few symbols, like every Huffman decoder; from the first checkpoint where both passes stand in the same state the rest
// is the previous pass word for word -- the lane stops there and keeps the old exit state and the old totals behind the
// checkpoint.  Without this every re-decode ran its whole sub-sequence again (3.2 full counting passes per file on average; now
// one and a fraction).  kSubCk checkpoints per lane live in the write sweep's block staging area (not in use before that sweep).
// 16 bytes each: the DC sums modulo 2^16 as well -- only differences between two passes are used, and only the low 16 bits of a DC value reach its
// coefficient (an int16 product).  A lane's checkpoints share its piece of the staging area with the write sweep's block (dense hand-off: 144
// bytes, 9 checkpoints) or token buffer (compact hand-off: 64 bytes; the area is 80 bytes then, 5 checkpoints -- 29 KB of LDS per workgroup
// instead of 45, five workgroups per compute unit instead of three: the kernel is chains of dependent operations, what it lacks is waves).
constexpr int kSubCkMax = 9;
struct SubCk { uint32_t pos; uint16_t cz, nblk, ntok; int16_t dcs[3]; };         // cz = c | z << 4; nblk / ntok modulo 2^16 (only differences are used)
static_assert(sizeof(SubCk) == 16, "checkpoint size");
struct SubTrace {
    SubCk* ck;                      // this lane's checkpoints (LDS), or nullptr
    int n_ck;                       // how many
    uint32_t first_bit, step_bits;  // checkpoint j stands at bit first_bit + j * step_bits, j = 1 .. kSubCk
    bool compare;                   // a pass after the first: stop at a checkpoint that matches
    int old_nblk, old_dcs[3];       // the previous pass's totals
    int old_ntok, ntok;             // tokens (every DC, every non-zero AC coefficient): the previous pass's total; this pass's (out)
    bool stopped;                   // out: the pass ended at a matching checkpoint (exit state = the previous one)
    int stop_ck;                    // out: which one (measurement builds)
};
template <int MODE>
__device__ __forceinline__ bool sub_decode(const SubCtx& x, SubState& s, int& nblk, int (&dcs)[3], int64_t b, int64_t b_end,
                                           int16_t* out, uint8_t* mz, int16_t* blk = nullptr, SubTrace* tr = nullptr, TokCtx* tk = nullptr)
{
    constexpr bool WRITE = MODE != SUB_COUNT;
    DevBits br; br.open(x.seg, s.pos);
    int c = s.c, z = s.z;
    nblk = 0;
    bool ok = true;
    bool inherited = z > 0;                                                      // the block under way was started by another lane
    const int q0 = x.par[0], q1 = x.par[1], q2 = x.par[2];
    // (the table is picked by NUMBER and addressed by a shift: sizeof(DevHuff) is 2048.  Picking among six addresses kept apart by an empty asm
    //  cost the loads their address space -- flat loads with 64-bit addresses instead of ds_read -- and a multiply by 1936 is a quarter-rate instruction.)
    const int d0 = x.par[3], d1 = x.par[4], d2 = x.par[5], a0 = x.par[6], a1 = x.par[7], a2 = x.par[8];
    int dc0 = dcs[0], dc1 = dcs[1], dc2 = dcs[2];
    int ntok = 0;
    int ck_j = 0;                                                                // checkpoints passed
    uint32_t ck_next = 0xFFFFFFFFu;
    uint32_t steps = 0;
    if (MODE == SUB_COUNT && tr) { ck_next = tr->first_bit + tr->step_bits; tr->stopped = false; }
    while (br.pos < x.end_bit && (!WRITE || b < b_end)) {
        br.refill();                                                             // >= 33 bits: a code (<= 16) and its value (<= 15)
        const int comp = (int)(x.org >> (2 * c)) & 3;
        const bool is_dc = z == 0;
        const DevHuff* h = x.huff + (is_dc ? (comp == 0 ? d0 : comp == 1 ? d1 : d2) : (comp == 0 ? a0 : comp == 1 ? a1 : a2));
        const uint32_t top16 = br.peek(16);
        uint32_t e = h->fast[top16 >> 7];
#if !JPEG_HUFF_SUB
        if (e >= kHuffLong) e = 0;                                               // (A/B: the canonical search for every long code, as before round 4)
#endif
        if (e >= kHuffLong) {                                                    // a code of more than 9 bits: the second level
            const uint32_t xb = (e >> 12) & 7u;
            e = h->sub[(e & 0xFFFu) + ((top16 >> (7u - xb)) & ((1u << xb) - 1u))];
        }
        int len = (int)(e >> 8), sym = (int)(e & 0xFFu);
        bool bad = false;
        if (!e) {                                                                // no second level for this prefix, or no such code: the canonical search
            int32_t code = (int32_t)(top16 >> 7); len = 9;
            while (code > h->maxcode[len]) { if (++len > 16) break; code = (int32_t)(top16 >> (16 - len)); }
            bad = len > 16;                                                      // no such code: the block ends here, 16 bits are skipped
            sym = bad ? 0 : (int)h->vals[(code + h->delta[len > 16 ? 16 : len]) & 0xFF];
            len = bad ? 16 : len;
        }
        const int size = sym & 15, run = sym >> 4;
        const bool ac = !is_dc && !bad;
        const bool err_run = ac && size != 0 && run != 0 && z + run > 63;       // a coefficient beyond the block
        const bool zrl = ac && size == 0 && run == 15;
        const bool err_zrl = zrl && z + 16 > 64;
        const bool eob = ac && size == 0 && run != 15;
        const bool take = !bad && (is_dc || (size != 0 && !err_run));           // a coefficient (or DC difference) follows the code
        const int used = len + (take ? size : 0);
        const uint32_t raw = (uint32_t)(br.acc >> (br.nbits - used)) & ((1u << size) - 1u);
        const int ext = take && size ? ((int)raw < (1 << (size - 1)) ? (int)raw + (int)(0xFFFFFFFFu << size) + 1 : (int)raw) : 0;   // JPGD_HUFF_EXTEND :816-822
        br.drop(used);
        const int k = is_dc ? 0 : z + run;                                       // zig-zag position of that coefficient
        const int dcv = (comp == 0 ? dc0 : comp == 1 ? dc1 : dc2) + ext;
        const bool tdc = take && is_dc;
        dc0 = tdc && comp == 0 ? dcv : dc0; dc1 = tdc && comp == 1 ? dcv : dc1; dc2 = tdc && comp == 2 ? dcv : dc2;
        ntok += take ? 1 : 0;
        if (MODE == SUB_TOKENS) {
            if (is_dc && tk->sb == 0) tk->strip_start[tk->strip] = tk->base + tk->emitted + tk->nl;       // this block opens a strip: its tokens begin here
            if (take) {
                const int16_t qf = x.quant[(comp == 0 ? q0 : comp == 1 ? q1 : q2) * 64 + k];
                const uint32_t cv = (uint32_t)(uint16_t)(int16_t)((uint32_t)(is_dc ? dcv : ext) * (uint32_t)(int32_t)qf);
                tk->lds[tk->nl] = (uint32_t)tk->sb << 26 | (uint32_t)x.zag[k] << 20 | cv;
                ++tk->nl;
            }
        } else
        if (WRITE && take && (MODE == SUB_FIRST || !inherited)) {
            const int16_t qf = x.quant[(comp == 0 ? q0 : comp == 1 ? q1 : q2) * 64 + k];
            const int16_t cv = (int16_t)((uint32_t)(is_dc ? dcv : ext) * (uint32_t)(int32_t)qf);
            if (MODE == SUB_FIRST) out[b * 64 + x.zag[k]] = cv; else blk[x.zag[k]] = cv;
        }
        const int znew = take ? k + 1 : (zrl && !err_zrl) ? z + 16 : z;
        const bool full = !is_dc && znew == 64;
        const bool done = bad || err_run || err_zrl || eob || full;
        ok = ok && !(bad || err_run || err_zrl);
        if (MODE == SUB_TOKENS && done) {
            mz[b] = (uint8_t)(full ? 64 : z);                                    // m_mcu_block_max_zag :2512 (EOB: the position it was read at)
            ++tk->sb; --tk->left_in_row;
            if (tk->sb == tk->SB || tk->left_in_row == 0) { tk->sb = 0; ++tk->strip; }
            if (tk->left_in_row == 0) tk->left_in_row = tk->row_blocks;
            if (b + 1 == b_end) tk->strip_start[tk->n_strips] = tk->base + tk->emitted + tk->nl;   // the segment's last block: the tokens end HERE (what the
        }                                                                                          // counting passes saw behind it is not the image's)
        if (MODE == SUB_WRITE && done) {
            mz[b] = (uint8_t)(full ? 64 : z);                                    // m_mcu_block_max_zag :2512 (EOB: the position it was read at)
            if (!inherited) {
                uint4* src = reinterpret_cast<uint4*>(blk);
                uint4* dst = reinterpret_cast<uint4*>(out + b * 64);             // 128-byte blocks at 128-byte-aligned offsets (checked by the host)
                #pragma unroll
                for (int i = 0; i < 8; ++i) { dst[i] = src[i]; src[i] = make_uint4(0, 0, 0, 0); }
            }
            inherited = false;
        }
        if (MODE == SUB_FIRST && done) break;                                    // the inherited block is complete
        b += done ? 1 : 0; nblk += done ? 1 : 0;
        z = done ? 0 : znew;
        c = done ? (c + 1 == x.nb ? 0 : c + 1) : c;
        if (MODE == SUB_TOKENS && (++steps & 15u) == 0) tk->flush();             // every lane of the wave at once: at most 16 tokens wait
        if (MODE == SUB_COUNT && br.pos >= ck_next) {                            // (never true without a trace)
            while (ck_j < tr->n_ck && br.pos >= ck_next) {                         // one symbol may step over several checkpoints (tiny steps)
                SubCk& k = tr->ck[ck_j];
                const uint16_t cz = (uint16_t)(c | z << 4);
                if (tr->compare && k.pos == br.pos && k.cz == cz) {              // in step with the previous pass from here on
                    // what this pass counted up to here instead of the previous one: the later checkpoints and the totals move by it
                    const int dn = (int16_t)((uint16_t)nblk - k.nblk), dt = (int16_t)((uint16_t)ntok - k.ntok);
                    const int e0 = (int16_t)((uint16_t)dc0 - (uint16_t)k.dcs[0]), e1 = (int16_t)((uint16_t)dc1 - (uint16_t)k.dcs[1]), e2 = (int16_t)((uint16_t)dc2 - (uint16_t)k.dcs[2]);
                    for (int jj = ck_j; jj < tr->n_ck; ++jj) {
                        SubCk& q = tr->ck[jj];
                        q.nblk = (uint16_t)(q.nblk + dn); q.ntok = (uint16_t)(q.ntok + dt);
                        q.dcs[0] = (int16_t)(q.dcs[0] + e0); q.dcs[1] = (int16_t)(q.dcs[1] + e1); q.dcs[2] = (int16_t)(q.dcs[2] + e2);
                    }
                    nblk = tr->old_nblk + dn; ntok = tr->old_ntok + dt; dc0 = tr->old_dcs[0] + e0; dc1 = tr->old_dcs[1] + e1; dc2 = tr->old_dcs[2] + e2;
                    tr->stopped = true; tr->stop_ck = ck_j;
                    break;
                }
                k.pos = br.pos; k.cz = cz; k.nblk = (uint16_t)nblk; k.ntok = (uint16_t)ntok; k.dcs[0] = (int16_t)dc0; k.dcs[1] = (int16_t)dc1; k.dcs[2] = (int16_t)dc2;
                ++ck_j; ck_next += tr->step_bits;
            }
            if (tr->stopped) break;
            if (ck_j == tr->n_ck) ck_next = 0xFFFFFFFFu;
        }
    }
    if (MODE == SUB_COUNT && tr) tr->ntok = ntok;
    // A pass that ran to its end leaves no checkpoint of an EARLIER pass standing: one that lies in the few bits a pass may overshoot its
    // sub-sequence by is reached by some passes and not by others, and a later pass that matched it would add its difference to the totals
    // of the pass before it -- which never stood there.  (Found in round 4 with nine checkpoints per lane, i.e. closer ones: the last block of
    // a restart segment lost; six had passed every test and 1 000 fuzzed files.)
    if (MODE == SUB_COUNT && tr && !tr->stopped) for (int jj = ck_j; jj < tr->n_ck; ++jj) tr->ck[jj].pos = 0xFFFFFFFFu;
    if (MODE == SUB_COUNT && tr && tr->stopped) { dcs[0] = dc0; dcs[1] = dc1; dcs[2] = dc2; return ok; }      // s: untouched, the caller keeps the previous exit state
    if (MODE == SUB_TOKENS) tk->flush();
    if (MODE == SUB_WRITE && z > 0 && !inherited && b < b_end) {                 // a block of this lane's that the next lanes finish: its line, as far as it goes
        const uint4* src = reinterpret_cast<const uint4*>(blk);
        uint4* dst = reinterpret_cast<uint4*>(out + b * 64);
        #pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = src[i];
    }
    dcs[0] = dc0; dcs[1] = dc1; dcs[2] = dc2;
    s.pos = br.pos; s.c = (uint16_t)c; s.z = (uint16_t)z;
    return ok;
}

// NH: Huffman / quantisation tables kept in LDS (0 = none: read from global memory).  A batch written by one encoder has four
// Huffman tables; with room for four instead of eight a workgroup needs 47 KB instead of 58, and a compute unit holds three of
// them instead of two -- the kernel is a chain of dependent operations per lane, what it lacks is waves to alternate with.
#ifndef JPEG_SYNC_PROFILE           // measurement only (tools/variant.sh jpeg_host:jprof:-DJPEG_SYNC_PROFILE=1, tools/jpeg_sync_prof.py): cycles per phase, thread 0's clock
#define JPEG_SYNC_PROFILE 0
#endif
#if JPEG_SYNC_PROFILE
__device__ unsigned long long g_sync_prof[16];      // 0-4 cycles per phase, 6 re-decode sweeps, 7 segments; 8 lanes that decoded again, 9 of them
                                                     // stopped at a checkpoint, 10 the sum of those checkpoints' numbers, 11 ran to the end of their sub-sequence,
                                                     // 12 of those: left in another state than before (the lane behind must decode again)
#define SPROF_DECL unsigned long long sp_t0 = clock64()
#define SPROF(slot) do { const unsigned long long now_ = clock64(); if (threadIdx.x == 0) atomicAdd(&g_sync_prof[slot], now_ - sp_t0); sp_t0 = now_; } while (0)
#define SPROF_COUNT(slot, n) do { if (threadIdx.x == 0) atomicAdd(&g_sync_prof[slot], (unsigned long long)(n)); } while (0)
#define SPROF_LANE(slot, n) atomicAdd(&g_sync_prof[slot], (unsigned long long)(n))
#else
#define SPROF_DECL
#define SPROF(slot)
#define SPROF_COUNT(slot, n)
#define SPROF_LANE(slot, n)
#endif
template <int NH, bool TOK = false>
__global__ __launch_bounds__(kSyncThreads) void k_jpeg_entropy_sync(const DevItem* items, const DevImage* images,
                                                                    const DevHuff* huff_g, int n_huff, const int16_t* quant_g, int n_quant,
                                                                    const uint8_t* blob, int16_t* coeffs, uint8_t* max_zag, uint32_t* status,
                                                                    uint32_t* tokens = nullptr, uint32_t* strip_tab = nullptr, int n_ck_arg = 0)
{
    constexpr bool IN_LDS = NH > 0;
    __shared__ DevHuff sh_huff[IN_LDS ? NH : 1];
    __shared__ int16_t sh_quant[IN_LDS ? NH * 64 : 1];
    __shared__ uint8_t sh_zag[64];
    __shared__ SubState exit_state[kSyncThreads];
    __shared__ int changed, failed, par[9];
    constexpr int kPitch = TOK ? 80 : 144, kSubCkDefault = TOK ? 5 : 6;                     // a lane's piece of the staging area; its checkpoints (up to kPitch / 16:
    const int kSubCk = n_ck_arg > 0 && n_ck_arg <= kPitch / (int)sizeof(SubCk) ? n_ck_arg : kSubCkDefault;    // GAMUT_HIP_JPEG_CHECKPOINTS, tests)             // a lane's piece of the staging area; its checkpoints
    __shared__ __attribute__((aligned(16))) uint8_t sh_blk[kSyncThreads * kPitch];          // the write sweep: a block per lane (144-byte pitch) / 16 tokens
    // blocks finished, DC-difference sums of the three components, tokens: prefix-summed between the counting passes and the write sweep,
    // in the staging area (whose checkpoints are done with by then, and which is cleared afterwards)
    int (*scan)[kSyncThreads] = reinterpret_cast<int (*)[kSyncThreads]>(sh_blk);
    static_assert(5 * kSyncThreads * sizeof(int) <= kSyncThreads * kPitch && kPitch / (int)sizeof(SubCk) <= kSubCkMax && kPitch % 16 == 0 && kPitch >= 64, "the scan, the checkpoints and 16 tokens borrow the staging area");
    load_tables<IN_LDS, kSyncThreads>(sh_huff, sh_quant, sh_zag, huff_g, n_huff, quant_g, n_quant);
    const int t = threadIdx.x;
    {
        uint4* zb = reinterpret_cast<uint4*>(sh_blk + t * kPitch);
        #pragma unroll
        for (int i = 0; i < kPitch / 16; ++i) zb[i] = make_uint4(0, 0, 0, 0);
    }
    SPROF_DECL;
    const DevItem it = items[blockIdx.x];
    const DevImage im = images[it.image];
    if (t == 0) {
        changed = 0; failed = 0;
        par[0] = im.quant[0]; par[1] = im.quant[1]; par[2] = im.quant[2]; par[3] = im.dc[0]; par[4] = im.dc[1]; par[5] = im.dc[2];
        par[6] = im.ac[0]; par[7] = im.ac[1]; par[8] = im.ac[2];
    }
    __syncthreads();
    const uint32_t len = (uint32_t)(it.end - it.begin);
    const uint32_t sub = max(64u, ((len + kSyncThreads - 1) / kSyncThreads + 3) & ~3u);     // bytes per lane
    const int nsub = (int)((len + sub - 1) / sub);
    const bool active = t < nsub;
    const SubCtx x{ IN_LDS ? sh_huff : huff_g, IN_LDS ? sh_quant : quant_g, sh_zag, blob + it.begin, min((uint32_t)(t + 1) * sub, len) * 8u,
                    par, (uint32_t)im.org, im.nb };
    int16_t* out = TOK ? nullptr : coeffs + im.coeff_off + (int64_t)it.first_mcu * im.nb * 64;
    uint8_t* mz = max_zag + im.zag_off + (int64_t)it.first_mcu * im.nb;
    const int64_t total_blocks = (int64_t)it.n_mcus * im.nb;

    // sweep 0: every lane from the start of its own sub-sequence, as if a block began there
    SubState entry{ (uint32_t)t * sub * 8u, 0, 0 }, mine = entry;
    int nblk = 0, dcs[3] = { 0, 0, 0 }, ntok = 0;
    SubTrace tr;
    tr.ck = reinterpret_cast<SubCk*>(sh_blk + t * kPitch); tr.n_ck = kSubCk; tr.first_bit = (uint32_t)t * sub * 8u; tr.step_bits = ((sub + (uint32_t)kSubCk) / (uint32_t)(kSubCk + 1)) * 8u;
    tr.compare = false; tr.old_nblk = 0; tr.old_dcs[0] = tr.old_dcs[1] = tr.old_dcs[2] = 0; tr.old_ntok = 0; tr.ntok = 0; tr.stopped = false;
    if (active) {
        for (int j = 0; j < kSubCk; ++j) tr.ck[j].pos = 0xFFFFFFFFu;             // (a pass that ends early leaves the later ones unset: never a match)
        sub_decode<SUB_COUNT>(x, mine, nblk, dcs, 0, 0, nullptr, nullptr, nullptr, &tr);
        ntok = tr.ntok;
    }
    exit_state[t] = mine;
    __syncthreads();
    SPROF(0);
    // sweeps 1..: from the predecessor's exit state, until nothing moves
    bool settled = false;
    for (int sweep = 0; sweep < kSyncThreads + 1; ++sweep) {
        SubState from = t == 0 ? SubState{ 0, 0, 0 } : exit_state[t - 1];
        __syncthreads();                                        // everybody has read its predecessor
        if (active && (sweep == 0 || !same(from, entry))) {
            entry = from; mine = from;
            tr.compare = true; tr.old_nblk = nblk; tr.old_dcs[0] = dcs[0]; tr.old_dcs[1] = dcs[1]; tr.old_dcs[2] = dcs[2]; tr.old_ntok = ntok;
            dcs[0] = dcs[1] = dcs[2] = 0;
            sub_decode<SUB_COUNT>(x, mine, nblk, dcs, 0, 0, nullptr, nullptr, nullptr, &tr);
            ntok = tr.ntok;
            SPROF_LANE(8, 1); if (tr.stopped) { SPROF_LANE(9, 1); SPROF_LANE(10, tr.stop_ck); } else { SPROF_LANE(11, 1); if (!same(mine, exit_state[t])) SPROF_LANE(12, 1); }
            if (!tr.stopped && !same(mine, exit_state[t])) { exit_state[t] = mine; changed = 1; }
        }
        __syncthreads();
        const int any = changed;
        __syncthreads();
        if (t == 0) changed = 0;
        SPROF_COUNT(6, 1);
        if (!any) { settled = true; break; }
    }
    SPROF(1);
    (void)settled;                                              // the loop bound (one lane settles per sweep at worst) always reaches the fixed point
    // exclusive prefix sums over the lanes: first block, DC predictors and first token of every lane
    scan[0][t] = active ? nblk : 0; scan[1][t] = active ? dcs[0] : 0; scan[2][t] = active ? dcs[1] : 0; scan[3][t] = active ? dcs[2] : 0;
    scan[4][t] = active ? ntok : 0;
    __syncthreads();
    for (int d = 1; d < kSyncThreads; d <<= 1) {
        int v[5];
        #pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = t >= d ? scan[k][t - d] : 0;
        __syncthreads();
        #pragma unroll
        for (int k = 0; k < 5; ++k) scan[k][t] += v[k];
        __syncthreads();
    }
    const int64_t b0 = active ? scan[0][t] - nblk : 0;
    const int pred_in[3] = { scan[1][t] - dcs[0], scan[2][t] - dcs[1], scan[3][t] - dcs[2] };
    const int64_t finished = scan[0][kSyncThreads - 1];
    const uint32_t tok0 = (uint32_t)(scan[4][t] - ntok), tok_total = (uint32_t)scan[4][kSyncThreads - 1];
    __syncthreads();                                            // everybody has its prefix sums: the area is the staging area again
    SPROF(2);
    {                                                           // the checkpoints are done with: the staging area starts out zero
        uint4* zb = reinterpret_cast<uint4*>(sh_blk + t * kPitch);
        #pragma unroll
        for (int i = 0; i < kPitch / 16; ++i) zb[i] = make_uint4(0, 0, 0, 0);
    }
    if constexpr (TOK) {
        // the compact hand-off: tokens instead of blocks (sub_decode<SUB_TOKENS>).  A stream that would emit more tokens than the image's
        // slot holds (only a damaged one can: the slot is sized from the scan's length) emits none and is flagged.
        uint32_t* const strip_start = strip_tab + im.strip_off;
        const bool fits = tok_total <= (uint32_t)im.tok_cap;
        if (!fits && t == 0) failed = 1;
        if (active && fits) {
            int pred[3] = { pred_in[0], pred_in[1], pred_in[2] };
            SubState s = entry; int n2 = 0;
            TokCtx tk;
            tk.out = tokens + im.tok_off + tok0; tk.lds = reinterpret_cast<uint32_t*>(sh_blk + t * kPitch); tk.strip_start = strip_start;
            tk.base = tok0; tk.total = (uint32_t)ntok; tk.emitted = 0; tk.nl = 0;
            tk.SB = im.sb; tk.row_blocks = im.row_blocks; tk.n_strips = (uint32_t)im.n_strips;
            const int64_t gb = (int64_t)it.first_mcu * im.nb + b0;               // the lane's first block, in the image
            const int64_t row = gb / im.row_blocks; const int local = (int)(gb - row * im.row_blocks);
            const int strips_per_row = (im.row_blocks + im.sb - 1) / im.sb;
            tk.strip = (uint32_t)(row * strips_per_row + local / im.sb); tk.sb = local % im.sb; tk.left_in_row = im.row_blocks - local;
            if (!sub_decode<SUB_TOKENS>(x, s, n2, pred, b0, total_blocks, nullptr, mz, nullptr, nullptr, &tk)) failed = 1;
            if (t == nsub - 1 && b0 + n2 < total_blocks) failed = 1;      // the segment ended before its last block did (behind it: not this interval's)
            if (it.tail >= 0 && b0 < total_blocks && b0 + n2 == total_blocks && restart_leftover_bad(x.seg, len, s.pos, it.tail)) atomicOr(status + it.image, kStatusBadRestart);   // the lane that ended the last block
        }
        __syncthreads();
        // the strips no block opened (a damaged stream ends early) are empty: their tokens "begin" at the end; blocks nobody reached
        // have no coefficients (their strips' tiles stay zero) and max_zag 1
        int64_t started = fits ? finished + (exit_state[nsub > 0 ? nsub - 1 : 0].z > 0 ? 1 : 0) : 0;
        started = started > total_blocks ? total_blocks : started;
        const int strips_per_row = (im.row_blocks + im.sb - 1) / im.sb;
        int64_t first_empty = 0;
        if (started > 0) { const int64_t lb = started - 1, row = lb / im.row_blocks; first_empty = row * strips_per_row + (lb - row * im.row_blocks) / im.sb + 1; }
        const uint32_t end_tok = fits ? tok_total : 0u;
        const bool complete = fits && finished >= total_blocks;                   // the lane that finished the last block wrote the end of the tokens itself
        for (int64_t sidx = first_empty + t; sidx < im.n_strips + (complete ? 0 : 1); sidx += kSyncThreads) strip_start[sidx] = end_tok;
        for (int64_t blk = started + t; blk < total_blocks; blk += kSyncThreads) mz[blk] = 1;
        SPROF(3);
        SPROF(4);
        SPROF_COUNT(7, 1);
        if (t == 0 && failed) atomicOr(status + it.image, 1u);
        return;
    }
    if (active) {
        int pred[3] = { pred_in[0], pred_in[1], pred_in[2] };
        SubState s = entry; int n2 = 0;
        if (!sub_decode<SUB_WRITE>(x, s, n2, pred, b0, total_blocks, out, mz, reinterpret_cast<int16_t*>(sh_blk + t * kPitch))) failed = 1;
        if (t == nsub - 1 && b0 + n2 < total_blocks) failed = 1;      // the segment ended before its last block did (behind it: not this interval's)
        if (it.tail >= 0 && b0 < total_blocks && b0 + n2 == total_blocks && restart_leftover_bad(x.seg, len, s.pos, it.tail)) atomicOr(status + it.image, kStatusBadRestart);       // the lane that ended the last block
    }
    __threadfence();                                            // the lines are on their way before anybody adds single coefficients to them
    __syncthreads();
    SPROF(3);
    if (active && entry.z > 0 && b0 < total_blocks) {           // the rest of the block this lane found half-decoded
        int pred[3] = { 0, 0, 0 }; SubState s = entry; int n3 = 0;
        sub_decode<SUB_FIRST>(x, s, n3, pred, b0, total_blocks, out, mz);
    }
    // A segment whose decode ended early (damaged data) leaves blocks nobody started: they are cleared, not left as they were
    const int64_t started = finished + (exit_state[nsub > 0 ? nsub - 1 : 0].z > 0 ? 1 : 0);
    for (int64_t blk = started + t; blk < total_blocks; blk += kSyncThreads) {
        uint4* dst = reinterpret_cast<uint4*>(out + blk * 64);
        #pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    SPROF(4);
    SPROF_COUNT(7, 1);
    if (t == 0 && failed) atomicOr(status + it.image, 1u);
}

// ---- the scan as it stands in the file -> the unstuffed, padded segments the two kernels above read, ON THE DEVICE --------------------
// get_bits_no_markers (jpegload.d:722-743) and get_octet (:683-696) skip the 0x00 behind a data 0xFF while they read, and
// process_restart (:2335-2402) finds the RSTn markers; round 3 did both on host threads (unstuff_file below: memchr + many small
// memcpy into pinned memory, "unstuff + upload at 33 GB/s", and a thread pool that is divided by the ranks of the node).  Here the host
// copies the file's bytes from the first scan byte on into pinned memory as they are -- one memcpy -- and a workgroup per file turns them
// into the blob form: tiles of 4 KiB, a byte dropped when it is 0x00 behind 0xFF (prefix sum of the kept bytes over the workgroup,
// staged in LDS, written out as whole aligned 16-byte chunks); the data of a segment ends at the first 0xFF that is followed by
// anything but 0x00 (rare: a tile is processed up to the first such place and the marker handled by the whole workgroup in step) --
// an expected RSTn closes the segment (64 bytes of 0xFF behind it, as the host writes them) and opens the next, anything else ends the
// scan.  The segment list (image, first MCU, MCU count per restart interval) comes from the host, which knows it from the header;
// begin / end are filled in here.  A wrong or missing restart marker flags the file (status bit 3) and empties its segments.
struct DevRaw {
    uint64_t raw_begin, raw_len;        // the file from its first scan byte to its end, in the raw blob
    uint64_t out_begin, out_cap;        // the file's slot in the unstuffed blob (16-byte aligned)
    int32_t  image, first_item, n_items, restart_interval;
};
constexpr int kUnstuffThreads = 256, kUnstuffTile = kUnstuffThreads * 16;
__global__ __launch_bounds__(kUnstuffThreads) void k_jpeg_unstuff(const DevRaw* files, DevItem* items, const uint8_t* raw, uint8_t* blob, uint32_t* status,
                                                                   uint32_t* scan_end /* [image]: bytes from the scan's first byte to the marker that ends it (find_eoi starts there) */)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    struct __attribute__((packed, aligned(1))) AnyVec { u32x4 v; };
    __shared__ __attribute__((aligned(16))) uint8_t stage[16 + kUnstuffTile + 64 + 16];
    __shared__ uint32_t wave_sum[kUnstuffThreads / 64];
    __shared__ uint32_t first_term;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const DevRaw f = files[blockIdx.x];
    const uint8_t* src = raw + f.raw_begin;
    uint8_t* const dst = blob + f.out_begin;
    uint64_t cur = 0, w = 0, seg_begin = 0;                 // next raw byte; bytes produced (the last w % 16 of them still in stage[])
    uint32_t carry = 0;
    int seg = 0, expect = 0;
    bool bad = false;
    // stage[0 .. m) = the last m of the w bytes produced -> dst, whole 16-byte chunks; the rest moves to the front (all threads call it)
    auto flush = [&](uint32_t m) {
        __syncthreads();
        const uint32_t chunks = m >> 4;
        const uint64_t base = w - m;                         // 16-byte aligned by construction
        for (uint32_t c = t; c < chunks; c += kUnstuffThreads) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(stage + c * 16);
            if (base + c * 16 + 16 <= f.out_cap) *reinterpret_cast<u32x4*>(dst + base + c * 16) = v;
        }
        uint8_t keep = 0;
        const uint32_t rest = m & 15u;
        if ((uint32_t)t < rest) keep = stage[chunks * 16 + t];
        __syncthreads();
        if ((uint32_t)t < rest) stage[t] = keep;
        __syncthreads();
        return rest;
    };
    auto close_segment = [&](int tail) {                     // the segment's item, then its 64 bytes of padding; tail: DevItem.tail
        if (t == 0 && seg < f.n_items) { items[f.first_item + seg].begin = f.out_begin + seg_begin; items[f.first_item + seg].end = f.out_begin + w; items[f.first_item + seg].tail = tail; }
        if (t < 64) stage[carry + t] = 0xFF;
        const uint32_t m = carry + 64;
        w += 64;
        carry = flush(m);
        seg_begin = w; ++seg;
    };
    bool ended = false;
    while (!ended) {
        const uint64_t left = f.raw_len - cur;
        const uint32_t n = left < (uint64_t)kUnstuffTile ? (uint32_t)left : (uint32_t)kUnstuffTile;
        if (t == 0) first_term = 0xFFFFFFFFu;
        __syncthreads();
        // this thread's 16 bytes, the byte in front of them and the byte behind them (zeros outside the data)
        const uint32_t o = (uint32_t)t * 16u;
        uint8_t b[18];
        #pragma unroll
        for (int k = 0; k < 18; ++k) b[k] = 0;
        if (o < n) {
            const u32x4 v = reinterpret_cast<const AnyVec*>(src + cur + o)->v;          // (the raw blob has 32 bytes of slack behind every file)
            #pragma unroll
            for (int k = 0; k < 16; ++k) b[1 + k] = (uint8_t)(v[k >> 2] >> ((k & 3) * 8));
            b[0] = cur + o > 0 ? src[cur + o - 1] : 0;
            b[17] = src[cur + o + 16];
        }
        uint32_t term = 0xFFFFFFFFu;
        #pragma unroll
        for (int k = 15; k >= 0; --k) {
            const uint64_t at = cur + o + (uint32_t)k;
            if (o + (uint32_t)k < n && b[1 + k] == 0xFF && at + 1 < f.raw_len && b[2 + k] != 0x00) term = o + (uint32_t)k;
        }
        if (term != 0xFFFFFFFFu) atomicMin(&first_term, term);
        __syncthreads();
        const uint32_t p = first_term;                       // offset of the first data-ending 0xFF in the tile, if any
        const uint32_t region = p < n ? p : n;
        // kept bytes of [0, region): everything but a 0x00 behind a 0xFF
        uint32_t cnt = 0, keepmask = 0;
        #pragma unroll
        for (int k = 0; k < 16; ++k) {
            const bool keep = o + (uint32_t)k < region && !(b[1 + k] == 0x00 && b[k] == 0xFF);
            keepmask |= keep ? 1u << k : 0u; cnt += keep ? 1u : 0u;
        }
        uint32_t incl = cnt;                                 // inclusive prefix sum over the wave, then over the workgroup
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
        #pragma unroll
        for (int k = 0; k < kUnstuffThreads / 64; ++k) { const uint32_t s = wave_sum[k]; before += k < wave ? s : 0u; total += s; }
        const uint32_t at = carry + before + incl - cnt;
        if (cnt == 16) {
            u32x4 v;
            #pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (uint32_t)b[1 + 4 * k] | (uint32_t)b[2 + 4 * k] << 8 | (uint32_t)b[3 + 4 * k] << 16 | (uint32_t)b[4 + 4 * k] << 24;
            reinterpret_cast<AnyVec*>(stage + at)->v = v;                               // (LDS takes byte-unaligned 16-byte accesses on gfx950)
        } else if (cnt) {
            uint32_t q = at;
            #pragma unroll
            for (int k = 0; k < 16; ++k) if (keepmask >> k & 1u) stage[q++] = b[1 + k];
        }
        w += total;
        carry = flush(carry + total);
        if (p < n) {                                         // a marker: skip fill bytes, look at its code (every thread reads the same bytes)
            uint64_t j = cur + p + 1;
            while (j < f.raw_len && src[j] == 0xFF) ++j;
            const uint32_t code = j < f.raw_len ? src[j] : 0xD9u;
            const bool rst = code >= 0xD0u && code <= 0xD7u && f.restart_interval > 0 && seg + 1 < f.n_items;
            if (rst && code != 0xD0u + (uint32_t)expect) { bad = true; ended = true; }
            else if (rst) { close_segment((int)min(j - (cur + p + 1), (uint64_t)4096)); expect = (expect + 1) & 7; cur = j + 1; }
            else { ended = true; if (t == 0) scan_end[f.image] = (uint32_t)(cur + p); }      // EOI or any other marker ends the scan
        } else {
            cur += n;
            if (cur >= f.raw_len) { ended = true; if (t == 0) scan_end[f.image] = (uint32_t)f.raw_len; }   // the file ends inside the scan: what is there is the data
        }
    }
    if (!bad) {
        if (seg + 1 == f.n_items) close_segment(-1);
        else bad = true;                                     // a restart marker is missing
    }
    if (carry) {                                             // the last, partial chunk (the slot has room for a whole one)
        __syncthreads();
        if (t == 0 && (w - carry) + 16 <= f.out_cap) *reinterpret_cast<u32x4*>(dst + (w - carry)) = *reinterpret_cast<const u32x4*>(stage);
    }
    if (bad) {
        for (int k = t; k < f.n_items; k += kUnstuffThreads) { items[f.first_item + k].begin = f.out_begin; items[f.first_item + k].end = f.out_begin; }
        if (t == 0) atomicOr(status + f.image, kStatusBadRestart);
    }
}

// Host side: header walk per file (on host threads), table de-duplication, restart-interval index, one upload, two launches.
// geometry + the position of the single baseline scan; shared by read_header and the device decoder
int parse_baseline(Parser& P, const uint8_t* data, size_t len, gamut_hip_jpeg_frame* f, bool want_scan)
{
    const int first = open_frame(P, data, len, f);
    if (first < 0) return GAMUT_HIP_ERR_DECODE;
    if (!want_scan) return first == 0xDA && !scan_tables_ok(P, f) ? GAMUT_HIP_ERR_DECODE : GAMUT_HIP_OK;      // (what the first scan would be refused for)
    if (P.progressive) return set_error(GAMUT_HIP_ERR_UNSUPPORTED, "jpeg: progressive frames are decoded by the host feeder (gamut_hip_jpeg_decode_coeffs)");
    if (P.scan.ncomp != f->comps) return fail(f, "only single-scan baseline files are supported");
    int order[6];
    if (!baseline_order_ok(P, f, order)) return fail(f, "the scan lists a component twice");
    if (!scan_tables_ok(P, f)) return GAMUT_HIP_ERR_DECODE;    // check_huff_tables / check_quant_tables of init_scan :3101-3106
    return GAMUT_HIP_OK;
}

template <class T> int intern(std::vector<T>& pool, const T& t)      // index of a bit-identical table, appended if new
{
    for (size_t i = 0; i < pool.size(); ++i) if (!memcmp(&pool[i], &t, sizeof(T))) return (int)i;
    pool.push_back(t);
    return (int)pool.size() - 1;
}
struct QuantTab { int16_t q[64]; };
// a DevHuff is a function of the code-length counts and the symbols (to_dev_huff): maxcode[] / delta[] hold the former, vals[] the latter -- 396 bytes
// decide whether two tables are the same one, not 2048 (the serial de-duplication walks six tables of every file of a batch)
inline int intern(std::vector<DevHuff>& pool, const DevHuff& t)
{
    static_assert(offsetof(DevHuff, vals) + sizeof(t.vals) - offsetof(DevHuff, maxcode) == sizeof(t.maxcode) + sizeof(t.delta) + sizeof(t.vals), "DevHuff: maxcode, delta, vals adjacent");
    const size_t from = offsetof(DevHuff, maxcode), n = sizeof(t.maxcode) + sizeof(t.delta) + sizeof(t.vals);
    for (size_t i = 0; i < pool.size(); ++i) if (!memcmp((const char*)&pool[i] + from, (const char*)&t + from, n)) return (int)i;
    pool.push_back(t);
    return (int)pool.size() - 1;
}

// What the host prepares for one file, independently of every other file (so files are spread over host threads), in two
// steps: (A) header walk + tables in device form + an upper bound of the file's share of the upload; (C) the entropy-coded
// segments copied UNSTUFFED (the 0x00 after a data 0xFF dropped) straight into the pinned upload image, cut at the RSTn
// markers, each followed by 64 bytes of 0xFF.
struct FilePrep {
    int rc = GAMUT_HIP_OK; char msg[200] = { 0 };
    bool progressive = false;                                  // SOF2: left to progressive_decode_device (rc = kDeferred)
    int comps = 0, nb = 0, ny = 0; uint32_t org = 0;           // org: DevImage.org
    QuantTab quant[3];
    // [component][DC, AC]: six tables of 2 KB -- not inside the record: a vector of 1024 records was 13 MB that the constructor cleared on ONE thread
    // in front of everything else (most of the "headers" phase).  The batch call points them into one uninitialised block per thread that it keeps.
    DevHuff (*huff)[2] = nullptr;
    size_t scan_pos = 0, cap = 0, used = 0;                    // first scan byte in the file; bound / actual size of the unstuffed segments
    int restart_interval = 0, total_mcus = 0;
    bool dev_unstuff = false;                                  // the scan goes up as it is and k_jpeg_unstuff makes the segments (items: begin / end filled in there)
    size_t raw_len = 0;                                        // bytes from scan_pos to the end of the file
    size_t scan_end = 0;                                       // host-unstuffed files: where the scan's data ended (the marker find_eoi starts at), from the start of the file
    std::vector<DevItem> items;                                // begin / end relative to the file's slot in the blob; pad = (estimated) bytes of the segment
};
// FilePrep.rc of a file the host feeder decodes after the device pass: not for the kernels (prepare_header), a restart structure the unstuffing could
// not follow, or anything a kernel flagged -- a bit pattern no code word begins, a segment that ran out, octets left in front of a marker.  The
// reference has a result for many of those (jpgd decodes symbol 0 and carries on); the host feeder is the part of the library that knows them all.
constexpr int kHostRedo = -1001;
// where the 0x00 stuffing is dropped and the restart markers are found: GAMUT_HIP_JPEG_UNSTUFF=host / device forces either (tests, measurements)
int unstuff_site() { const char* e = getenv("GAMUT_HIP_JPEG_UNSTUFF"); return !e || !*e ? 0 : !strcmp(e, "host") ? 1 : !strcmp(e, "device") ? 2 : 0; }

void prepare_header(int i, const uint8_t* base, size_t n, gamut_hip_jpeg_frame& f, FilePrep& out, Parser& P)
{
    P = Parser();
    out.rc = parse_baseline(P, base, n, &f, true);
    out.progressive = out.rc == GAMUT_HIP_ERR_UNSUPPORTED && P.progressive;
    if (out.rc != GAMUT_HIP_OK) { snprintf(out.msg, sizeof(out.msg), "image %d: %s", i, last_error_buf()); return; }
    // Files the kernels are not given (the host feeder decodes them behind the device pass, host_redo): a scan header that ends in the padding behind
    // the file (the scan's data is what the padding holds), a DC table with a category above 15 (the kernels mask it, the reference leaves its tables)
    if (P.pos >= n) { out.rc = kHostRedo; return; }
    for (int k = 0; k < P.scan.ncomp; ++k) if (P.huff[P.td[P.scan.comp[k]]].dc_limit > 15) { out.rc = kHostRedo; return; }
    out.comps = f.comps; out.nb = f.blocks_per_mcu; out.ny = f.comps == 1 ? 1 : P.hs[0] * P.vs[0];
    { int order[6] = { 0, 0, 0, 0, 0, 0 }; const int blocks = scan_block_order(P, order, 6); out.org = 0; for (int b = 0; b < blocks && b < 6; ++b) out.org |= (uint32_t)order[b] << (2 * b); }
    for (int c = 0; c < f.comps; ++c) {
        memcpy(out.quant[c].q, P.quant[P.tq[c]], sizeof(out.quant[c].q));
        for (int k = 0; k < 2; ++k) {
            to_dev_huff(P.huff[k ? P.ta[c] : P.td[c]], out.huff[c][k]);
        }
    }
    out.scan_pos = P.pos; out.restart_interval = P.restart_interval; out.total_mcus = f.mcus_per_row * f.mcus_per_col;
    // unstuffing only drops bytes; every segment (at most one per restart interval) gains 64 bytes of padding
    const size_t segs = out.restart_interval ? (size_t)(out.total_mcus / out.restart_interval + 1) : 1;
    out.cap = (n - out.scan_pos) + 64 * segs + 16;
    out.raw_len = n - out.scan_pos;
    // On the device unless the restart intervals are tiny (the kernel handles a marker by the whole workgroup in step: a file of
    // thousands of 30-byte intervals is quicker through memchr on a host thread; its segments take a lane each anyway)
    const size_t n_seg = out.restart_interval ? (size_t)((out.total_mcus + out.restart_interval - 1) / out.restart_interval) : 1;
    const size_t est = n_seg ? out.raw_len / n_seg : out.raw_len;
    out.dev_unstuff = unstuff_site() == 2 || (unstuff_site() == 0 && (out.restart_interval == 0 || est >= 1024));
    if (out.dev_unstuff) {
        out.items.resize(n_seg);
        for (size_t k = 0; k < n_seg; ++k) {
            DevItem& it = out.items[k]; memset(&it, 0, sizeof(it));
            it.image = i; it.first_mcu = out.restart_interval ? (int)k * out.restart_interval : 0;
            it.n_mcus = out.restart_interval ? std::min(out.restart_interval, out.total_mcus - it.first_mcu) : out.total_mcus;
            it.pad = (int32_t)std::min<size_t>(est, 0x7fffffff);
        }
        out.used = out.cap;
        // (Tried: scans of files that live in page-locked memory uploaded from where they are, a DMA per file -- 1024 x 225 kB: 17.0 ->
        // 20.1 ms, profiles/r04_pinned.txt: a DMA per file costs more than the copy into one pinned image and one DMA per group.  The QOI
        // call, whose files are megabytes, keeps that path: same time, no host copy.)
    }
}

void unstuff_file(int i, const uint8_t* base, size_t n, gamut_hip_jpeg_frame& f, FilePrep& out, uint8_t* dst)
{
    const int total_mcus = out.total_mcus, ri = out.restart_interval;
    int next_mcu = 0, expect = 0;
    size_t q = out.scan_pos, copy_from = out.scan_pos, seg_begin = 0, w = 0;          // w = bytes written to dst
    bool copying = true, bad = false;
    auto flush = [&](size_t upto) {
        if (copying && upto > copy_from) {
            const size_t k = upto - copy_from;
            if (w + k > out.cap) { bad = true; return; }                               // cannot happen (see cap); never write past the slot
            memcpy(dst + w, base + copy_from, k); w += k;
        }
    };
    int fill = 0;                                          // 0xFF bytes in front of the marker under way, besides its own
    auto close_segment = [&](int nm, int tail) {
        if (w + 64 > out.cap) { bad = true; return; }
        DevItem it{}; it.image = i; it.first_mcu = next_mcu; it.n_mcus = nm; it.begin = seg_begin; it.end = w; it.tail = tail;
        it.pad = (int32_t)std::min<size_t>(w - seg_begin, 0x7fffffff);
        out.items.push_back(it); next_mcu += nm;
        memset(dst + w, 0xFF, 64); w += 64;
        seg_begin = w;
    };
    while (!bad) {
        const uint8_t* hit = q < n ? (const uint8_t*)memchr(base + q, 0xFF, n - q) : nullptr;
        if (!hit || hit + 1 >= base + n) { flush(n); q = n; break; }
        const uint8_t m = hit[1];
        q = (size_t)(hit - base);
        if (m == 0x00) {
            if (!copying) { bad = true; break; }                                          // FF .. FF 00: the reference's input stopped at the FIRST of these FFs (not a stuffed byte: get_octet
            flush(q + 1); copy_from = q + 2; q += 2; continue;                            //  looks one byte ahead), process_restart then reads "FF, fill, 00": for the host feeder (kHostRedo)
        }                                                                                 // stuffed 0xFF: keep the FF, drop the 00
        flush(q); copying = false;                                                        // FF + non-zero: the data ends here (get_octet :683-696)
        if (m == 0xFF) { q += 1; fill = std::min(fill + 1, 4096); continue; }             // fill bytes before a marker
        if (m >= 0xD0 && m <= 0xD7 && ri && next_mcu + ri < total_mcus) {
            if (m != 0xD0 + expect) { bad = true; break; }
            close_segment(ri, fill); fill = 0;
            expect = (expect + 1) & 7; q += 2; copy_from = q; copying = true;
            continue;
        }
        break;                                             // EOI or any other marker ends the scan
    }
    if (!bad && next_mcu < total_mcus) {
        if (ri && total_mcus - next_mcu > ri) bad = true;      // a restart marker is missing
        else close_segment(total_mcus - next_mcu, -1);
    }
    out.used = w; out.scan_end = q;
    if (bad) { out.items.clear(); out.used = 0; out.rc = kHostRedo; }      // a wrong or missing RSTn: what the reference makes of it is the host feeder's to say
}

// The files of a batch the host feeder decodes (decode_coeffs, the whole of it: markers, scans, find_eoi): `deliver` takes the dense coefficients
// of a file that decoded; rcs[k] / msgs[k] receive the verdicts.  info[] gets the geometry and the density as the host feeder found them.
int host_redo(const uint8_t* const* data, const size_t* len, const std::vector<int>& idx, gamut_hip_jpeg_frame* info,
              const std::function<int(int i, const gamut_hip_jpeg_frame& fr)>& deliver, std::vector<int>& rcs, std::vector<std::string>& msgs)
{
    const int n = (int)idx.size();
    rcs.assign((size_t)n, GAMUT_HIP_OK); msgs.assign((size_t)n, std::string());
    if (!n) return GAMUT_HIP_OK;
    int workers = host_threads();
    workers = workers < 1 ? 1 : workers > 16 ? 16 : workers;
    if (workers > n) workers = n;
    // in chunks of `workers` files -- decode, deliver, free: a batch of a thousand damaged 4K files would otherwise hold their dense
    // coefficients (25 MB each) in host memory all at once
    int hip_rc = GAMUT_HIP_OK;
    std::vector<gamut_hip_jpeg_frame> frames((size_t)workers);
    for (int k0 = 0; k0 < n; k0 += workers) {
        const int m = std::min(workers, n - k0);
        for (int j = 0; j < m; ++j) memset(&frames[(size_t)j], 0, sizeof(gamut_hip_jpeg_frame));
        parallel_for(m, m, [&](int, int j) {
            const int k = k0 + j, i = idx[(size_t)k];
            rcs[(size_t)k] = decode_coeffs(data[i], len[i], &frames[(size_t)j]);
            if (rcs[(size_t)k] != GAMUT_HIP_OK) msgs[(size_t)k] = last_error_buf();
        });
        for (int j = 0; j < m; ++j) {
            const int k = k0 + j, i = idx[(size_t)k];
            gamut_hip_jpeg_frame& fr = frames[(size_t)j];
            if (rcs[(size_t)k] == GAMUT_HIP_OK && hip_rc == GAMUT_HIP_OK) hip_rc = deliver(i, fr);
            info[i] = fr; info[i].coeffs = nullptr; info[i].max_zag = nullptr;
            free(fr.coeffs); free(fr.max_zag); fr.coeffs = nullptr; fr.max_zag = nullptr;
        }
    }
    return hip_rc;
}
// ... into the caller's dense buffers (the coefficient-level entry point; progressive files at either level)
inline std::function<int(int, const gamut_hip_jpeg_frame&)> deliver_dense(const int64_t* coeff_offset, const int64_t* zag_offset, int16_t* d_coeffs, uint8_t* d_max_zag,
                                                                           uint32_t* d_status, hipStream_t stream)
{
    return [=](int i, const gamut_hip_jpeg_frame& fr) -> int {
        const size_t nblk = (size_t)fr.mcus_per_row * fr.mcus_per_col * fr.blocks_per_mcu;
        if (hipMemcpyAsync(d_coeffs + coeff_offset[i], fr.coeffs, nblk * 64 * sizeof(int16_t), hipMemcpyHostToDevice, stream) != hipSuccess ||
            hipMemcpyAsync(d_max_zag + zag_offset[i], fr.max_zag, nblk, hipMemcpyHostToDevice, stream) != hipSuccess ||
            (d_status && hipMemsetAsync(d_status + i, 0, sizeof(uint32_t), stream) != hipSuccess) ||
            hipStreamSynchronize(stream) != hipSuccess)          // (pageable sources, freed when host_redo returns)
            return set_error(GAMUT_HIP_ERR_HIP, "jpeg: upload of the coefficients of image %d failed", i);
        return GAMUT_HIP_OK;
    };
}


#include "jpeg_prog.hpp"

constexpr int kDeferred = -1000;                               // FilePrep.rc of a progressive file: not this function's business

// Hooks of the files -> pixels entry point (gamut_hip_jpeg_decode_batch_device): `layout` runs once the headers are known (it
// sizes and places the coefficient buffers: the caller of the coefficient-level entry point did that beforehand);
// `group_done(lo, hi, gs)` runs right after the entropy kernels of files [lo, hi) have been queued on stream gs -- the
// reconstruction of a group is queued behind its own entropy decode, beside the decode of the next group.
// The compact hand-off (tokens instead of dense blocks, SUB_TOKENS above) is the hook owner's choice per image: `tok_ok[i]` says which
// images could take it (4:2:0, one long segment, scan unstuffed on the device), `scan_len[i]` bounds their token count; the hook sets
// tok[i] and places the image's token slot and strip table (tok_off / strip_off / tok_cap; the buffers through d_tokens / d_strip_tab).
struct TokenPlan {
    const char* tok_ok; const size_t* scan_len;               // in
    char* tok; int64_t* tok_off; int64_t* strip_off; int32_t* tok_cap; uint32_t** d_tokens; uint32_t** d_strip_tab;     // out
};
struct DecodeHooks {
    std::function<int(const gamut_hip_jpeg_frame* info, const int* rc, const char* progressive, int64_t* coeff_offset, int64_t* zag_offset, int16_t** d_coeffs, uint8_t** d_max_zag,
                      const TokenPlan& plan)> layout;
    std::function<int(int lo, int hi, hipStream_t gs)> group_done;
    std::function<int(int i, const gamut_hip_jpeg_frame& fr, hipStream_t s)> redo;      // a file the host feeder decoded (host_redo): coefficients -> the hook owner's output
};
int entropy_decode_device(const uint8_t* const* data, const size_t* len, int count,
                          const int64_t* coeff_offset_in, const int64_t* zag_offset_in,
                          int16_t* d_coeffs, uint8_t* d_max_zag, uint32_t* d_status,
                          gamut_hip_jpeg_frame* info, int* host_status, hipStream_t stream, const DecodeHooks* hooks = nullptr)
{
    std::vector<int64_t> own_offsets;
    const int64_t* coeff_offset = coeff_offset_in; const int64_t* zag_offset = zag_offset_in;
    if (hooks) { own_offsets.assign((size_t)count * 2, 0); coeff_offset = own_offsets.data(); zag_offset = own_offsets.data() + count; }
    const bool trace = getenv("GAMUT_HIP_TRACE") != nullptr;           // stage timings on stderr (tools/e2e_bench.py)
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    if (!hooks && ((uintptr_t)d_coeffs & 15) != 0) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_entropy_decode_device: the coefficient buffer must be 16-byte aligned");
    for (int i = 0; i < count && !hooks; ++i)
        if ((coeff_offset[i] & 7) != 0) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_entropy_decode_device: coefficient offsets must be multiples of 8 elements (image %d)", i);

    // Pipeline: (A) headers on host threads -> (B) layout, and the coefficient clears go out on `stream` -> (C) segments
    // unstuffed straight into the pinned upload image, slice by slice, each slice DMA'd on a private copy stream as soon as
    // it is complete (so unstuffing, PCIe and the clears overlap) -> (D) tables, then the kernels on `stream` behind an event.
    int workers = host_threads();
    workers = workers < 1 ? 1 : workers > 16 ? 16 : workers;
    if (workers > (count + 7) / 8) workers = (count + 7) / 8;
    std::vector<FilePrep> prep((size_t)count);
    {
        struct FreeIt { void operator()(DevHuff* p) const { free(p); } };
        static thread_local std::unique_ptr<DevHuff, FreeIt> table_store; static thread_local size_t table_store_cap = 0;      // (kept by the thread, host memory only: warm pages,
        if ((size_t)count * 6 > table_store_cap) {                                                                              //  no clearing; every table is written before it is read)
            table_store.reset(); table_store_cap = 0;
            table_store.reset(static_cast<DevHuff*>(malloc((size_t)count * 6 * sizeof(DevHuff))));
            if (!table_store) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: out of memory");
            table_store_cap = (size_t)count * 6;
        }
        for (int i = 0; i < count; ++i) prep[(size_t)i].huff = reinterpret_cast<DevHuff (*)[2]>(table_store.get() + (size_t)i * 6);
    }
    {
        std::vector<Parser*> parsers((size_t)workers, nullptr);
        for (Parser*& p : parsers) p = new Parser();
        parallel_for(count, workers, [&](int w, int i) { prepare_header(i, data[i], len[i], info[i], prep[(size_t)i], *parsers[(size_t)w]); });
        for (Parser* p : parsers) delete p;
    }
    std::vector<int> progressive;                              // SOF2 files: decoded by progressive_decode_device after the baseline ones
    for (int i = 0; i < count; ++i) if (prep[(size_t)i].progressive) { progressive.push_back(i); prep[(size_t)i].rc = kDeferred; }
    std::vector<char> tok((size_t)count, 0);
    std::vector<int64_t> tok_off((size_t)count, 0), strip_off((size_t)count, 0);
    std::vector<int32_t> tok_cap((size_t)count, 0);
    uint32_t* d_tokens = nullptr; uint32_t* d_strip_tab = nullptr;
    if (hooks && hooks->layout) {
        std::vector<int> rcs((size_t)count); std::vector<char> progs((size_t)count), tok_ok((size_t)count, 0);
        std::vector<size_t> scan_len((size_t)count, 0);
        for (int i = 0; i < count; ++i) {
            const FilePrep& fp = prep[(size_t)i];
            rcs[(size_t)i] = fp.rc == kDeferred || fp.rc == kHostRedo ? GAMUT_HIP_OK : fp.rc; progs[(size_t)i] = fp.progressive;
            tok_ok[(size_t)i] = fp.rc == GAMUT_HIP_OK && info[i].scan_type == GAMUT_JPGD_YH2V2 && fp.dev_unstuff && fp.items.size() == 1 && (uint32_t)fp.items[0].pad >= kSyncMinBytes;
            scan_len[(size_t)i] = fp.raw_len;
        }
        const TokenPlan plan{ tok_ok.data(), scan_len.data(), tok.data(), tok_off.data(), strip_off.data(), tok_cap.data(), &d_tokens, &d_strip_tab };
        if (int rc = hooks->layout(info, rcs.data(), progs.data(), own_offsets.data(), own_offsets.data() + count, &d_coeffs, &d_max_zag, plan)) return rc;
    }
    // B. serial: table de-duplication, slots of the files in the blob
    std::vector<DevImage> images((size_t)count);
    std::vector<DevHuff> huffs;
    std::vector<QuantTab> quants;
    std::vector<size_t> blob_off((size_t)count, 0);
    size_t blob_size = 0;
    for (int i = 0; i < count; ++i) {
        FilePrep& fp = prep[(size_t)i];
        DevImage& im = images[(size_t)i];
        memset(&im, 0, sizeof(im));
        if (fp.rc != GAMUT_HIP_OK) continue;
        im.coeff_off = coeff_offset[i]; im.zag_off = zag_offset[i]; im.nb = fp.nb; im.org = (int32_t)fp.org;
        if (tok[(size_t)i]) {                                  // strips = the reconstruction kernel's: 8 MCUs of 6 blocks along an MCU row
            im.tok = 1; im.tok_off = tok_off[(size_t)i]; im.strip_off = strip_off[(size_t)i]; im.tok_cap = tok_cap[(size_t)i];
            im.sb = 48; im.row_blocks = info[i].mcus_per_row * 6; im.n_strips = info[i].mcus_per_col * ((info[i].mcus_per_row + 7) / 8);
        }
        for (int c = 0; c < fp.comps; ++c) { im.quant[c] = intern(quants, fp.quant[c]); im.dc[c] = intern(huffs, fp.huff[c][0]); im.ac[c] = intern(huffs, fp.huff[c][1]); }
        blob_off[(size_t)i] = blob_size;
        blob_size += (fp.cap + 15) & ~(size_t)15;
    }
    const double ms_parse = ms_since(t_begin);
    double ms_upload = 0;
    std::vector<DevItem> items;
    int first_failure = GAMUT_HIP_OK; const char* first_msg = "";
    int n_long = 0;
    bool any_ok = false;
    for (int i = 0; i < count; ++i) any_ok = any_ok || prep[(size_t)i].rc == GAMUT_HIP_OK;
    uint32_t* d_scan_end = nullptr; uint32_t* st_used = nullptr;
    // The caller's status words start out clear whatever happens below: a batch in which NO baseline file is the kernels' (every one
    // of them kHostRedo at prepare_header, or progressive) never reaches the clear inside `any_ok`, and the file-level caller folds
    // these words into its verdicts -- stale memory there turned a file the host feeder had decoded into ERR_DECODE.
    if (d_status) GAMUT_HIP_CHECK(hipMemsetAsync(d_status, 0, (size_t)count * sizeof(uint32_t), stream));

    if (any_ok) {
        const auto t_up = std::chrono::steady_clock::now();
        static thread_local PerDevice<hipStream_t> copy_stream_pd;
        hipStream_t& copy_stream = copy_stream_pd.cur();
        if (!copy_stream) GAMUT_HIP_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        static thread_local PerDevice<DeviceScratch> scratch_pd, tab_scratch_pd;
        static thread_local PerDevice<PinnedScratch> pinned_pd, tab_pinned_pd;
        DeviceScratch& scratch = scratch_pd.cur(); DeviceScratch& tab_scratch = tab_scratch_pd.cur();
        PinnedScratch& pinned = pinned_pd.cur(); PinnedScratch& tab_pinned = tab_pinned_pd.cur();
        uint8_t* d_blob_w = (uint8_t*)scratch.get(blob_size + kBlobSlack, stream);
        uint8_t* h_blob = pinned.get(blob_size + kBlobSlack, copy_stream);
        if (!d_blob_w || !h_blob) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: staging allocation of %zu bytes failed", blob_size + kBlobSlack);
        // files whose scan goes up as it is (k_jpeg_unstuff): their slots in the raw image, pinned and device
        size_t raw_size = 0; int n_dev_files = 0; bool dev_restarts = false;
        std::vector<size_t> raw_off((size_t)count, 0);
        for (int i = 0; i < count; ++i) {
            const FilePrep& fp = prep[(size_t)i];
            if (fp.rc != GAMUT_HIP_OK) continue;
            dev_restarts = dev_restarts || fp.restart_interval > 0;        // (whoever unstuffs: what lies between an interval's last bit and its marker is the kernels' to judge)
            if (!fp.dev_unstuff) continue;
            raw_off[(size_t)i] = raw_size; raw_size += (fp.raw_len + 32 + 15) & ~(size_t)15; ++n_dev_files;
        }
        static thread_local PerDevice<DeviceScratch> raw_scratch_pd;
        static thread_local PerDevice<PinnedScratch> raw_pinned_pd;
        uint8_t* d_raw = nullptr; uint8_t* h_raw = nullptr;
        if (n_dev_files) {
            d_raw = (uint8_t*)raw_scratch_pd.cur().get(raw_size + 64, stream);
            h_raw = raw_pinned_pd.cur().get(raw_size + 64, copy_stream);
            if (!d_raw || !h_raw) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: staging allocation of %zu bytes failed", raw_size);
        }
        // (both kernels write every block of a segment as a whole line -- zeros included, blocks a damaged segment never reaches as
        // well: the coefficient buffer needs no clearing)
        uint32_t* st = d_status;
        if (!st) {                                             // the kernels want somewhere to flag errors
            static thread_local PerDevice<DeviceScratch> sink_pd;
            DeviceScratch& sink = sink_pd.cur();
            st = (uint32_t*)sink.get((size_t)count * sizeof(uint32_t), stream);
            if (!st) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: status allocation failed");
        }
        GAMUT_HIP_CHECK(hipMemsetAsync(st, 0, (size_t)count * sizeof(uint32_t), stream));
        static thread_local PerDevice<DeviceScratch> scan_end_pd;                 // per image: where k_jpeg_unstuff saw the scan end
        d_scan_end = (uint32_t*)scan_end_pd.cur().get((size_t)count * sizeof(uint32_t), stream);
        if (!d_scan_end) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: status allocation failed");
        GAMUT_HIP_CHECK(hipMemsetAsync(d_scan_end, 0xFF, (size_t)count * sizeof(uint32_t), stream));
        st_used = st;
        // C/D. groups of files.  A group's bytes are copied into pinned memory on the workers (as they are; files with tiny restart
        //      intervals: unstuffed there) and DMA'd at once; its segment list follows and the group's kernels -- unstuff, entropy
        //      decode, and through the hook the reconstruction -- are queued behind an event on a stream of the group's own, so the
        //      first group decodes while the rest is still on its way.
        constexpr int kMaxGroups = 8;
        static const int groups_env = [] { const char* e = getenv("GAMUT_HIP_JPEG_GROUPS"); return e && *e ? atoi(e) : 0; }();        // measurements
        const int n_groups = groups_env > 0 ? std::min(kMaxGroups, std::min(groups_env, count)) : (count < 128 ? 1 : std::max(2, std::min(kMaxGroups, count / 256)));        // (1024 files, a stream per group: 4 groups 14.8 ms, 8 groups 16.1 ms, 2 groups 14.9 ms -- profiles/r04_jpeg_sweep.txt; three streams in turn: 4 groups 12.7 / 13.1, 6 groups 13.1 / 13.4, 8 groups 13.2 / 13.0; 256 files: 1 group 4.5, 2 groups 4.3, 4 groups 5.0 ms)
        struct EvN { hipEvent_t e[kMaxGroups]; }; struct StN { hipStream_t s[kMaxGroups]; };
        static thread_local PerDevice<EvN> group_ready_pd;
        hipEvent_t (&group_ready)[kMaxGroups] = group_ready_pd.cur().e;
        for (int g = 0; g < n_groups; ++g) if (!group_ready[g]) GAMUT_HIP_CHECK(hipEventCreateWithFlags(&group_ready[g], hipEventDisableTiming));
        // A long segment is decoded by one workgroup in its own time (its sweeps), whatever else the chip does: the groups' kernels
        // are therefore queued on streams of their own (behind what `stream` holds now) and run side by side, each as soon as its
        // bytes are there -- on one stream a group waited for the group before it to finish.
        static thread_local PerDevice<StN> side_pd;
        static thread_local PerDevice<hipEvent_t> fork_pd;
        hipStream_t (&side)[kMaxGroups] = side_pd.cur().s;
        hipEvent_t& fork = fork_pd.cur();
        if (!fork) GAMUT_HIP_CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        // ... but not a stream per group: the runtime maps streams onto FOUR hardware queues (GPU_MAX_HW_QUEUES), a fifth stream shares one with
        // another, and its kernels wait for that stream's.  The timeline of round 4 (profiles/r04_jpeg_timeline.txt; rocprofv3 --kernel-trace
        // --memory-copy-trace): with `stream`, the copy stream and three side streams the LAST group's kernels started 4 ms after its bytes had
        // arrived, when the group before it had finished.  Three compute streams, taken in turn (a group's kernels take about as long as three
        // groups' uploads): 1024 files 14.0 / 13.6 -> 12.7 / 13.1 ms on one box (GAMUT_HIP_JPEG_STREAMS=5 is the old shape).  What is left is the
        // GPU's own time: side by side the groups' entropy kernels slow each other down (2.1 ms alone, 3.4 - 4.5 ms beside another one and a
        // reconstruction kernel) -- with four waves per SIMD the decode is bound by instruction issue, no longer by the length of a lane's chain.
        static const int lanes_env = [] { const char* e = getenv("GAMUT_HIP_JPEG_STREAMS"); return e && *e ? atoi(e) : 0; }();        // measurements
        const int n_lanes = std::max(1, std::min(n_groups, lanes_env > 0 ? std::min(lanes_env, kMaxGroups) : 3));
        for (int g = 1; g < n_lanes; ++g) if (!side[g]) GAMUT_HIP_CHECK(hipStreamCreateWithFlags(&side[g], hipStreamNonBlocking));
        GAMUT_HIP_CHECK(hipEventRecord(fork, stream));
        for (int g = 1; g < n_lanes; ++g) GAMUT_HIP_CHECK(hipStreamWaitEvent(side[g], fork, 0));
        auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
        size_t max_items = 0;
        for (int i = 0; i < count; ++i) if (prep[(size_t)i].rc == GAMUT_HIP_OK) max_items += prep[(size_t)i].restart_interval ? (size_t)(prep[(size_t)i].total_mcus / prep[(size_t)i].restart_interval + 1) : 1;
        // one small pinned image of the tables: [images][huff][quant][raw-file records][items of group 0][items of group 1]...
        const size_t o_img = 0, o_huff = align(o_img + images.size() * sizeof(DevImage)), o_quant = align(o_huff + huffs.size() * sizeof(DevHuff)),
                     o_raws = align(o_quant + quants.size() * sizeof(QuantTab)), o_items = align(o_raws + (size_t)n_dev_files * sizeof(DevRaw)),
                     total = o_items + align(max_items * sizeof(DevItem)) + 256 * (size_t)n_groups;
        uint8_t* d = (uint8_t*)tab_scratch.get(total, stream);
        uint8_t* h = tab_pinned.get(total, copy_stream);
        if (!d || !h) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: staging allocation of %zu bytes failed", total);
        memcpy(h + o_img, images.data(), images.size() * sizeof(DevImage));
        memcpy(h + o_huff, huffs.data(), huffs.size() * sizeof(DevHuff));
        memcpy(h + o_quant, quants.data(), quants.size() * sizeof(QuantTab));
        GAMUT_HIP_CHECK(hipMemcpyAsync(d, h, o_raws, hipMemcpyHostToDevice, copy_stream));
        const int n_huff = (int)huffs.size(), n_quant = (int)quants.size();
        const bool in_lds = n_huff <= kLdsHuff && n_quant <= kLdsQuant;
        const DevImage* d_img = (const DevImage*)(d + o_img);
        const DevHuff* d_huff = (const DevHuff*)(d + o_huff); const int16_t* d_quant = (const int16_t*)(d + o_quant);
        const uint8_t* d_blob = d_blob_w;
        size_t items_off = o_items;
        int raws_done = 0;
        int total_long = 0, total_short = 0;
        double ms_kernels_issue = 0;
        std::vector<DevRaw> raws;
        for (int g = 0; g < n_groups; ++g) {
            const int g_lo = (int)((int64_t)count * g / n_groups), g_hi = (int)((int64_t)count * (g + 1) / n_groups);
            if (g_hi <= g_lo) continue;
            parallel_for(g_hi - g_lo, workers, [&](int, int k) {
                const int i = g_lo + k;
                FilePrep& fp = prep[(size_t)i];
                if (fp.rc != GAMUT_HIP_OK) return;
                if (fp.dev_unstuff) memcpy(h_raw + raw_off[(size_t)i], data[i] + fp.scan_pos, fp.raw_len);
                else unstuff_file(i, data[i], len[i], info[i], fp, h_blob + blob_off[(size_t)i]);
            });
            {   // the group's uploads: one DMA per run of neighbouring slots of the same kind
                int i = g_lo;
                while (i < g_hi) {
                    const FilePrep& fp = prep[(size_t)i];
                    if (fp.rc != GAMUT_HIP_OK || fp.used == 0) { ++i; continue; }
                    const bool dev = fp.dev_unstuff;
                    int j = i; size_t b0 = dev ? raw_off[(size_t)i] : blob_off[(size_t)i], b1 = b0;
                    while (j < g_hi && (prep[(size_t)j].rc != GAMUT_HIP_OK || prep[(size_t)j].used == 0 || prep[(size_t)j].dev_unstuff == dev)) {
                        const FilePrep& fj = prep[(size_t)j];
                        if (fj.rc == GAMUT_HIP_OK && fj.used) b1 = dev ? raw_off[(size_t)j] + fj.raw_len : blob_off[(size_t)j] + fj.used;
                        ++j;
                    }
                    if (b1 > b0) GAMUT_HIP_CHECK(hipMemcpyAsync((dev ? d_raw : d_blob_w) + b0, (dev ? h_raw : h_blob) + b0, b1 - b0, hipMemcpyHostToDevice, copy_stream));
                    i = j;
                }
            }
            // the group's segment list: long segments get a workgroup each (self-synchronising decode), short ones a lane each
            items.clear();
            for (int i = g_lo; i < g_hi; ++i) {
                const FilePrep& fp = prep[(size_t)i];
                if (fp.rc != GAMUT_HIP_OK) continue;
                for (DevItem it : fp.items) { if (!fp.dev_unstuff) { it.begin += blob_off[(size_t)i]; it.end += blob_off[(size_t)i]; } items.push_back(it); }
            }
            if (items.empty()) continue;
            auto klass = [&](const DevItem& it) { return tok[(size_t)it.image] ? 0 : (uint32_t)it.pad >= kSyncMinBytes ? 1 : 2; };
            std::stable_sort(items.begin(), items.end(), [&](const DevItem& x, const DevItem& y) { return klass(x) < klass(y); });
            int n_tok = 0;
            while (n_tok < (int)items.size() && klass(items[(size_t)n_tok]) == 0) ++n_tok;
            n_long = n_tok;
            while (n_long < (int)items.size() && klass(items[(size_t)n_long]) == 1) ++n_long;
            const int n_items = (int)items.size(), n_short = n_items - n_long;
            total_long += n_long; total_short += n_short;
            if (items_off + items.size() * sizeof(DevItem) > total) return set_error(GAMUT_HIP_ERR_DECODE, "jpeg: more segments than the restart intervals allow");
            // the files of the group that are unstuffed on the device: where their segments stand in the list (a file's segments
            // are all of one class, the partition keeps their order)
            raws.clear();
            {
                int k = 0;
                while (k < n_items) {
                    const int img = items[(size_t)k].image; int e = k + 1;
                    while (e < n_items && items[(size_t)e].image == img) ++e;
                    const FilePrep& fp = prep[(size_t)img];
                    if (fp.dev_unstuff) {
                        DevRaw r; memset(&r, 0, sizeof(r));
                        r.raw_begin = raw_off[(size_t)img]; r.raw_len = fp.raw_len; r.out_begin = blob_off[(size_t)img]; r.out_cap = (fp.cap + 15) & ~(size_t)15;
                        r.image = img; r.first_item = k; r.n_items = e - k; r.restart_interval = fp.restart_interval;
                        raws.push_back(r);
                    }
                    k = e;
                }
            }
            memcpy(h + items_off, items.data(), items.size() * sizeof(DevItem));
            GAMUT_HIP_CHECK(hipMemcpyAsync(d + items_off, h + items_off, items.size() * sizeof(DevItem), hipMemcpyHostToDevice, copy_stream));
            DevItem* d_items = (DevItem*)(d + items_off);
            items_off = align(items_off + items.size() * sizeof(DevItem));
            const DevRaw* d_raws = (const DevRaw*)(d + o_raws) + raws_done;
            if (!raws.empty()) {
                if (raws_done + (int)raws.size() > n_dev_files) return set_error(GAMUT_HIP_ERR_DECODE, "jpeg: internal: raw-file records overflow");
                memcpy(h + o_raws + (size_t)raws_done * sizeof(DevRaw), raws.data(), raws.size() * sizeof(DevRaw));
                GAMUT_HIP_CHECK(hipMemcpyAsync(d + o_raws + (size_t)raws_done * sizeof(DevRaw), h + o_raws + (size_t)raws_done * sizeof(DevRaw), raws.size() * sizeof(DevRaw), hipMemcpyHostToDevice, copy_stream));
                raws_done += (int)raws.size();
            }
            GAMUT_HIP_CHECK(hipEventRecord(group_ready[g], copy_stream));
            const hipStream_t gs = g % n_lanes == 0 ? stream : side[g % n_lanes];
            GAMUT_HIP_CHECK(hipStreamWaitEvent(gs, group_ready[g], 0));
            if (trace) { (void)hipStreamSynchronize(gs); ms_upload = ms_since(t_up) - ms_kernels_issue; }
            const auto t_k = std::chrono::steady_clock::now();
            if (!raws.empty()) {
                hipLaunchKernelGGL(k_jpeg_unstuff, dim3((unsigned)raws.size()), dim3(kUnstuffThreads), 0, gs, d_raws, d_items, (const uint8_t*)d_raw, d_blob_w, st, d_scan_end);
                if (int rc = launch_status("jpeg_unstuff")) return rc;
            }
            static const int force_nh = [] { const char* e = getenv("GAMUT_HIP_JPEG_TABLES_LDS"); return e && *e ? atoi(e) : -1; }();     // measurements: 0 / 4 / 8
            const char* const ck_env = getenv("GAMUT_HIP_JPEG_CHECKPOINTS");                       // tests: checkpoints per lane of the counting passes (0 = the kernel's own)
            const int n_checkpoints = ck_env ? atoi(ck_env) : 0;
            const int nh_sel = force_nh == 0 ? 0 : (n_huff <= 4 && n_quant <= 4 && force_nh != 8) ? 4 : in_lds ? kLdsHuff : 0;
            if (n_tok) {                                       // long segments of the images that hand over tokens
#define GAMUT_SYNC_TOK(NH) hipLaunchKernelGGL((k_jpeg_entropy_sync<NH, true>), dim3(n_tok), dim3(kSyncThreads), 0, gs, d_items, d_img, d_huff, n_huff, d_quant, n_quant, d_blob, d_coeffs, d_max_zag, st, d_tokens, d_strip_tab, n_checkpoints)
                if (nh_sel == 4) GAMUT_SYNC_TOK(4); else if (nh_sel == kLdsHuff) GAMUT_SYNC_TOK(kLdsHuff); else GAMUT_SYNC_TOK(0);
#undef GAMUT_SYNC_TOK
                if (int rc = launch_status("jpeg_entropy_sync (tokens)")) return rc;
            }
            if (n_long > n_tok) {
#define GAMUT_SYNC_DENSE(NH) hipLaunchKernelGGL((k_jpeg_entropy_sync<NH, false>), dim3(n_long - n_tok), dim3(kSyncThreads), 0, gs, d_items + n_tok, d_img, d_huff, n_huff, d_quant, n_quant, d_blob, d_coeffs, d_max_zag, st, (uint32_t*)nullptr, (uint32_t*)nullptr, n_checkpoints)
                if (nh_sel == 4) GAMUT_SYNC_DENSE(4); else if (nh_sel == kLdsHuff) GAMUT_SYNC_DENSE(kLdsHuff); else GAMUT_SYNC_DENSE(0);
#undef GAMUT_SYNC_DENSE
                if (int rc = launch_status("jpeg_entropy_sync")) return rc;
            }
            if (n_short) {
                const dim3 grid((n_short + kEntropyThreads - 1) / kEntropyThreads), block(kEntropyThreads);
                if (in_lds) hipLaunchKernelGGL(k_jpeg_entropy<true>, grid, block, 0, gs, d_items + n_long, n_short, d_img, d_huff, n_huff, d_quant, n_quant, d_blob, d_coeffs, d_max_zag, st);
                else        hipLaunchKernelGGL(k_jpeg_entropy<false>, grid, block, 0, gs, d_items + n_long, n_short, d_img, d_huff, n_huff, d_quant, n_quant, d_blob, d_coeffs, d_max_zag, st);
            }
            if (int rc = launch_status("jpeg_entropy")) return rc;
            if (hooks && hooks->group_done) if (int rc = hooks->group_done(g_lo, g_hi, gs)) return rc;
            if (trace) { (void)hipStreamSynchronize(gs); ms_kernels_issue += ms_since(t_k); }
        }
        GAMUT_HIP_CHECK(hipStreamSynchronize(copy_stream));
        for (int g = 1; g < n_lanes; ++g) GAMUT_HIP_CHECK(hipStreamSynchronize(side[g]));
        GAMUT_HIP_CHECK(hipStreamSynchronize(stream));         // the per-thread staging buffers are reused by the next call
        // What the kernels could not vouch for goes to the host feeder below: any flag (a bit pattern no code word begins, a run past coefficient 63, a
        // segment that ran out, a wrong / missing RSTn, octets between an interval's last bit and its marker).  For the rest, find_eoi (:2826-2848):
        // the markers behind the scan are processed like those in front of it -- nearly always FF D9 and nothing to do.
        (void)dev_restarts;
        std::vector<uint32_t> flags((size_t)count), ends((size_t)count);
        GAMUT_HIP_CHECK(hipMemcpy(flags.data(), st, (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToHost));
        GAMUT_HIP_CHECK(hipMemcpy(ends.data(), d_scan_end, (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (int i = 0; i < count; ++i) {
            FilePrep& fp = prep[(size_t)i];
            if (fp.rc != GAMUT_HIP_OK) continue;
            if (flags[(size_t)i]) { fp.rc = kHostRedo; continue; }
            const size_t e = fp.dev_unstuff ? (ends[(size_t)i] == 0xFFFFFFFFu ? len[i] : fp.scan_pos + ends[(size_t)i]) : fp.scan_end;
            if (e >= len[i] || (e + 1 < len[i] && data[i][e] == 0xFF && data[i][e + 1] == 0xD9)) continue;        // the end of the file, or EOI at once
            std::unique_ptr<Parser> tp(new Parser());
            tp->data = data[i]; tp->len = len[i]; tp->pos = e;
            if (next_scan(*tp, &info[i], Walk::kTrailer) < 0) { fp.rc = GAMUT_HIP_ERR_DECODE; snprintf(fp.msg, sizeof(fp.msg), "image %d: %s", i, last_error_buf()); }
        }
        if (trace) fprintf(stderr, "[gamut_hip] jpeg_entropy_decode_device: %d files in %d group(s), %d long + %d short segments, %d+%d tables, %.1f MB compressed, %d host threads: headers %.1f ms, unstuff + upload %.1f ms, kernels %.1f ms (stages serialised by the trace)\n",
                           count, n_groups, total_long, total_short, n_huff, n_quant, blob_size / 1e6, workers, ms_parse, ms_upload, ms_kernels_issue);
    }
    {   // the host feeder's files (see kHostRedo)
        std::vector<int> redo;
        for (int i = 0; i < count; ++i) if (prep[(size_t)i].rc == kHostRedo) redo.push_back(i);
        if (!redo.empty()) {
            std::vector<int> rcs; std::vector<std::string> msgs;
            const auto deliver = hooks && hooks->redo ? std::function<int(int, const gamut_hip_jpeg_frame&)>([&](int i, const gamut_hip_jpeg_frame& fr) {
                                     uint32_t* const stw = st_used ? st_used : d_status;
                                     if (stw && hipMemsetAsync(stw + i, 0, sizeof(uint32_t), stream) != hipSuccess) return set_error(GAMUT_HIP_ERR_HIP, "jpeg: status reset failed");
                                     return hooks->redo(i, fr, stream); })
                                                      : deliver_dense(coeff_offset, zag_offset, d_coeffs, d_max_zag, st_used ? st_used : d_status, stream);
            if (int rc = host_redo(data, len, redo, info, deliver, rcs, msgs)) return rc;
            for (size_t k = 0; k < redo.size(); ++k) {
                FilePrep& fp = prep[(size_t)redo[k]];
                fp.rc = rcs[k];
                if (rcs[k] != GAMUT_HIP_OK) snprintf(fp.msg, sizeof(fp.msg), "image %d: %s", redo[k], msgs[k].c_str());
            }
        }
    }
    int first_index = -1;
    for (int i = 0; i < count; ++i) {
        const FilePrep& fp = prep[(size_t)i];
        if (fp.rc == kDeferred) continue;
        if (host_status) host_status[i] = fp.rc;
        if (fp.rc != GAMUT_HIP_OK && first_failure == GAMUT_HIP_OK) { first_failure = fp.rc; first_msg = fp.msg; first_index = i; }
    }
    // the progressive files of the batch, into the same buffers
    char prog_msg[256] = { 0 };
    if (!progressive.empty()) {
        int prog_first = -1;
        const int rc = progressive_decode_device(data, len, progressive, coeff_offset, zag_offset, d_coeffs, d_max_zag, d_status, info, host_status, stream,
                                                 &prog_first, prog_msg, sizeof(prog_msg));
        if (rc != GAMUT_HIP_OK && prog_first < 0) return rc;                              // an allocation / HIP failure, message set
        if (rc != GAMUT_HIP_OK && (first_index < 0 || prog_first < first_index)) { first_failure = rc; first_msg = prog_msg; first_index = prog_first; }
    }
    if (first_failure != GAMUT_HIP_OK) return set_error(first_failure, "%s", first_msg);
    return GAMUT_HIP_OK;
}

} // namespace
} // namespace gamut

using namespace gamut;

#if JPEG_SYNC_PROFILE
extern "C" int gamut_hip_jpeg_sync_profile(unsigned long long* out8, int reset)      // measurement builds only
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_sync_prof), 128) != hipSuccess) return 1;          // (sixteen counters)
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_sync_prof), z, 128) != hipSuccess) return 1; }
    return 0;
}
#endif
extern "C" {

int gamut_hip_jpeg_decode_coeffs(const uint8_t* data, size_t len, gamut_hip_jpeg_frame* out)
{
    clear_error();
    if (!out) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_decode_coeffs: null frame");
    return decode_coeffs(data, len, out);
}

int gamut_hip_jpeg_decode_coeffs_batch(const uint8_t* const* data, const size_t* len, int count,
                                       gamut_hip_jpeg_frame* out, int* status, int threads)
{
    clear_error();
    if (count < 0 || (count > 0 && (!data || !len || !out)))
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_decode_coeffs_batch: bad arguments");
    if (threads <= 0) threads = host_threads();
    if (threads < 1) threads = 1;
    if (threads > count) threads = count;
    try {
    // images are independent: workers pull the next index; every worker keeps the message of its lowest failing image
    std::atomic<int> next{ 0 };
    struct Failure { int index = INT32_MAX, code = GAMUT_HIP_OK; char msg[256] = { 0 }; };
    std::vector<Failure> fails((size_t)(threads > 0 ? threads : 1));
    auto work = [&](int tid) {
        for (int i; (i = next.fetch_add(1, std::memory_order_relaxed)) < count; ) {
            const int rc = decode_coeffs(data[i], len[i], &out[i]);
            if (status) status[i] = rc;
            if (rc != GAMUT_HIP_OK && i < fails[tid].index) {
                fails[tid].index = i; fails[tid].code = rc;
                snprintf(fails[tid].msg, sizeof(fails[tid].msg), "%s", last_error_buf());       // this worker thread's message
            }
        }
    };
    if (threads <= 1) { if (count > 0) work(0); }
    else {
        std::vector<std::thread> pool;
        pool.reserve((size_t)threads - 1);
        try {                                                  // thread creation may fail (EAGAIN): the rest runs on fewer workers
            for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
        } catch (...) {}
        work(0);
        for (std::thread& th : pool) th.join();
    }
    const Failure* first = nullptr;
    for (const Failure& fl : fails) if (fl.code != GAMUT_HIP_OK && (!first || fl.index < first->index)) first = &fl;
    if (!first) { clear_error(); return GAMUT_HIP_OK; }
    return set_error(first->code, "image %d: %s", first->index, first->msg);
    } catch (...) {
        return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg_decode_coeffs_batch: out of host memory");
    }
}

int gamut_hip_jpeg_read_header(const uint8_t* data, size_t len, gamut_hip_jpeg_frame* out)
{
    clear_error();
    if (!out) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_read_header: null frame");
    Parser* ps = new (std::nothrow) Parser();
    if (!ps) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: out of memory");
    const int rc = parse_baseline(*ps, data, len, out, false);
    delete ps;
    return rc;
}

int gamut_hip_jpeg_scan_layout(const uint8_t* data, size_t len, gamut_hip_jpeg_frame* info, int32_t* segments, uint64_t* entropy_bytes)
{
    clear_error();
    if (!info) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_scan_layout: null frame");
    try {
        Parser* ps = new Parser();
        FilePrep fp;
        std::vector<DevHuff> fp_tables(6);
        fp.huff = reinterpret_cast<DevHuff (*)[2]>(fp_tables.data());
        prepare_header(0, data, len, *info, fp, *ps);
        delete ps;
        if (fp.progressive) {                                  // every scan of the file, cut at its restart markers (jpeg_prog.hpp)
            Parser* pp = new Parser();
            ProgPrep pr;
            prog_prepare(0, data, len, *info, pr, *pp);
            delete pp;
            std::vector<uint8_t> bytes(pr.rc == GAMUT_HIP_OK ? pr.cap : 0);
            if (pr.rc == GAMUT_HIP_OK) prog_unstuff(0, data, *info, pr, bytes.data());
            if (segments) *segments = (int32_t)pr.items.size();
            if (entropy_bytes) *entropy_bytes = pr.used;
            if (pr.rc != GAMUT_HIP_OK) return set_error(pr.rc, "%s", pr.msg);
            return GAMUT_HIP_OK;
        }
        std::vector<uint8_t> bytes(fp.rc == GAMUT_HIP_OK ? fp.cap : 0);
        if (fp.rc == GAMUT_HIP_OK) { fp.items.clear(); fp.dev_unstuff = false; unstuff_file(0, data, len, *info, fp, bytes.data()); }   // (the host's walk: exact sizes)
        if (segments) *segments = (int32_t)fp.items.size();
        if (entropy_bytes) *entropy_bytes = fp.used;
        if (fp.rc != GAMUT_HIP_OK) return set_error(fp.rc, "%s", fp.msg);
        return GAMUT_HIP_OK;
    } catch (...) {
        return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg_scan_layout: out of host memory");
    }
}

int gamut_hip_jpeg_entropy_decode_device(const uint8_t* const* data, const size_t* len, int count,
                                         const int64_t* coeff_offset, const int64_t* zag_offset,
                                         int16_t* coeffs, uint8_t* max_zag, uint32_t* status_dev,
                                         gamut_hip_jpeg_frame* info, int* status_host, void* stream)
{
    clear_error();
    if (count < 0 || (count > 0 && (!data || !len || !coeff_offset || !zag_offset || !coeffs || !max_zag || !info)))
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_entropy_decode_device: bad arguments");
    if (count == 0) return GAMUT_HIP_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_error(GAMUT_HIP_ERR_NO_DEVICE, "no HIP device available (libgamut_hip has no CPU fallback)");
    try {                                                      // std::vector / bad_alloc must not escape a C entry point
        return entropy_decode_device(data, len, count, coeff_offset, zag_offset, coeffs, max_zag, status_dev, info, status_host, pick_stream(stream));
    } catch (...) {
        return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg_entropy_decode_device: out of host memory");
    }
}

int gamut_hip_jpeg_decode_batch_device(const uint8_t* const* data, const size_t* len, int count, int req_comps,
                                       const int64_t* out_offset, uint8_t* out, gamut_hip_jpeg_frame* info, int* status_host,
                                       uint32_t* status_dev, void* stream)
{
    clear_error();
    if (count < 0 || (req_comps != 1 && req_comps != 3 && req_comps != 4) || (count > 0 && (!data || !len || !out_offset || !out || !info)))
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_decode_batch_device: bad arguments");
    if (count == 0) return GAMUT_HIP_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_error(GAMUT_HIP_ERR_NO_DEVICE, "no HIP device available (libgamut_hip has no CPU fallback)");
    try {
        hipStream_t st = pick_stream(stream);
        static thread_local PerDevice<DeviceScratch> co_pd, zz_pd, tok_pd, strip_pd, offs_pd;
        DeviceScratch& s_co = co_pd.cur(); DeviceScratch& s_zz = zz_pd.cur();
        std::vector<int64_t> co_off((size_t)count, 0), zz_off((size_t)count, 0), tk_off((size_t)count, 0), sp_off((size_t)count, 0);
        std::vector<char> ok((size_t)count, 0), prog((size_t)count, 0), tokm((size_t)count, 0);
        int16_t* d_co = nullptr; uint8_t* d_zz = nullptr;
        uint32_t* d_tok = nullptr; uint32_t* d_strip = nullptr; int64_t* d_offs = nullptr;      // d_offs: [count] token offsets, [count] strip-table offsets
        // the hand-off between the entropy kernels and the reconstruction: tokens (the default where an image can: 4:2:0, one long
        // segment) or the dense 128-byte blocks of the public coefficient-level entry point; GAMUT_HIP_JPEG_HANDOFF=dense forces the latter
        const char* ho_env = getenv("GAMUT_HIP_JPEG_HANDOFF");
        const bool want_tokens = !(ho_env && !strcmp(ho_env, "dense"));
        DecodeHooks hooks;
        hooks.layout = [&](const gamut_hip_jpeg_frame* f, const int* rc, const char* progressive, int64_t* coeff_offset, int64_t* zag_offset, int16_t** pco, uint8_t** pzz,
                           const TokenPlan& plan) -> int {
            int64_t blocks = 0, zblocks = 0, tokens = 0, strips = 0;
            for (int i = 0; i < count; ++i) {
                coeff_offset[i] = blocks * 64; zag_offset[i] = zblocks;
                co_off[(size_t)i] = blocks * 64; zz_off[(size_t)i] = zblocks;
                ok[(size_t)i] = rc[i] == GAMUT_HIP_OK; prog[(size_t)i] = progressive[i];
                if (!ok[(size_t)i]) continue;
                const int64_t nblk = (int64_t)f[i].mcus_per_row * f[i].mcus_per_col * f[i].blocks_per_mcu;
                zblocks += nblk;
                // a token per DC and per non-zero AC coefficient: at most 64 per block, and no more than the scan has bit pairs plus a
                // DC token per block (a token costs two bits of the scan at least, a DC token of a block that ends at once one)
                const int64_t cap = std::min<int64_t>(nblk * 64, (int64_t)plan.scan_len[i] * 4 + nblk) + 16;
                // (the token kernel stores whole rgba8 pixels as dwords: an image whose rows would not be dword-aligned keeps the dense hand-off, whose
                //  launch falls back to the byte-wise kernel -- jpeg_reconstruct_launch)
                const bool dword_rows = req_comps != 4 || ((((uintptr_t)out + (uint64_t)out_offset[i]) & 3) == 0);
                if (want_tokens && plan.tok_ok[i] && cap < 0x7fffffff && dword_rows) {
                    tokm[(size_t)i] = 1; plan.tok[i] = 1;
                    tk_off[(size_t)i] = tokens; plan.tok_off[i] = tokens; plan.tok_cap[i] = (int32_t)cap;
                    sp_off[(size_t)i] = strips; plan.strip_off[i] = strips;
                    tokens += (cap + 3) & ~(int64_t)3;
                    strips += (int64_t)f[i].mcus_per_col * ((f[i].mcus_per_row + 7) / 8) + 1;
                } else blocks += nblk;
            }
            d_co = (int16_t*)s_co.get((size_t)blocks * 128 + 256, st); d_zz = (uint8_t*)s_zz.get((size_t)zblocks + 256, st);
            if (!d_co || !d_zz) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg_decode_batch_device: %lld coefficient blocks do not fit the device", (long long)blocks);
            *pco = d_co; *pzz = d_zz;
            if (tokens) {
                d_tok = (uint32_t*)tok_pd.cur().get((size_t)tokens * 4 + 256, st);
                d_strip = (uint32_t*)strip_pd.cur().get((size_t)strips * 4 + 256, st);
                d_offs = (int64_t*)offs_pd.cur().get((size_t)count * 16 + 256, st);
                if (!d_tok || !d_strip || !d_offs) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg_decode_batch_device: %lld tokens do not fit the device", (long long)tokens);
                *plan.d_tokens = d_tok; *plan.d_strip_tab = d_strip;
                // (pageable sources: the runtime has read them when the calls return; the kernels that use the table are queued later)
                GAMUT_HIP_CHECK(hipMemcpyAsync(d_offs, tk_off.data(), (size_t)count * 8, hipMemcpyHostToDevice, st));
                GAMUT_HIP_CHECK(hipMemcpyAsync(d_offs + count, sp_off.data(), (size_t)count * 8, hipMemcpyHostToDevice, st));
            }
            return GAMUT_HIP_OK;
        };
        // images [lo, hi) -> pixels: runs of images of one geometry (and one kind of hand-off) at even strides go out as one batched launch
        auto reconstruct = [&](int lo, int hi, hipStream_t gs, bool progressive_only) -> int {
            for (int i = lo; i < hi; ) {
                const gamut_hip_jpeg_frame& f = info[i];
                const bool mine = ok[(size_t)i] && (prog[(size_t)i] != 0) == progressive_only;
                if (!mine) { ++i; continue; }
                const int64_t nblk = (int64_t)f.mcus_per_row * f.mcus_per_col * f.blocks_per_mcu;
                const bool tk = tokm[(size_t)i] != 0;
                int j = i + 1;
                const int64_t ostride = j < hi ? out_offset[j] - out_offset[i] : 0;
                while (j < hi && ok[(size_t)j] && (prog[(size_t)j] != 0) == progressive_only && (tokm[(size_t)j] != 0) == tk && info[j].width == f.width && info[j].height == f.height &&
                       info[j].scan_type == f.scan_type && (tk || co_off[(size_t)j] - co_off[(size_t)j - 1] == nblk * 64) && zz_off[(size_t)j] - zz_off[(size_t)j - 1] == nblk &&
                       out_offset[j] - out_offset[j - 1] == ostride && ostride > 0) ++j;
                if (tk) {
                    if (int rc = jpeg_reconstruct_tokens_launch(d_tok, d_strip, d_offs + i, d_offs + count + i, d_zz + zz_off[(size_t)i], nblk, out + out_offset[i],
                                                                (int64_t)f.width * req_comps, j - i > 1 ? ostride : 0, f.width, f.height, req_comps, j - i, gs)) return rc;
                } else
                if (int rc = jpeg_reconstruct_launch(d_co + co_off[(size_t)i], nblk * 64, d_zz + zz_off[(size_t)i], nblk, out + out_offset[i], (int64_t)f.width * req_comps,
                                                     j - i > 1 ? ostride : 0, f.width, f.height, f.scan_type, req_comps, j - i, gs)) return rc;
                i = j;
            }
            return GAMUT_HIP_OK;
        };
        hooks.group_done = [&](int lo, int hi, hipStream_t gs) -> int { return reconstruct(lo, hi, gs, false); };
        // a baseline file the host feeder decoded behind the device pass (kHostRedo): its coefficients go up and through the dense reconstruction
        hooks.redo = [&](int i, const gamut_hip_jpeg_frame& fr, hipStream_t s) -> int {
            static thread_local PerDevice<DeviceScratch> rco_pd, rzz_pd;
            const size_t nblk = (size_t)fr.mcus_per_row * fr.mcus_per_col * fr.blocks_per_mcu;
            int16_t* c = (int16_t*)rco_pd.cur().get(nblk * 128 + 256, s); uint8_t* z = (uint8_t*)rzz_pd.cur().get(nblk + 256, s);
            if (!c || !z) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg_decode_batch_device: coefficient scratch of image %d", i);
            if (hipMemcpyAsync(c, fr.coeffs, nblk * 128, hipMemcpyHostToDevice, s) != hipSuccess || hipMemcpyAsync(z, fr.max_zag, nblk, hipMemcpyHostToDevice, s) != hipSuccess)
                return set_error(GAMUT_HIP_ERR_HIP, "jpeg_decode_batch_device: upload of the coefficients of image %d failed", i);
            if (int rc = jpeg_reconstruct_launch(c, (int64_t)nblk * 64, z, (int64_t)nblk, out + out_offset[i], (int64_t)fr.width * req_comps, 0, fr.width, fr.height, fr.scan_type,
                                                 req_comps, 1, s)) return rc;
            if (hipStreamSynchronize(s) != hipSuccess) return set_error(GAMUT_HIP_ERR_HIP, "jpeg_decode_batch_device: reconstruction of image %d failed", i);
            return GAMUT_HIP_OK;
        };
        // per-file verdicts: the header-level ones (host) and what the kernels flag (device) are folded into ONE status per file, as
        // gamut_hip_png_decode_batch_device does -- a caller that passes no status arrays still gets the failure as the return value
        std::vector<int> own_host; int* hst = status_host;
        if (!hst) { own_host.assign((size_t)count, GAMUT_HIP_OK); hst = own_host.data(); }
        static thread_local PerDevice<DeviceScratch> st_pd;
        uint32_t* d_st = status_dev;
        if (!d_st) { d_st = (uint32_t*)st_pd.cur().get((size_t)count * sizeof(uint32_t), st); if (!d_st) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg_decode_batch_device: status allocation failed"); }
        int rc = entropy_decode_device(data, len, count, nullptr, nullptr, nullptr, nullptr, d_st, info, hst, st, &hooks);
        if (rc != GAMUT_HIP_OK && rc != GAMUT_HIP_ERR_DECODE && rc != GAMUT_HIP_ERR_UNSUPPORTED) return rc;     // not a per-file verdict: allocation / HIP failure
        char first_msg[200]; snprintf(first_msg, sizeof(first_msg), "%s", last_error_buf());
        // (entropy_decode_device returns when every stream it used has drained: the baseline files' pixels are in place.  The
        // progressive files' coefficients were decoded after the groups: their pixels follow here -- not those of a file whose scans
        // could not be prepared: its share of the coefficient scratch holds whatever an earlier batch left there.)
        if (d_co) {
            for (int i = 0; i < count; ++i) if (prog[(size_t)i] && hst[i] != GAMUT_HIP_OK) ok[(size_t)i] = 0;
            if (int rc2 = reconstruct(0, count, st, true)) return rc2;
            std::vector<uint32_t> flags((size_t)count);
            GAMUT_HIP_CHECK(hipMemcpyAsync(flags.data(), d_st, (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            GAMUT_HIP_CHECK(hipStreamSynchronize(st));          // the per-thread coefficient scratch is reused by the next call
            int first = -1;
            for (int i = 0; i < count; ++i) {
                if (hst[i] == GAMUT_HIP_OK && flags[(size_t)i]) {       // a damaged entropy-coded segment: the file's pixels are what the decode of the damage gives
                    hst[i] = GAMUT_HIP_ERR_DECODE;
                    if (first < 0) first = i;
                }
            }
            int lowest = -1;
            for (int i = 0; i < count; ++i) if (hst[i] != GAMUT_HIP_OK) { lowest = i; break; }
            if (lowest >= 0 && lowest == first) return set_error(GAMUT_HIP_ERR_DECODE, "image %d: corrupt entropy-coded data", first);
            if (lowest >= 0) return set_error(hst[lowest], "%s", first_msg[0] ? first_msg : "jpeg_decode_batch_device: a file failed");
        }
        if (rc != GAMUT_HIP_OK) return set_error(rc, "%s", first_msg);
        return GAMUT_HIP_OK;
    } catch (...) {
        return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg_decode_batch_device: out of host memory");
    }
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
    // device staging of the call: per-thread buffers that grow and stay (a hipMalloc / hipFree pair per buffer and call cost more than
    // the decode of a small image)
    static thread_local PerDevice<DeviceScratch> s_co_pd, s_zz_pd, s_out_pd;
    DeviceScratch& s_co = s_co_pd.cur(); DeviceScratch& s_zz = s_zz_pd.cur(); DeviceScratch& s_out = s_out_pd.cur();
    if (ok) {
        dco = s_co.get(nblk * 128 + 16); dzz = s_zz.get(nblk + 16); dout = s_out.get(out_bytes + 16);
        if (!dco || !dzz || !dout) { (void)hipGetLastError(); set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "decompress_jpeg: device staging allocation failed"); ok = false; }
    }
    ok = ok && hip_ok(hipMemcpyAsync(dco, f.coeffs, nblk * 128, hipMemcpyHostToDevice, st), "hipMemcpyAsync") &&
         hip_ok(hipMemcpyAsync(dzz, f.max_zag, nblk, hipMemcpyHostToDevice, st), "hipMemcpyAsync");
    ok = ok && jpeg_reconstruct_launch((const int16_t*)dco, 0, (const uint8_t*)dzz, 0, (uint8_t*)dout, (int64_t)dst_bpl, 0,
                                       f.width, f.height, f.scan_type, req_comps, 1, st) == GAMUT_HIP_OK;
    ok = ok && hip_ok(hipMemcpyAsync(result, dout, out_bytes, hipMemcpyDeviceToHost, st), "hipMemcpyAsync") &&
         hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize");
    if (ok) {
        if (pixelAspectRatio) *pixelAspectRatio = f.pixel_aspect_ratio;                     // :3804-3805
        if (dotsPerInchY) *dotsPerInchY = f.dpi_y;
    } else { free(result); result = nullptr; }
    gamut_hip_jpeg_frame_free(&f);
    return result;
}

} // extern "C"
