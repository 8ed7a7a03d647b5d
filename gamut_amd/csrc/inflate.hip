// inflate.hip -- DEFLATE (RFC 1951) on the GPU, for batches of PNG files.
//
// Replaces, for a batch, stbi_zlib_decode_malloc_guesssize_headerflag (stbdec.d:1267-1321 -> the `miniz` inflate): IDAT
// streams resident in HBM are inflated straight into the arena the de-filter kernels read (png.hip), so a file batch no
// longer waits for sixteen host threads running zlib.  One 256-thread workgroup per stream; inside a stream:
//
//   * block headers and the code-length alphabet of a dynamic block are read by one thread (a few hundred bits); the two
//     canonical Huffman codes are turned into lookup tables by all threads (counting by LDS atomics, ranks, one table entry
//     per thread and pass);
//   * the block's symbols are decoded 4 KiB at a time, SPECULATIVELY: lane k starts at bit 128 k of the chunk as if a
//     token began there, lanes then restart from their predecessor's exit position until nothing moves -- Huffman streams
//     re-synchronise within a few tokens, so two or three sweeps settle all 256 lanes (the chain is exact from lane 0 on,
//     and a lane that meets the end-of-block code, an invalid code or its output cap ends the chunk there);
//   * prefix sums of the lanes' output bytes / match counts place everything: a last sweep writes the literals into a
//     64 KiB ring in LDS (the 32 KiB window plus the chunk's output) and lists the matches;
//   * matches are resolved 256 at a time in rounds: a match copies as soon as its source lies below the first byte that is
//     still pending (overlapping copies, distance < length, are byte-serial inside their lane);
//   * the chunk's bytes leave the ring for HBM in dwords.
// No data-dependent branch is taken on a whole stream's behalf by a single lane except the block headers.
#include "common.hpp"

namespace gamut {
namespace {

constexpr int kT = 256;                              // threads per stream
constexpr int kSubBits = 128;                        // compressed bits a lane owns per chunk
constexpr int kChunkBytes = kT * kSubBits / 8;       // 4096
constexpr int kWinDwords = kChunkBytes / 4 + 8;      // + 32 bytes: the last lane runs up to 47 bits past its end and peeks 64 bits from there
constexpr int kRing = 65536, kRingMask = kRing - 1;
constexpr int kNewMax = 32768 - 1024;                // bytes a chunk may add to the ring: the 32 KiB history must survive them
constexpr int kLaneOutMax = 8192;                    // a lane stops early beyond this (ends the chunk: pathological match runs)
constexpr int kMatchCap = 3072;
constexpr int kLitBits = 10, kDistBits = 9;          // primary lookup widths; longer codes take the canonical search

enum : uint32_t { F_EOB = 1, F_BAD = 2, F_EARLY = 4 };
// status word per stream (0 = ok)
enum : uint32_t { E_BLOCK_TYPE = 1, E_STORED = 2, E_LENGTHS = 3, E_CODE = 4, E_DISTANCE = 5, E_INPUT = 6 };
enum { C_FIRST_BAD = 0, C_FIRST_STOP, C_CUT, C_HWM, C_ERR, C_BTYPE, C_FINAL, C_HDR_END_LO, C_HDR_END_HI, C_HLIT, C_HDIST, C_N };

struct InfItem { const uint8_t* src; uint8_t* dst; uint32_t src_len, dst_cap; };

struct Canon { uint32_t first[16], count[16], offs[16]; };

struct Shared {
    __attribute__((aligned(16))) uint8_t ring[kRing];
    __attribute__((aligned(16))) uint32_t win[kWinDwords];
    uint32_t lit_lut[1 << kLitBits];
    uint32_t dist_lut[1 << kDistBits];
    uint2    matches[kMatchCap];                     // x = destination (absolute output offset), y = length | distance << 16
    uint32_t exit_bit[kT], flags[kT];
    uint32_t scan_a[kT], scan_b[kT];
    uint32_t lit_sorted[288], dist_sorted[32];       // symbol payloads in canonical order
    Canon    lit, dist;
    uint32_t cl_lut[128];
    uint32_t ctrl[C_N];
    uint8_t  lens[320];
};

// 64 stream bits from bit `bit` of the window (LSB first, as DEFLATE packs them)
__device__ __forceinline__ uint64_t peek64(const uint32_t* win, uint32_t bit)
{
    const uint32_t w = bit >> 5, sh = bit & 31u;
    const uint32_t a = win[w], b = win[w + 1], c = win[w + 2];
    return (uint64_t)__builtin_amdgcn_alignbit(c, b, sh) << 32 | __builtin_amdgcn_alignbit(b, a, sh);
}

// symbol -> table payload: bits 4-6 kind (0 literal, 1 length / distance, 2 end of block, 3 invalid), 8-12 extra bits, 16-31 base
__device__ __forceinline__ uint32_t lit_payload(uint32_t s)
{
    if (s < 256u) return s << 16;
    if (s == 256u) return 2u << 4;
    if (s > 285u) return 3u << 4;
    const uint32_t i = s - 257u;
    if (i < 8u) return (3u + i) << 16 | 1u << 4;
    if (i == 28u) return 258u << 16 | 1u << 4;
    const uint32_t x = (i - 4u) >> 2;
    return (3u + ((4u + (i & 3u)) << x)) << 16 | x << 8 | 1u << 4;
}
__device__ __forceinline__ uint32_t dist_payload(uint32_t d)
{
    if (d > 29u) return 3u << 4;
    if (d < 4u) return (1u + d) << 16 | 1u << 4;
    const uint32_t x = (d - 2u) >> 1;
    return (1u + ((2u + (d & 1u)) << x)) << 16 | x << 8 | 1u << 4;
}

// canonical search over code lengths [from, to]: `bits` holds the stream bits LSB first; 0 = no code of these lengths matches
__device__ __forceinline__ uint32_t canon_decode(uint32_t bits, const Canon& c, const uint32_t* sorted, int from, int to)
{
    const uint32_t rev = __brev(bits);
    for (int l = from; l <= to; ++l) {
        const uint32_t d = (rev >> (32 - l)) - c.first[l];
        if (d < c.count[l]) return sorted[c.offs[l] + d] | (uint32_t)l;
    }
    return 0;
}

// inclusive prefix sums of two values over the workgroup
__device__ __forceinline__ void block_scan2(Shared& S, uint32_t& a, uint32_t& b)
{
    const int t = threadIdx.x;
    S.scan_a[t] = a; S.scan_b[t] = b;
    __syncthreads();
    for (int d = 1; d < kT; d <<= 1) {
        const uint32_t pa = t >= d ? S.scan_a[t - d] : 0u, pb = t >= d ? S.scan_b[t - d] : 0u;
        __syncthreads();
        a += pa; b += pb;
        S.scan_a[t] = a; S.scan_b[t] = b;
        __syncthreads();
    }
}

// canonical code of `n` symbols with lengths S.lens[base .. base + n): Canon, payloads in canonical order, primary table.
// Validity as zlib's inflate_table: over-subscribed sets fail; incomplete ones too, unless the set is a single 1-bit code.
template <bool DIST>
__device__ void build_table(Shared& S, int base, int n)
{
    const int t = threadIdx.x;
    Canon& c = DIST ? S.dist : S.lit;
    uint32_t* sorted = DIST ? S.dist_sorted : S.lit_sorted;
    uint32_t* lut = DIST ? S.dist_lut : S.lit_lut;
    constexpr int P = DIST ? kDistBits : kLitBits;
    if (t < 16) c.count[t] = 0;
    __syncthreads();
    for (int s = t; s < n; s += kT) { const uint32_t L = S.lens[base + s]; if (L) atomicAdd(&c.count[L], 1u); }
    __syncthreads();
    if (t == 0) {
        int left = 1, maxlen = 0; bool over = false;
        for (int l = 1; l <= 15; ++l) { left = left * 2 - (int)c.count[l]; if (left < 0) over = true; if (c.count[l]) maxlen = l; }
        if (over || (left > 0 && maxlen > 1)) atomicMax(&S.ctrl[C_ERR], (uint32_t)E_LENGTHS);
        c.first[0] = 0; c.offs[0] = 0; c.first[1] = 0; c.offs[1] = 0;
        for (int l = 2; l <= 15; ++l) { c.first[l] = (c.first[l - 1] + c.count[l - 1]) << 1; c.offs[l] = c.offs[l - 1] + c.count[l - 1]; }
    }
    __syncthreads();
    for (int s = t; s < n; s += kT) {
        const uint32_t L = S.lens[base + s];
        if (!L) continue;
        uint32_t rank = 0;
        for (int k = 0; k < s; ++k) rank += S.lens[base + k] == L;
        sorted[c.offs[L] + rank] = DIST ? dist_payload((uint32_t)s) : lit_payload((uint32_t)s);
    }
    __syncthreads();
    for (int e = t; e < (1 << P); e += kT) lut[e] = canon_decode((uint32_t)e, c, sorted, 1, P);
    __syncthreads();
}

struct LaneResult { uint32_t exit, flags, out, nm; };

// tokens of one lane: from bit `start` until a token begins at or beyond `end` (window-relative bits).  WRITE: literals into the
// ring at absolute output offset obase.., matches into S.matches[mbase..]
template <bool WRITE>
__device__ __forceinline__ LaneResult lane_decode(Shared& S, uint32_t start, uint32_t end, uint32_t obase, uint32_t mbase)
{
    uint32_t pos = start, o = 0, nm = 0, fl = 0;
    while (pos < end && !fl) {
        uint64_t bits = peek64(S.win, pos);
        uint32_t used = 0;
        do {
            uint32_t e = S.lit_lut[(uint32_t)bits & ((1u << kLitBits) - 1u)];
            if ((e & 15u) == 0u) { e = canon_decode((uint32_t)bits, S.lit, S.lit_sorted, kLitBits + 1, 15); if (!e) { fl = F_BAD; break; } }
            uint32_t nb = e & 15u;
            bits >>= nb; used += nb;
            const uint32_t kind = (e >> 4) & 7u;
            if (kind == 0u) {
                if (WRITE) S.ring[(obase + o) & kRingMask] = (uint8_t)(e >> 16);
                ++o;
            } else if (kind == 1u) {
                uint32_t xb = (e >> 8) & 31u;
                const uint32_t len = (e >> 16) + ((uint32_t)bits & ((1u << xb) - 1u));
                bits >>= xb; used += xb;
                uint32_t de = S.dist_lut[(uint32_t)bits & ((1u << kDistBits) - 1u)];
                if ((de & 15u) == 0u) { de = canon_decode((uint32_t)bits, S.dist, S.dist_sorted, kDistBits + 1, 15); if (!de) { fl = F_BAD; break; } }
                if (((de >> 4) & 7u) != 1u) { fl = F_BAD; break; }                 // distance symbols 30 / 31
                nb = de & 15u;
                bits >>= nb; used += nb;
                xb = (de >> 8) & 31u;
                const uint32_t dist = (de >> 16) + ((uint32_t)bits & ((1u << xb) - 1u));
                bits >>= xb; used += xb;
                if (WRITE) {
                    S.matches[mbase + nm] = make_uint2(obase + o, len | dist << 16);
                    if (dist > obase + o) atomicMax(&S.ctrl[C_ERR], (uint32_t)E_DISTANCE);     // reaches before the first output byte
                }
                ++nm; o += len;
                if (o > (uint32_t)kLaneOutMax) fl = F_EARLY;
            } else { fl = kind == 2u ? F_EOB : F_BAD; break; }
        } while (!fl && used <= 16u && pos + used < end);
        pos += used;
    }
    return LaneResult{ pos, fl, o, nm };
}

__global__ __launch_bounds__(kT) void k_inflate(const InfItem* items, int n_items, uint32_t* out_len, uint32_t* status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Shared& S = *reinterpret_cast<Shared*>(smem);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    struct __attribute__((packed, aligned(1))) AnyVec { u32x4 v; };
    struct __attribute__((packed, aligned(1))) AnyU32 { uint32_t v; };
    const int t = threadIdx.x;
    if ((int)blockIdx.x >= n_items) return;
    const InfItem it = items[blockIdx.x];
    const uint8_t* src = it.src;
    const uint64_t src_bits = (uint64_t)it.src_len * 8u;

    uint64_t pos = 0;                                 // next unread bit of the stream
    uint32_t produced = 0;                            // bytes inflated so far (ring index = produced & kRingMask)
    uint32_t err = 0;
    bool done = false;
    if (t < C_N) S.ctrl[t] = 0;
    __syncthreads();

    // the window: kWinDwords dwords from the dword that holds bit `pos` (relative to the stream start); zeros past the end
    auto load_window = [&](uint64_t at_bit) -> uint64_t {
        const uint64_t base_byte = (at_bit >> 3) & ~(uint64_t)3;
        for (int q = t; q < kWinDwords / 4; q += kT) {
            const uint64_t b = base_byte + (uint64_t)q * 16u;
            u32x4 v = {0, 0, 0, 0};
            if (b + 16u <= it.src_len) v = reinterpret_cast<const AnyVec*>(src + b)->v;
            else if (b < it.src_len) {
                uint32_t w[4] = {0, 0, 0, 0};
                for (uint32_t k = 0; k < 16u && b + k < it.src_len; ++k) w[k >> 2] |= (uint32_t)src[b + k] << (8u * (k & 3u));
                v = u32x4{w[0], w[1], w[2], w[3]};
            }
            *reinterpret_cast<u32x4*>(&S.win[q * 4]) = v;
        }
        __syncthreads();
        return base_byte;
    };
    // ring bytes [from, from + n) -> dst (clamped to the caller's capacity)
    auto flush = [&](uint32_t from, uint32_t n) {
        if (from >= it.dst_cap) return;
        if (n > it.dst_cap - from) n = it.dst_cap - from;
        uint8_t* d = it.dst + from;
        const uint32_t head = (uint32_t)((4u - ((uintptr_t)d & 3u)) & 3u) < n ? (uint32_t)((4u - ((uintptr_t)d & 3u)) & 3u) : n;
        if ((uint32_t)t < head) d[t] = S.ring[(from + t) & kRingMask];
        const uint32_t body = (n - head) >> 2;
        const uint32_t* ring32 = reinterpret_cast<const uint32_t*>(S.ring);
        for (uint32_t k = t; k < body; k += kT) {
            const uint32_t i = from + head + 4u * k, r = i & kRingMask;
            const uint32_t w0 = ring32[r >> 2], w1 = ring32[((r >> 2) + 1u) & (kRing / 4 - 1)];
            reinterpret_cast<uint32_t*>(d + head)[k] = __builtin_amdgcn_alignbit(w1, w0, 8u * (r & 3u));
        }
        const uint32_t tail0 = head + 4u * body;
        if ((uint32_t)t < n - tail0) d[tail0 + t] = S.ring[(from + tail0 + t) & kRingMask];
    };

    while (!done && !err) {
        // ------------------------------------------------------------------ block header
        const uint64_t base_byte = load_window(pos);
        if (t == 0) {
            uint32_t p = (uint32_t)(pos - base_byte * 8u);                       // window-relative bit
            uint64_t bb = peek64(S.win, p);
            const uint32_t bfinal = (uint32_t)bb & 1u, btype = ((uint32_t)bb >> 1) & 3u;
            p += 3;
            S.ctrl[C_FINAL] = bfinal; S.ctrl[C_BTYPE] = btype;
            if (btype == 2u) {
                bb = peek64(S.win, p);
                const uint32_t hlit = ((uint32_t)bb & 31u) + 257u, hdist = ((uint32_t)(bb >> 5) & 31u) + 1u, hclen = ((uint32_t)(bb >> 10) & 15u) + 4u;
                p += 14;
                S.ctrl[C_HLIT] = hlit; S.ctrl[C_HDIST] = hdist;
                uint32_t bad = (hlit > 286u || hdist > 30u) ? 1u : 0u;
                // the code-length code: 19 symbols of at most 7 bits, into a 128-entry table (entry = symbol << 4 | length)
                const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
                uint32_t cl[19]; uint32_t count[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
                for (int i = 0; i < 19; ++i) cl[i] = 0;
                bb = peek64(S.win, p);
                for (uint32_t i = 0; i < hclen; ++i) { const uint32_t v = (uint32_t)(bb >> (3u * i)) & 7u; cl[order[i]] = v; }
                p += 3u * hclen;
                for (int i = 0; i < 19; ++i) ++count[cl[i]];
                count[0] = 0;
                int left = 1;
                for (int l = 1; l <= 7; ++l) { left = left * 2 - (int)count[l]; if (left < 0) bad = 1; }
                if (left > 0) bad = 1;                                               // zlib: an incomplete code-length code is invalid
                for (int e = 0; e < 128; ++e) S.cl_lut[e] = 0;
                if (!bad) {
                    uint32_t next[8]; next[1] = 0;
                    for (int l = 2; l <= 7; ++l) next[l] = (next[l - 1] + count[l - 1]) << 1;
                    for (uint32_t s = 0; s < 19u; ++s) {
                        const uint32_t L = cl[s];
                        if (!L) continue;
                        const uint32_t code = next[L]++, rev = __brev(code) >> (32u - L);
                        for (uint32_t e = rev; e < 128u; e += 1u << L) S.cl_lut[e] = s << 4 | L;
                    }
                    // the hlit + hdist code lengths
                    const uint32_t total = hlit + hdist;
                    uint32_t i = 0, prev = 0, have = 0; uint64_t buf = 0;
                    while (i < total && !bad) {
                        if (have < 16u) { buf = peek64(S.win, p); have = 64; }
                        const uint32_t e = S.cl_lut[(uint32_t)buf & 127u];
                        const uint32_t L = e & 15u, s = e >> 4;
                        if (!L) { bad = 1; break; }
                        buf >>= L; have -= L; p += L;
                        if (s < 16u) { S.lens[i++] = (uint8_t)s; prev = s; continue; }
                        uint32_t rep, val = 0;
                        if (s == 16u) { if (i == 0) { bad = 1; break; } val = prev; rep = 3u + ((uint32_t)buf & 3u); buf >>= 2; have -= 2; p += 2; }
                        else if (s == 17u) { rep = 3u + ((uint32_t)buf & 7u); buf >>= 3; have -= 3; p += 3; prev = 0; }
                        else { rep = 11u + ((uint32_t)buf & 127u); buf >>= 7; have -= 7; p += 7; prev = 0; }
                        if (i + rep > total) { bad = 1; break; }
                        for (uint32_t k = 0; k < rep; ++k) S.lens[i++] = (uint8_t)val;
                    }
                    if (!bad && S.lens[256] == 0) bad = 1;                           // no end-of-block code
                }
                if (bad) atomicMax(&S.ctrl[C_ERR], (uint32_t)E_LENGTHS);
            } else if (btype == 1u) {
                for (int s = 0; s < 288; ++s) S.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
                for (int s = 0; s < 32; ++s) S.lens[288 + s] = 5;
                S.ctrl[C_HLIT] = 288; S.ctrl[C_HDIST] = 32;
            } else if (btype == 3u) atomicMax(&S.ctrl[C_ERR], (uint32_t)E_BLOCK_TYPE);
            const uint64_t end = base_byte * 8u + p;
            S.ctrl[C_HDR_END_LO] = (uint32_t)end; S.ctrl[C_HDR_END_HI] = (uint32_t)(end >> 32);
        }
        __syncthreads();
        const uint32_t btype = S.ctrl[C_BTYPE];
        const bool bfinal = S.ctrl[C_FINAL] != 0;
        pos = (uint64_t)S.ctrl[C_HDR_END_HI] << 32 | S.ctrl[C_HDR_END_LO];
        err = S.ctrl[C_ERR];
        if (!err && pos > src_bits) err = E_INPUT;
        if (err) break;

        if (btype == 0u) {
            // -------------------------------------------------------------- stored block: LEN, ~LEN, bytes
            const uint64_t p = (pos + 7u) >> 3;
            if (p + 4u > it.src_len) { err = E_INPUT; break; }
            const uint32_t len = src[p] | (uint32_t)src[p + 1] << 8, nlen = src[p + 2] | (uint32_t)src[p + 3] << 8;
            if (len != (nlen ^ 0xFFFFu)) { err = E_STORED; break; }
            if (p + 4u + len > it.src_len) { err = E_INPUT; break; }
            for (uint32_t o = 0; o < len; o += (uint32_t)kNewMax) {
                const uint32_t n = len - o < (uint32_t)kNewMax ? len - o : (uint32_t)kNewMax;
                for (uint32_t i = t; i < n; i += kT) S.ring[(produced + i) & kRingMask] = src[p + 4u + o + i];
                __syncthreads();
                flush(produced, n);
                __syncthreads();
                produced += n;
            }
            pos = (p + 4u + len) * 8u;
        } else {
            // -------------------------------------------------------------- Huffman block: tables, then chunks
            const int hlit = (int)S.ctrl[C_HLIT], hdist = (int)S.ctrl[C_HDIST];
            build_table<false>(S, 0, hlit);
            build_table<true>(S, btype == 1u ? 288 : hlit, hdist);
            err = S.ctrl[C_ERR];
            if (err) break;
            bool in_block = true;
            while (in_block && !err) {
                const uint64_t wbase = load_window(pos);
                const uint32_t rel0 = (uint32_t)(pos - wbase * 8u);
                // speculative sweep, then sweeps from the predecessors' exits until the chain is consistent up to its end
                uint32_t my_start = t == 0 ? rel0 : (uint32_t)t * kSubBits;
                const uint32_t my_end = (uint32_t)(t + 1) * kSubBits;
                LaneResult r = lane_decode<false>(S, my_start, my_end, 0, 0);
                S.exit_bit[t] = r.exit; S.flags[t] = r.flags;
                if (t == 0) { S.ctrl[C_FIRST_BAD] = kT; S.ctrl[C_FIRST_STOP] = kT; }
                __syncthreads();
                uint32_t first_stop = kT;
                for (int sweep = 0; sweep <= kT; ++sweep) {
                    const uint32_t prev = t ? S.exit_bit[t - 1] : rel0;
                    const bool moved = t > 0 && prev != my_start;
                    if (moved) atomicMin(&S.ctrl[C_FIRST_BAD], (uint32_t)t);
                    if (r.flags) atomicMin(&S.ctrl[C_FIRST_STOP], (uint32_t)t);
                    __syncthreads();
                    const uint32_t first_bad = S.ctrl[C_FIRST_BAD];
                    first_stop = S.ctrl[C_FIRST_STOP];
                    __syncthreads();
                    if (first_bad == (uint32_t)kT || first_stop < first_bad) break;
                    if (t == 0) { S.ctrl[C_FIRST_BAD] = kT; S.ctrl[C_FIRST_STOP] = kT; }
                    if (moved) { my_start = prev; r = lane_decode<false>(S, my_start, my_end, 0, 0); }
                    __syncthreads();
                    S.exit_bit[t] = r.exit; S.flags[t] = r.flags;
                    __syncthreads();
                }
                __syncthreads();
                if (t == 0) { S.ctrl[C_FIRST_BAD] = kT; S.ctrl[C_FIRST_STOP] = kT; S.ctrl[C_CUT] = kT; }
                uint32_t nvalid = first_stop < (uint32_t)kT ? first_stop + 1u : (uint32_t)kT;      // lanes 0 .. nvalid - 1 form the chain
                // where everything goes; lanes that would overflow the ring / the match list wait for the next chunk
                uint32_t inc_o = (uint32_t)t < nvalid ? r.out : 0u, inc_m = (uint32_t)t < nvalid ? r.nm : 0u;
                block_scan2(S, inc_o, inc_m);
                if ((uint32_t)t < nvalid && (inc_o > (uint32_t)kNewMax || inc_m > (uint32_t)kMatchCap)) atomicMin(&S.ctrl[C_CUT], (uint32_t)t);
                __syncthreads();
                const uint32_t cut = S.ctrl[C_CUT];
                const bool was_cut = cut < nvalid;
                if (was_cut) nvalid = cut;                                            // >= 1: one lane alone always fits
                const uint32_t total = S.scan_a[nvalid - 1], nm_total = S.scan_b[nvalid - 1];
                const uint32_t last_exit = S.exit_bit[nvalid - 1], last_flags = was_cut ? 0u : S.flags[nvalid - 1];
                __syncthreads();
                if ((uint32_t)t < nvalid) lane_decode<true>(S, my_start, my_end, produced + (inc_o - r.out), inc_m - r.nm);
                __syncthreads();
                err = S.ctrl[C_ERR];
                if (!err && (last_flags & F_BAD)) err = E_CODE;
                if (!err && wbase * 8u + last_exit > src_bits) err = E_INPUT;
                if (err) break;
                // matches, 256 at a time: a match copies once its source lies below the first pending destination
                for (uint32_t g = 0; g < nm_total; g += kT) {
                    const bool have = g + t < nm_total;
                    uint32_t d = 0, len = 0, dist = 1, src_end = 0;
                    if (have) {
                        const uint2 m = S.matches[g + t];
                        d = m.x; len = m.y & 0xFFFFu; dist = m.y >> 16;
                        src_end = d - dist + (len < dist ? len : dist);
                    }
                    bool pending = have;
                    for (;;) {
                        if (t == 0) S.ctrl[C_HWM] = 0xFFFFFFFFu;
                        __syncthreads();
                        if (pending) atomicMin(&S.ctrl[C_HWM], d);
                        __syncthreads();
                        const uint32_t hwm = S.ctrl[C_HWM];
                        if (hwm == 0xFFFFFFFFu) break;
                        if (pending && src_end <= hwm) {
                            for (uint32_t i = 0; i < len; ++i) S.ring[(d + i) & kRingMask] = S.ring[(d + i - dist) & kRingMask];
                            pending = false;
                        }
                        __syncthreads();
                    }
                }
                __syncthreads();
                flush(produced, total);
                __syncthreads();
                produced += total;
                pos = wbase * 8u + last_exit;
                if (last_flags & F_EOB) in_block = false;
                if (produced >= it.dst_cap) { in_block = false; done = true; }          // the caller wants no more than this
            }
        }
        if (bfinal) done = true;
    }
    if (t == 0) {
        out_len[blockIdx.x] = produced < it.dst_cap ? produced : it.dst_cap;
        status[blockIdx.x] = err;
    }
}

} // namespace

int inflate_launch(const gamut_hip_inflate_desc* descs, int count, uint32_t* out_len_dev, uint32_t* status_dev, hipStream_t stream)
{
    static_assert(sizeof(InfItem) == sizeof(gamut_hip_inflate_desc), "descriptor layout");
    std::vector<InfItem> items((size_t)count);                 // pageable: the upload below has read it when hipMemcpyAsync returns
    for (int i = 0; i < count; ++i) {
        if (!descs[i].src || (!descs[i].dst && descs[i].dst_cap)) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "inflate: stream %d: null pointer", i);
        items[(size_t)i] = InfItem{ descs[i].src, descs[i].dst, descs[i].src_len, descs[i].dst_cap };
    }
    static const bool attr_set = hipFuncSetAttribute((const void*)k_inflate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Shared)) == hipSuccess;
    if (!attr_set) return set_error(GAMUT_HIP_ERR_HIP, "inflate: %zu bytes of LDS are not available", sizeof(Shared));
    void* d_items = nullptr;                                   // stream-ordered: lives until the kernel has run
    GAMUT_HIP_CHECK(hipMallocAsync(&d_items, items.size() * sizeof(InfItem), stream));
    GAMUT_HIP_CHECK(hipMemcpyAsync(d_items, items.data(), items.size() * sizeof(InfItem), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_inflate, dim3((unsigned)count), dim3(kT), sizeof(Shared), stream, (const InfItem*)d_items, count, out_len_dev, status_dev);
    const int rc = launch_status("inflate");
    GAMUT_HIP_CHECK(hipFreeAsync(d_items, stream));
    return rc;
}

} // namespace gamut

using namespace gamut;

extern "C" int gamut_hip_inflate_batch_device(const gamut_hip_inflate_desc* descs, int count, uint32_t* out_len_dev, uint32_t* status_dev, void* stream)
{
    clear_error();
    if (count < 0 || (count > 0 && (!descs || !out_len_dev || !status_dev))) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "inflate_batch_device: bad arguments");
    if (count == 0) return GAMUT_HIP_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_error(GAMUT_HIP_ERR_NO_DEVICE, "no HIP device available (libgamut_hip has no CPU fallback)");
    try { return inflate_launch(descs, count, out_len_dev, status_dev, pick_stream(stream)); }
    catch (...) { return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "inflate_batch_device: out of host memory"); }
}
