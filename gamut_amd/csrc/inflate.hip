// inflate.hip -- DEFLATE (RFC 1951) on the GPU, for batches of PNG files.
//
// Replaces, for a batch, stbi_zlib_decode_malloc_guesssize_headerflag (stbdec.d:1267-1321 -> the `miniz` inflate): IDAT
// streams resident in HBM are inflated straight into the arena the de-filter kernels read (png.hip), so a file batch no
// longer waits for sixteen host threads running zlib.  One workgroup of kT = 1024 threads per stream.  Huffman decoding does
// not depend on the bytes the matches copy, so a block is taken in two decoupled steps:
//
//   TOKENS, 16 KiB of compressed input per round (kT lanes x 128 bits), whatever they inflate to:
//   * block headers and the code-length alphabet of a dynamic block are read by one thread (a few hundred bits); the two
//     canonical Huffman codes are turned into lookup tables by all threads;
//   * SPECULATIVE decode: lane k starts at bit 128 k of the round as if a token began there and remembers where its tokens
//     began (a 128-bit map).  Lanes then restart from their predecessor's exit -- only until they step on a bit of their
//     own map: from there on the old chain holds (Huffman streams re-synchronise within a few tokens).  Inside a wave the
//     exits travel by DPP, without a barrier; waves exchange their last lane's exit through LDS, a few times per round;
//   * the settled lanes decode once more and EMIT their tokens into a scratch list in HBM (8 bytes: literal pair or match,
//     offset inside the lane's output, lane), slots from a prefix sum of the maps' bit counts;
//   BYTES, in tiles of at most 28 KiB of output (whole lanes; a round of smooth data is several tiles):
//   * a token per thread: literals go into a 64 KiB ring in LDS (the 32 KiB window + the tile), every byte of a match gets
//     the position it copies from (from[], 16 bits per byte; byte i of an overlapping match points at byte i mod distance
//     of the block before it, so runs are one link long; long matches are expanded by whole waves);
//   * pointer doubling on from[] alone: a byte whose source is not a literal or history adopts its source's source -- a chain
//     of n dependent copies is done after log2 n rounds, whatever the match structure; then ONE pass moves the values;
//   * the tile's bytes leave the ring for HBM in dwords.
// No data-dependent branch is taken on a whole stream's behalf by a single lane except the block headers.
#include "common.hpp"

namespace gamut {
namespace {

#ifndef INFLATE_T                 // tuning knobs (tools/variant.sh)
#define INFLATE_T 1024
#endif
#ifndef INFLATE_NEW_MAX
#define INFLATE_NEW_MAX 28672
#endif
constexpr int kT = INFLATE_T;                        // threads per stream
constexpr int kWaves = kT / 64;
constexpr int kSubBits = 256;                        // compressed bits a lane owns per round, eight words of 32 (every pass of a round is a chain of
                                                     // dependent look-ups per lane, not arithmetic: the longer the lanes, the more bytes per pass)
constexpr int kRoundBytes = kT * kSubBits / 8;       // 32768
constexpr int kWinDwords = kRoundBytes / 4;          // (the last lane would run past it: it decodes along, but a round ends with the lane in front of it)
constexpr int kRing = 65536, kRingMask = kRing - 1;
constexpr int kNewMax = INFLATE_NEW_MAX;             // bytes a tile may add to the ring: the 32 KiB history must survive them
// (a lane's tokens begin within its 256 bits and take 2 bits at the least: up to 151 matches of 258 bytes = 39 KB, more than a tile: tiles begin and
// end anywhere, a token that straddles two of them is worked into both)
constexpr int kHist = 32768;                         // DEFLATE's window
constexpr int kTokCap = 32768;                       // tokens a round may emit (its scratch list in HBM); one lane holds at most 151
#ifndef INFLATE_LONG_CAP
#define INFLATE_LONG_CAP 768
#endif
#ifndef INFLATE_LIT_BITS
#define INFLATE_LIT_BITS 11
#endif
#ifndef INFLATE_PAIRS               // two literals behind one look-up (tools/variant.sh A/B)
#define INFLATE_PAIRS 1
#endif
#ifndef INFLATE_LONG_MIN
#define INFLATE_LONG_MIN 32
#endif
constexpr int kLongMin = INFLATE_LONG_MIN, kLongCap = INFLATE_LONG_CAP;        // matches at least this long are expanded by a whole wave, not by the thread that holds the token
constexpr int kLitBits = INFLATE_LIT_BITS, kDistBits = 10;         // primary lookup widths; longer codes take the canonical search

enum : uint32_t { F_EOB = 1, F_BAD = 2 };
// status word per stream (0 = ok)
enum : uint32_t { E_BLOCK_TYPE = 1, E_STORED = 2, E_LENGTHS = 3, E_CODE = 4, E_DISTANCE = 5, E_INPUT = 6 };
enum { C_JOBS0 = 0, C_JOBS1, C_JOBS2, C_STOP0, C_STOP1, C_STOP2, C_CUT, C_LAST, C_OPEN0, C_OPEN1, C_OPEN2, C_NLONG, C_ERR, C_BTYPE, C_FINAL, C_HDR_END_LO, C_HDR_END_HI, C_HLIT, C_HDIST, C_HCLEN, C_SLOT, C_N };

struct InfItem { const uint8_t* src; uint8_t* dst; uint32_t src_len, dst_cap; };
// A stream between two launches of a sliced call (inflate_launch_sliced: the launches follow the upload, slice by slice): where it stands,
// and the code lengths of the Huffman block it stands in (the tables are built again from them; the window comes back from the output)
struct InfState {
    uint64_t pos; uint32_t produced, blocks, err, flags;       // flags: 1 = done, 2 = inside a Huffman block, 4 = a launch has run
    uint32_t btype, bfinal, hlit, hdist;
    uint8_t lens[320];
};
enum : uint32_t { ST_DONE = 1, ST_IN_BLOCK = 2, ST_VALID = 4 };

#ifndef INFLATE_PROFILE           // measurement only (tools/variant.sh inflate:prof:-DINFLATE_PROFILE=1): cycles per phase, summed over streams
#define INFLATE_PROFILE 0
#endif
enum { P_HEADER = 0, P_TABLES, P_WINDOW, P_SWEEP0, P_SWEEPS, P_SCAN, P_WRITE, P_MATCH, P_FLUSH, P_STORED,
       P_H_FIELDS, P_H_CODE, P_H_WALKS, P_H_EMIT, P_T_RANKS, P_T_STARTS, P_T_SORT, P_T_LIT, P_T_DIST, P_T_LONG, P_F_SETUP, P_F_INIT, P_S_DETECT, P_R_EXPAND, P_R_INIT, P_R_ROUNDS, P_S_TURN1,
       P_N_BLOCKS, P_N_CHUNKS, P_N_SWEEPS, P_N_ROUNDS, P_N_MATCHES, P_N };
#if INFLATE_PROFILE
__device__ unsigned long long g_inflate_prof[P_N];
struct Prof {
    unsigned long long t0 = clock64(), acc[P_N] = {};
    __device__ __forceinline__ void mark(int slot) { const unsigned long long now = clock64(); acc[slot] += now - t0; t0 = now; }
    __device__ __forceinline__ void count(int slot, unsigned long long n) { acc[slot] += n; }
    __device__ __forceinline__ void flush() { if (threadIdx.x == 0) for (int k = 0; k < P_N; ++k) atomicAdd(&g_inflate_prof[k], acc[k]); }
};
#else
struct Prof {
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void count(int, unsigned long long) {}
    __device__ __forceinline__ void flush() {}
};
#endif
#define PROF_DECL Prof prof
#define PROF(slot) prof.mark(slot)
#define PROF_COUNT(slot, n) prof.count(slot, n)
#define PROF_FLUSH prof.flush()

struct __attribute__((aligned(16))) Canon { uint32_t first[16], count[16], offs[16]; };

struct Shared {
    __attribute__((aligned(16))) uint8_t ring[kRing];
    union {                                          // the compressed window and the lanes' states serve the token rounds, from[] the tiles behind them
        struct {
            __attribute__((aligned(16))) uint32_t win[kWinDwords];
            uint32_t lane_start[kT];                 // where the lane's chain begins (window-relative bit)
            __attribute__((aligned(16))) uint4 lane_map[kT];      // per 32-bit word of the lane: where its first token begins, how many begin in it (Lane)
            uint16_t jobs[kT];                       // lanes that have to walk again
        };
        __attribute__((aligned(16))) uint16_t from[kNewMax + 8];   // per byte of the tile: where its value comes from, as a position in [tile start - 32768, ...); itself = a literal
    };
    uint32_t lut[(1 << kLitBits) + (1 << kDistBits)];   // primary tables: literal / length codes, then distance codes
    uint32_t lane_exit[kT];                          // exit bit | flags << 24, as last published
    uint32_t cum_out0[kT + 1], cum_tok[kT];          // prefix sums over the lanes of a round: cum_out0[k] = bytes in front of lane k, cum_tok[k] = tokens up to and with lane k
    uint2    longm[kLongCap];                        // long matches of the tile: x = first byte (tile-relative), y = length | distance << 16
    uint32_t wave_sum[kWaves];
    uint32_t wave_count[5][16];                      // build_tables: literal / length symbols of every code length, per wave
    uint32_t cl_code[19];                            // code-length alphabet: bit-reversed code << 4 | length (0: unused symbol)
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

// symbol -> table payload: bits 0-3 code length (0 in a primary table: longer than the table covers), 4-6 kind (0 literal, 1 length,
// 2 end of block, 3 invalid, 4 two literals, 5 distance), 8-11 extra bits, 12-15 the first code's length of a pair, 16-31 base
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
    if (d < 4u) return (1u + d) << 16 | 5u << 4;
    const uint32_t x = (d - 2u) >> 1;
    return (1u + ((2u + (d & 1u)) << x)) << 16 | x << 8 | 5u << 4;
}

// the code lengths [FROM, TO] of a canonical code in registers: a search costs no LDS round trip but the last
template <int FROM, int TO = 15> struct CodeRange {
    uint32_t first[TO - FROM + 1], count[TO - FROM + 1], offs[TO - FROM + 1];
    __device__ __forceinline__ void load(const Canon& c)
    {
        // (every read in flight first; then the values -- the same for every lane -- move to scalar registers)
        uint32_t f[TO - FROM + 1], n[TO - FROM + 1], o[TO - FROM + 1];
        #pragma unroll
        for (int l = FROM; l <= TO; ++l) { f[l - FROM] = c.first[l]; n[l - FROM] = c.count[l]; o[l - FROM] = c.offs[l]; }
        #pragma unroll
        for (int l = FROM; l <= TO; ++l) {
            first[l - FROM] = (uint32_t)__builtin_amdgcn_readfirstlane((int)f[l - FROM]);
            count[l - FROM] = (uint32_t)__builtin_amdgcn_readfirstlane((int)n[l - FROM]);
            offs[l - FROM] = (uint32_t)__builtin_amdgcn_readfirstlane((int)o[l - FROM]);
        }
    }
    // `bits` holds the stream bits LSB first; lengths up to `to` only; 0 = no code of these lengths matches
    __device__ __forceinline__ uint32_t decode(uint32_t bits, const uint32_t* sorted, int to = TO) const
    {
        const uint32_t rev = __brev(bits);
        uint32_t at = 0xFFFFFFFFu, len = 0;
        #pragma unroll
        for (int l = TO; l >= FROM; --l) {                         // (at most one length matches: the code is prefix-free)
            const uint32_t d = (rev >> (32 - l)) - first[l - FROM];
            const bool hit = d < count[l - FROM] && l <= to;
            at = hit ? offs[l - FROM] + d : at; len = hit ? (uint32_t)l : len;
        }
        return len ? sorted[at] | len : 0u;
    }
};
template <int FROM> using LongCodes = CodeRange<FROM, 15>;

// The two canonical codes of a block -- hlit literal / length symbols with lengths S.lens[0 ..), hdist distance symbols with lengths
// S.lens[dist_base ..) -- in one go: Canon, payloads in canonical order, primary tables.  Threads 0-287 hold a literal / length symbol
// each, threads 288-319 a distance symbol: a symbol's rank among the symbols of its length is a ballot inside its wave plus the
// counts of the waves before it; code starts and offsets are prefix sums over the 15 lengths (DPP, lanes 0-15 and 16-31 of wave 0).
// Validity as zlib's inflate_table: over-subscribed sets fail; incomplete ones too, unless no code is longer than one bit.
__device__ __noinline__ void build_tables(Shared& S, int hlit, int dist_base, int hdist, Prof& prof)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool is_lit = t < 288;
    uint32_t mylen = 0, rank = 0;
    if (wave < 5) {
        if (is_lit) { if (t < hlit) mylen = S.lens[t]; }
        else if (t - 288 < hdist) mylen = S.lens[dist_base + t - 288];
        const uint64_t lit_lanes = wave < 4 ? ~0ull : 0xFFFFFFFFull;        // wave 4: symbols 256-287, then the distance symbols
        const uint64_t below = (1ull << lane) - 1ull;
        for (uint32_t l = 1; l <= 15; ++l) {
            const uint64_t m = __ballot(mylen == l);
            if (mylen == l) rank = (uint32_t)__popcll((is_lit ? m & lit_lanes : m & ~lit_lanes) & below);
            if (lane == 0) { S.wave_count[wave][l] = (uint32_t)__popcll(m & lit_lanes); if (wave == 4) S.dist.count[l] = (uint32_t)__popcll(m & ~lit_lanes); }
        }
    }
    __syncthreads();
    PROF(P_T_RANKS);
    if (t < 32) {
        const uint32_t l = (uint32_t)t & 15u;
        Canon& c = t < 16 ? S.lit : S.dist;
        uint32_t cnt = 0;
        if (l) cnt = t < 16 ? S.wave_count[0][l] + S.wave_count[1][l] + S.wave_count[2][l] + S.wave_count[3][l] + S.wave_count[4][l] : S.dist.count[l];
        // first[l] = sum over j < l of count[j] << (l - j): a prefix sum of count[j] << (15 - j), shifted back; offs[l] = codes shorter than l
        uint32_t kraft = cnt << (15u - l), offs = cnt;
#define ROW_SCAN_STEP(CTRL) kraft += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kraft, CTRL, 0xF, 0xF, false); offs += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)offs, CTRL, 0xF, 0xF, false)
        ROW_SCAN_STEP(0x111); ROW_SCAN_STEP(0x112); ROW_SCAN_STEP(0x114); ROW_SCAN_STEP(0x118);      // row_shr:1, 2, 4, 8: inside the row of 16 lanes
#undef ROW_SCAN_STEP
        c.count[l] = cnt;
        c.first[l] = l ? (kraft - (cnt << (15u - l))) >> (15u - l) : 0u;
        c.offs[l] = offs - cnt;
        const uint64_t used = __ballot(cnt != 0);
        const uint32_t row = t < 16 ? (uint32_t)used & 0xFFFFu : (uint32_t)(used >> 16) & 0xFFFFu;
        const uint32_t maxlen = row ? 31u - (uint32_t)__builtin_clz(row) : 0u;
        if (l == 15u && (kraft > 32768u || (kraft < 32768u && maxlen > 1u))) atomicMax(&S.ctrl[C_ERR], (uint32_t)E_LENGTHS);
    }
    __syncthreads();
    PROF(P_T_STARTS);
    if (mylen) {
        const Canon& c = is_lit ? S.lit : S.dist;
        uint32_t at = c.offs[mylen] + rank;
        if (is_lit) for (int w = 0; w < wave; ++w) at += S.wave_count[w][mylen];
        if (is_lit) S.lit_sorted[at] = lit_payload((uint32_t)t); else S.dist_sorted[at] = dist_payload((uint32_t)t - 288u);
    }
    __syncthreads();
    PROF(P_T_SORT);
    {
        CodeRange<1, kLitBits> codes; codes.load(S.lit);
        for (int e = t; e < (1 << kLitBits); e += kT) {
            uint32_t r = codes.decode((uint32_t)e, S.lit_sorted);
            // two literals behind one look-up when both codes fit the index (kind 4: base = first | second << 8): residual data
            // of photographs is mostly literals of 3-6 bits
            const uint32_t l1 = r & 15u;
            if (INFLATE_PAIRS && r && ((r >> 4) & 7u) == 0u && l1 < (uint32_t)kLitBits) {
                const uint32_t r2 = codes.decode((uint32_t)e >> l1, S.lit_sorted, kLitBits - (int)l1);
                if (r2 && ((r2 >> 4) & 7u) == 0u) r = ((r >> 16) | (r2 >> 16) << 8) << 16 | l1 << 12 | 4u << 4 | (l1 + (r2 & 15u));
            }
            S.lut[e] = r;
        }
    }
    PROF(P_T_LIT);
    {
        CodeRange<1, kDistBits> codes; codes.load(S.dist);
        for (int e = t; e < (1 << kDistBits); e += kT) S.lut[(1 << kLitBits) + e] = codes.decode((uint32_t)e, S.dist_sorted);
    }
    __syncthreads();
    PROF(P_T_DIST);
}

// What a lane knows about its sub-sequence: where it starts, where its chain of tokens leaves it (the first token boundary at or
// beyond its end), whether the chain met the end-of-block code or an invalid code, and per 32-bit word of the lane (6 bits each)
// `first`: 32 | the bit its first token begins at (0: none begins there), `count`: how many tokens begin in it.  Two chains that begin
// a word's first token on the same bit are one chain from there on.
struct Lane { uint32_t start, exit, flags; uint64_t first, count; };

// One code: literal / length / end of block from the first table, or, after a length, the distance from the second -- every
// trip of the decode loops runs the same instructions whatever it finds (a wave whose lanes meet literals, pairs and matches
// side by side does not walk three paths): entry -> code bits, extra bits, base + extra, then selects.
struct Code { uint32_t kind, val, bits; };
template <class LL, class LD>
__device__ __forceinline__ Code read_code(const Shared& S, const LL& long_lit, const LD& long_dist, uint32_t b, uint32_t state, uint32_t at, uint32_t end)
{
    uint32_t e = S.lut[state ? (1u << kLitBits) + (b & ((1u << kDistBits) - 1u)) : (b & ((1u << kLitBits) - 1u))];
    if ((e & 15u) == 0u) { e = state ? long_dist.decode(b, S.dist_sorted) : long_lit.decode(b, S.lit_sorted); if (!e) e = 3u << 4; }
    uint32_t nb = e & 15u, kind = (e >> 4) & 7u;
    if (INFLATE_PAIRS && kind == 4u) {
        // a pair whose second literal would begin at or beyond `end` is taken as its first literal alone: a lane must
        // leave at the FIRST token boundary past its end whatever way it came in, or the lanes never fall into step
        const uint32_t l1 = (e >> 12) & 15u;
        if (at + l1 >= end) { kind = 0u; nb = l1; e &= 0x00FF0FFFu; }
    }
    const uint32_t xb = (e >> 8) & 15u;
    return Code{ kind, (e >> 16) + __builtin_amdgcn_ubfe(b, nb, xb), nb + xb };
}

typedef LongCodes<kLitBits + 1> LongLit;
typedef LongCodes<kDistBits + 1> LongDist;

// The tokens of a lane from bit `from` on (window-relative), until one begins at or beyond its end.  RESYNC: the walk stops as soon as
// it begins a word's first token where the lane's known chain does -- from there on the chain is the one already known (at most a
// word later than the two met); otherwise the lane's exit, flags and words are replaced.  64 peeked bits serve trips of at most 28.
template <bool RESYNC>
__device__ __forceinline__ void lane_trace(const Shared& S, const LongLit& long_lit, const LongDist& long_dist, Lane& L, uint32_t from, uint32_t base)
{
    const uint32_t end = base + kSubBits;
    uint32_t pos = from, fl = 0, state = 0, last_word = 8;
    uint64_t first = 0, count = 0;
    bool joined = false;
    while (!fl && !joined && (state || pos < end)) {
        const uint64_t bits = peek64(S.win, pos);
        uint32_t used = 0;
        do {
            if (!state) {                                         // a token begins here
                const uint32_t rel = pos + used - base, w = rel >> 5;      // w < 8 on every chain that matters (lanes behind a stop may come in below their base)
                if (w < 8u) {
                    if (w != last_word) {
                        const uint32_t mark = 32u | (rel & 31u);
                        if (RESYNC && ((uint32_t)(L.first >> (6u * w)) & 63u) == mark) {
                            const uint64_t keep = ~0ull << (6u * w);
                            L.first = first | (L.first & keep); L.count = count | (L.count & keep);
                            joined = true; break;
                        }
                        first |= (uint64_t)mark << (6u * w);
                        last_word = w;
                    }
                    count += 1ull << (6u * w);
                }
            }
            const Code c = read_code(S, long_lit, long_dist, (uint32_t)(bits >> used), state, pos + used, end);
            used += c.bits;
            state = c.kind == 1u ? 1u : 0u;
            fl = c.kind == 2u ? (uint32_t)F_EOB : c.kind == 3u ? (uint32_t)F_BAD : 0u;
        } while (!fl && used <= 36u && (state || pos + used < end));
        pos += used;
    }
    L.start = from;
    if (!joined) { L.exit = pos; L.flags = fl; L.first = first; L.count = count; }
}

// The settled lane once more: its tokens into the round's list (8 bytes each: x = length | distance << 16 for a match, count (0-2) |
// first << 16 | second << 24 for literals; y = offset inside the lane's output | lane << 16 | match << 31) -> bytes the lane inflates to.
// Every token counted in the lane's words gets a slot: the end-of-block (or invalid) code leaves an empty literal token.
__device__ __forceinline__ uint32_t lane_emit(const Shared& S, const LongLit& long_lit, const LongDist& long_dist, const Lane& L, uint32_t base, uint2* tok, uint32_t lane)
{
    const uint32_t end = base + kSubBits;
    uint32_t pos = L.start, fl = 0, state = 0, o = 0, len = 0;
    while (!fl && (state || pos < end)) {
        const uint64_t bits = peek64(S.win, pos);
        uint32_t used = 0;
        do {
            const Code c = read_code(S, long_lit, long_dist, (uint32_t)(bits >> used), state, pos + used, end);
            used += c.bits;
            if (c.kind != 1u) {                                   // a token is complete
                const bool match = c.kind == 5u, lits = c.kind == 0u || c.kind == 4u;
                const uint32_t n = match ? len : lits ? 1u + (c.kind >> 2) : 0u;
                *tok++ = make_uint2(match ? len | c.val << 16 : n | c.val << 16, o | lane << 16 | (match ? 1u << 31 : 0u));
                o += n;
            }
            len = c.val;
            state = c.kind == 1u ? 1u : 0u;
            fl = c.kind == 2u ? (uint32_t)F_EOB : c.kind == 3u ? (uint32_t)F_BAD : 0u;
        } while (!fl && used <= 36u && (state || pos + used < end));
        pos += used;
    }
    return o;
}

__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);      // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);      // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);      // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);      // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);      // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);      // row_bcast:31 into rows 2 and 3
    return v;
}
// inclusive prefix sum over the workgroup, left in `out[]` too (two barriers)
__device__ __forceinline__ uint32_t block_inclusive_sum(Shared& S, uint32_t v, uint32_t* out)
{
    const int t = threadIdx.x, w = t >> 6;
    uint32_t incl = wave_inclusive_sum(v);
    if ((t & 63) == 63) S.wave_sum[w] = incl;
    __syncthreads();
    uint32_t before = 0;
    #pragma unroll
    for (int k = 0; k < kWaves; ++k) before += k < w ? S.wave_sum[k] : 0u;
    incl += before;
    out[t] = incl;
    __syncthreads();
    return incl;
}

// S.from[] for the matches the token pass left in S.longm (x = first byte to write | bytes to write << 15, y = bytes between the point
// the run is copied from and that first byte | distance << 16): one match per wave and pass, 64 bytes per step
__device__ __forceinline__ void expand_long_matches(Shared& S)
{
    const uint32_t n = S.ctrl[C_NLONG] < (uint32_t)kLongCap ? S.ctrl[C_NLONG] : (uint32_t)kLongCap;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t m = wave; m < n; m += kWaves) {
        const uint2 e = S.longm[m];
        const uint32_t at = e.x & 0x7FFFu, len = e.x >> 15, since = e.y & 0xFFFFu, dist = e.y >> 16, q = at - since + kHist - dist;
        if (since + len <= dist) { for (uint32_t i = lane; i < len; i += 64u) S.from[at + i] = (uint16_t)(q + since + i); }
        else {
            const uint32_t step = 64u % dist;
            uint32_t r = (since + lane) % dist;                    // (since + i) mod dist, kept up by additions
            for (uint32_t i = lane; i < len; i += 64u) { S.from[at + i] = (uint16_t)(q + r); r += step; if (r >= dist) r -= dist; }
        }
    }
}

// Every byte of the tile [cs, cs + total) takes its value: S.from[j] names the position byte j copies from (a position p
// counts from cs - 32768: p < 32768 is window history; p == j + 32768 is the byte itself = a literal).  Rounds of pointer
// doubling first: a byte whose source is neither adopts its source's source.  A thread looks at 8 neighbouring bytes (one
// 16-byte read of their entries) and remembers which of them are settled.  Then one pass moves the values.
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int resolve_copies(Shared& S, uint32_t cs, uint32_t total, Prof& prof)      // -> rounds taken
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int kUnits = (kNewMax / 8 + kT - 1) / kT;           // 8-byte units per thread
    const int t = threadIdx.x;
    const uint32_t units = (total + 7u) >> 3;
    uint32_t open[kUnits];                                          // per unit: one bit per byte that still points at a copied byte
    #pragma unroll
    for (int k = 0; k < kUnits; ++k) {
        const uint32_t u = (uint32_t)t + (uint32_t)k * kT;
        open[k] = 0;
        if (u >= units) continue;
        const uint32_t j0 = u * 8u, self0 = j0 + kHist;
        const u32x4 f = *reinterpret_cast<const u32x4*>(&S.from[j0]);
        const uint32_t p[8] = { f.x & 0xFFFFu, f.x >> 16, f.y & 0xFFFFu, f.y >> 16, f.z & 0xFFFFu, f.z >> 16, f.w & 0xFFFFu, f.w >> 16 };
        #pragma unroll
        for (int i = 0; i < 8; ++i) if (j0 + i < total && p[i] >= (uint32_t)kHist && p[i] != self0 + i) open[k] |= 1u << i;
    }
    PROF(P_R_INIT);
    // Three flags in turn say "somebody still has work": round r raises flag r % 3 and thread 0 clears the next one, which was last
    // read behind the barrier of round r - 2 (every thread has passed the barrier of round r - 1 since).
    int round = 0;
    for (; round < 20; ++round) {
        bool any = false;
        if (t == 0) S.ctrl[C_OPEN0 + (round + 1) % 3] = 0;
        #pragma unroll
        for (int k = 0; k < kUnits; ++k) {
            if (!open[k]) continue;
            const uint32_t j0 = ((uint32_t)t + (uint32_t)k * kT) * 8u;
            const u32x4 f = *reinterpret_cast<const u32x4*>(&S.from[j0]);
            // Two entries per register all the way (no branch per byte: the step is arithmetic-bound otherwise).  Every byte reads the entry
            // of position p & 0x7FFF: its source's for an open byte (p >= kHist = 0x8000), its own for a literal, anything for a window byte.
            const uint32_t P[4] = { f.x, f.y, f.z, f.w };
            uint32_t pp[8];
            #pragma unroll
            for (int d = 0; d < 4; ++d) { pp[2 * d] = S.from[P[d] & 0x7FFFu]; pp[2 * d + 1] = S.from[(P[d] >> 16) & 0x7FFFu]; }
            uint32_t N[4], still = 0;
            const uint32_t bits = open[k];
            #pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t PP = pp[2 * d] | pp[2 * d + 1] << 16;
                const uint32_t M = ((bits >> (2 * d)) & 1u ? 0xFFFFu : 0u) | ((bits >> (2 * d + 1)) & 1u ? 0xFFFF0000u : 0u);      // the open ones
                N[d] = (PP & M) | (P[d] & ~M);                      // an open byte adopts its source's entry (its own again if the source is a literal)
                const uint32_t X = (PP ^ P[d]) & M;                 // moved on -- and still open if it did not land in the window
                still |= ((X & 0xFFFFu) && (PP & 0x8000u) ? 1u : 0u) << (2 * d) | ((X >> 16) && (PP & 0x80000000u) ? 1u : 0u) << (2 * d + 1);
            }
            // (an entry another thread reads while its owner replaces it holds the old or the new pointer: both are sources of the byte)
            *reinterpret_cast<u32x4*>(&S.from[j0]) = u32x4{ N[0], N[1], N[2], N[3] };
            open[k] = still;
            any |= open[k] != 0;
        }
        if (any) S.ctrl[C_OPEN0 + round % 3] = 1u;
        __syncthreads();
        if (!S.ctrl[C_OPEN0 + round % 3]) break;
    }
    __syncthreads();
    PROF(P_R_ROUNDS);
    // values: every source is a literal of the tile or a byte of the window.  Positions are 16-bit and the ring is 65 536 bytes: ring
    // addresses are packed 16-bit sums.  A literal "copies" itself (no test per byte; bytes of the last unit beyond the tile too: they
    // are ring bytes older than the window).
    const uint32_t src_base = ((cs - (uint32_t)kHist) & 0xFFFFu) * 0x00010001u;
    for (uint32_t u = (uint32_t)t; u < units; u += kT) {
        const uint32_t j0 = u * 8u, self0 = j0 + kHist;
        const u32x4 f = *reinterpret_cast<const u32x4*>(&S.from[j0]);
        const uint32_t s0 = self0 | (self0 + 1u) << 16;
        if (f.x == s0 && f.y == s0 + 0x00020002u && f.z == s0 + 0x00040004u && f.w == s0 + 0x00060006u) continue;      // eight literals
        const uint32_t P[4] = { f.x, f.y, f.z, f.w };
        uint8_t v[8];
        #pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t A = pk_add_u16(P[d], src_base);         // two ring addresses
            v[2 * d] = S.ring[A & 0xFFFFu]; v[2 * d + 1] = S.ring[A >> 16];
        }
        const uint32_t dst0 = (cs + j0) & 0xFFFFu;
        if (dst0 <= 0xFFF8u) {                                      // (all but the unit that straddles the ring's end)
            uint8_t* d8 = &S.ring[dst0];
            #pragma unroll
            for (int i = 0; i < 8; ++i) d8[i] = v[i];
        } else {
            #pragma unroll
            for (int i = 0; i < 8; ++i) S.ring[(dst0 + (uint32_t)i) & kRingMask] = v[i];
        }
    }
    return round + 1;
}

// The code lengths of a dynamic block are themselves a Huffman stream (symbols 0-15 a length, 16 / 17 / 18 repeats with 2 / 3 / 7
// extra bits): wave 0 reads it the way the workgroup reads the block -- lane k walks 32 bits from bit 32 k as if a symbol began
// there, lanes restart from their predecessor's exit (DPP) until nobody moves, a prefix sum of the lengths each lane yields
// places them, one more walk writes S.lens[].
struct ClWalk { uint32_t exit, count, last; };       // last: 0x100 | length, once a symbol other than 16 has set the running length
template <bool EMIT>
__device__ __forceinline__ ClWalk cl_walk(Shared& S, uint32_t from, uint32_t end, uint32_t i0, uint32_t prev, uint32_t total, uint32_t& bad, uint32_t& end_pos)
{
    // (a lane is 32 bits and a symbol 14 at the most: 64 bits read once serve the whole walk)
    const uint64_t bits = peek64(S.win, from);
    uint32_t pos = from, n = 0, last = 0;
    while (pos < end) {
        const uint32_t b = (uint32_t)(bits >> (pos - from));
        const uint32_t e = S.cl_lut[b & 127u];
        const uint32_t L = e & 15u, sym = e >> 4;                   // (the code is complete: every pattern is a symbol)
        const uint32_t xb = sym < 16u ? 0u : sym == 16u ? 2u : sym == 17u ? 3u : 7u;
        const uint32_t x = __builtin_amdgcn_ubfe(b, L, xb);
        const uint32_t rep = sym < 16u ? 1u : sym == 18u ? 11u + x : 3u + x;
        const uint32_t val = sym < 16u ? sym : sym == 16u ? prev : 0u;
        if (EMIT) {
            const uint32_t i = i0 + n;
            if (i < total) {                                      // (what follows the last length is the block's data)
                if ((sym == 16u && i == 0u) || i + rep > total) bad = 1u;
                else {
                    for (uint32_t k = 0; k < rep; ++k) S.lens[i + k] = (uint8_t)val;
                    if (i + rep == total) end_pos = pos + L + xb;
                }
            }
        }
        if (sym != 16u) { prev = val; last = 0x100u | val; }
        n += rep; pos += L + xb;
        if (!L) break;                                            // (never: the caller has checked the code)
    }
    return ClWalk{ pos, n, last };
}

// -> window-relative bit behind the last code length, 0 = the lengths are invalid (an over-subscribed or incomplete code-length code
// -- zlib accepts neither --, a repeat with nothing to repeat, or lengths past the end).  `h`: the bit the hclen 3-bit lengths begin at.
__device__ __noinline__ uint32_t read_code_lengths(Shared& S, uint32_t h, uint32_t hclen, uint32_t total, Prof& prof)
{
    const uint32_t lane = threadIdx.x & 63u;
    // the code-length code: lane s < 19 owns symbol s; its 3-bit length stands at the place the permutation gives it
    constexpr uint64_t kPlaceLo = 3ull | 17ull << 5 | 15ull << 10 | 13ull << 15 | 11ull << 20 | 9ull << 25 | 7ull << 30 | 5ull << 35 | 4ull << 40 | 6ull << 45 | 8ull << 50 | 10ull << 55;   // symbols 0-11
    constexpr uint64_t kPlaceHi = 12ull | 14ull << 5 | 16ull << 10 | 18ull << 15 | 0ull << 20 | 1ull << 25 | 2ull << 30;                                                        // symbols 12-18
    const uint32_t place = lane < 12u ? (uint32_t)(kPlaceLo >> (5u * lane)) & 31u : (uint32_t)(kPlaceHi >> (5u * (lane < 19u ? lane - 12u : 0u))) & 31u;
    const uint64_t bb = peek64(S.win, h);                          // 19 x 3 bits at the most
    const uint32_t len = lane < 19u && place < hclen ? (uint32_t)(bb >> (3u * place)) & 7u : 0u;
    uint32_t next = 0, code = 0, kraft = 0;
    #pragma unroll
    for (uint32_t l = 1; l <= 7; ++l) {
        const uint64_t m = __ballot(len == l);
        if (len == l) code = next + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const uint32_t n = (uint32_t)__popcll(m);
        kraft += n << (7u - l);
        next = (next + n) << 1;
    }
    if (kraft != 128u) return 0u;
    h += 3u * hclen;
    if (lane < 19u) S.cl_code[lane] = len ? (__brev(code) >> (32u - len)) << 4 | len : 0u;
    for (uint32_t e = lane; e < 128u; e += 64u) {
        uint32_t r = 0;
        for (uint32_t sym = 0; sym < 19u; ++sym) {
            const uint32_t c = S.cl_code[sym], L = c & 15u;
            if (L && (e & ((1u << L) - 1u)) == c >> 4) r = sym << 4 | L;
        }
        S.cl_lut[e] = r;
    }
    PROF(P_H_CODE);
    uint32_t done = 0, carry = 0, bad = 0, end_pos = 0;
    for (int round = 0; round < 6 && done < total && !bad; ++round) {
        const uint32_t base = h + 32u * lane, end = base + 32u;
        uint32_t start = base;
        ClWalk r = cl_walk<false>(S, start, end, 0, 0, 0, bad, end_pos);
        for (int turn = 0; turn < 64; ++turn) {
            const uint32_t prev_exit = (uint32_t)__builtin_amdgcn_update_dpp((int)h, (int)r.exit, 0x138, 0xF, 0xF, false);      // wave_shr:1
            const uint64_t enough = __ballot(done + wave_inclusive_sum(r.count) >= total);
            const uint32_t need = enough ? (uint32_t)__builtin_ctzll(enough) : 64u;            // lanes behind it read the block's data
            const bool moved = lane > 0u && prev_exit != start && lane <= need;
            if (!__ballot(moved)) break;
            if (moved) { start = prev_exit; r = cl_walk<false>(S, start, end, 0, 0, 0, bad, end_pos); }
        }
        PROF(P_H_WALKS);
        const uint32_t incl = wave_inclusive_sum(r.count);
        // the running length a lane begins with: the last one set in front of it
        uint32_t set = r.last;
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)set, d); if (lane >= (uint32_t)d && !(set & 0x100u)) set = o; }
        uint32_t before = (uint32_t)__shfl_up((int)set, 1);
        if (lane == 0u || !(before & 0x100u)) before = carry;
        if (done + incl - r.count < total) cl_walk<true>(S, start, end, done + incl - r.count, before & 0xFFu, total, bad, end_pos);
        bad = __ballot(bad != 0) ? 1u : 0u;
        const uint64_t ended = __ballot(end_pos != 0u);
        if (ended) end_pos = (uint32_t)__builtin_amdgcn_readlane((int)end_pos, __builtin_ctzll(ended));
        done += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        h = (uint32_t)__builtin_amdgcn_readlane((int)r.exit, 63);
        const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)set, 63);
        if (all & 0x100u) carry = all;
        PROF(P_H_EMIT);
    }
    return bad || !end_pos ? 0u : end_pos;
}

// states / avail: null for a call of its own; else the stream resumes from states[i], reads no byte at or beyond avail[i] (what has been
// uploaded so far) and stops in front of a round (or block header) it cannot finish within them, leaving its state for the next launch
// tok_scratch holds n_slots token lists, one per RESIDENT workgroup (157 KB of LDS: one workgroup per compute unit), not one per stream:
// a workgroup claims a free list when it starts (slot_busy, a flag per list; there are at least as many lists as workgroups can be
// resident, so the search ends) and gives it back when it ends.
__global__ __launch_bounds__(kT) void k_inflate(const InfItem* items, int n_items, uint2* tok_scratch, uint32_t* slot_busy, uint32_t n_slots,
                                                uint32_t* out_len, uint32_t* status, InfState* states, const uint32_t* avail)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Shared& S = *reinterpret_cast<Shared*>(smem);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    struct __attribute__((packed, aligned(1))) AnyVec { u32x4 v; };
    const int t = threadIdx.x, lane_in_wave = t & 63, wave = t >> 6;
    if ((int)blockIdx.x >= n_items) return;
    const InfItem it = items[blockIdx.x];
    const uint8_t* src = it.src;
    const uint64_t src_bits = (uint64_t)it.src_len * 8u;

    uint64_t pos = 0;                                 // next unread bit of the stream
    uint32_t produced = 0;                            // bytes inflated so far (ring index = produced & kRingMask)
    uint32_t err = 0;
    bool done = false;
    if (t < C_N) S.ctrl[t] = 0;
    if (t == 0) S.cum_out0[0] = 0;
    __syncthreads();
    if (t == 0) {                                     // claim a token list
        uint32_t s = blockIdx.x % n_slots;
        while (atomicCAS(&slot_busy[s], 0u, 1u) != 0u) { s = s + 1 == n_slots ? 0 : s + 1; __builtin_amdgcn_s_sleep(2); }
        S.ctrl[C_SLOT] = s;
    }
    __syncthreads();
    const uint32_t my_slot = S.ctrl[C_SLOT];
    uint2* const toks = tok_scratch + (size_t)my_slot * kTokCap;
    // a sliced call: the bytes that are there, and where the stream stands
    const uint32_t have = avail ? (avail[blockIdx.x] < it.src_len ? avail[blockIdx.x] : it.src_len) : it.src_len;
    const bool all_there = have >= it.src_len;
    auto enough = [&](uint64_t at_bit) -> bool { return all_there || (at_bit >> 3) + kRoundBytes + 64u <= have; };      // a whole window from there
    InfState* const state = states ? states + blockIdx.x : nullptr;
    uint32_t blocks = 0;
    bool resume_block = false, suspended = false, suspended_in_block = false;
    if (state && (state->flags & ST_VALID)) {
        pos = state->pos; produced = state->produced; blocks = state->blocks; err = state->err; done = (state->flags & ST_DONE) != 0;
        resume_block = (state->flags & ST_IN_BLOCK) != 0 && !err && !done;
        if (resume_block) {
            for (int k = t; k < 320; k += kT) S.lens[k] = state->lens[k];
            if (t == 0) { S.ctrl[C_BTYPE] = state->btype; S.ctrl[C_FINAL] = state->bfinal; S.ctrl[C_HLIT] = state->hlit; S.ctrl[C_HDIST] = state->hdist; }
        }
        if (!err && !done && produced && produced < it.dst_cap) {                   // the window: the last 32 KiB of the output so far
            const uint32_t n = produced < (uint32_t)kHist ? produced : (uint32_t)kHist;
            for (uint32_t i = t; i < n; i += kT) S.ring[(produced - n + i) & kRingMask] = it.dst[produced - n + i];
        }
        __syncthreads();
    }
    PROF_DECL;

    // the window: kWinDwords dwords from the dword that holds bit `pos` (relative to the stream start); zeros past the end.
    // A round asks for its successor's window as soon as it knows where it ends (prefetch): the 16 bytes per thread wait in
    // registers while the round's tokens and tiles are worked off.
    auto window_piece = [&](uint64_t base_byte, int q) -> u32x4 {
        const uint64_t b = base_byte + (uint64_t)q * 16u;
        u32x4 v = {0, 0, 0, 0};
        if (b + 16u <= it.src_len) v = reinterpret_cast<const AnyVec*>(src + b)->v;
        else if (b < it.src_len) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (uint32_t k = 0; k < 16u && b + k < it.src_len; ++k) w[k >> 2] |= (uint32_t)src[b + k] << (8u * (k & 3u));
            v = u32x4{w[0], w[1], w[2], w[3]};
        }
        return v;
    };
    static_assert(kWinDwords / 4 <= 2 * kT, "two window pieces per thread at most");
    uint64_t pf_base = ~(uint64_t)0, win_base = ~(uint64_t)0;          // what the prefetch registers / the LDS window hold
    u32x4 pf0 = {0, 0, 0, 0}, pf1 = {0, 0, 0, 0};
    auto prefetch = [&](uint64_t at_bit) {
        pf_base = (at_bit >> 3) & ~(uint64_t)3;
        pf0 = window_piece(pf_base, t);
        if (t + kT < kWinDwords / 4) pf1 = window_piece(pf_base, t + kT);
    };
    auto load_window = [&](uint64_t at_bit) -> uint64_t {
        const uint64_t base_byte = (at_bit >> 3) & ~(uint64_t)3;
        if (base_byte != pf_base) prefetch(at_bit);
        *reinterpret_cast<u32x4*>(&S.win[t * 4]) = pf0;
        if (t + kT < kWinDwords / 4) *reinterpret_cast<u32x4*>(&S.win[(t + kT) * 4]) = pf1;
        pf_base = ~(uint64_t)0;
        win_base = base_byte;
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

    // A stream is untrusted input, and a block costs ~40 000 cycles whatever it holds: an "image" made of empty blocks (10 bits
    // each) would keep a workgroup busy for minutes.  No encoder emits more than a block per scanline or per few KiB; streams
    // with more than one block per 8 compressed bytes (beyond the first 4096) are turned away as corrupt.  The tables of the
    // fixed code are built once per stream.
    const uint32_t block_budget = it.src_len / 8u + 4096u;
    bool fixed_tables = false;                        // the LDS tables hold the fixed code
    uint64_t last_block_bits = 0;                     // the length of the Huffman block before this one (0: none yet)
    while (!done && !err) {
      const uint64_t pos_block = pos;
      if (!resume_block) {
        if (!enough(pos)) { suspended = true; break; }
        if (++blocks > block_budget) { err = E_INPUT; break; }
        // ------------------------------------------------------------------ block header
        PROF(P_FLUSH);
        const uint64_t base_byte = load_window(pos);
        PROF(P_WINDOW); PROF_COUNT(P_N_BLOCKS, 1);
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
                S.ctrl[C_HCLEN] = hclen;
                if (hlit > 286u || hdist > 30u) atomicMax(&S.ctrl[C_ERR], (uint32_t)E_LENGTHS);
            } else if (btype == 1u) {
                if (!fixed_tables) {
                    for (int s = 0; s < 288; ++s) S.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
                    for (int s = 0; s < 32; ++s) S.lens[288 + s] = 5;
                }
                S.ctrl[C_HLIT] = 288; S.ctrl[C_HDIST] = 32;
            } else if (btype == 3u) atomicMax(&S.ctrl[C_ERR], (uint32_t)E_BLOCK_TYPE);
            const uint64_t end = base_byte * 8u + p;
            S.ctrl[C_HDR_END_LO] = (uint32_t)end; S.ctrl[C_HDR_END_HI] = (uint32_t)(end >> 32);
        }
        __syncthreads();
        PROF(P_H_FIELDS);
        if (S.ctrl[C_BTYPE] == 2u && !S.ctrl[C_ERR]) {                                       // the hlit + hdist code lengths
            if (wave == 0) {
                const uint32_t h = S.ctrl[C_HDR_END_LO] - (uint32_t)(base_byte * 8u);      // (window-relative: the header began inside the first dword)
                const uint32_t end_rel = read_code_lengths(S, h, S.ctrl[C_HCLEN], S.ctrl[C_HLIT] + S.ctrl[C_HDIST], prof);
                if (t == 0) {
                    if (!end_rel || S.lens[256] == 0) atomicMax(&S.ctrl[C_ERR], (uint32_t)E_LENGTHS);      // (no end-of-block code)
                    const uint64_t end = base_byte * 8u + end_rel;
                    S.ctrl[C_HDR_END_LO] = (uint32_t)end; S.ctrl[C_HDR_END_HI] = (uint32_t)(end >> 32);
                }
            }
            __syncthreads();
        }
        pos = (uint64_t)S.ctrl[C_HDR_END_HI] << 32 | S.ctrl[C_HDR_END_LO];
        err = S.ctrl[C_ERR];
        if (!err && pos > src_bits) err = E_INPUT;
        PROF(P_HEADER);
        if (err) break;
      }
        resume_block = false;
        const uint32_t btype = S.ctrl[C_BTYPE];
        const bool bfinal = S.ctrl[C_FINAL] != 0;

        if (btype == 0u) {
            // -------------------------------------------------------------- stored block: LEN, ~LEN, bytes
            const uint64_t p = (pos + 7u) >> 3;
            if (!all_there && p + 4u + 65535u > have) { pos = pos_block; --blocks; suspended = true; break; }      // (the header is read again next time)
            if (p + 4u > it.src_len) { err = E_INPUT; break; }
            const uint32_t len = src[p] | (uint32_t)src[p + 1] << 8, nlen = src[p + 2] | (uint32_t)src[p + 3] << 8;
            if (len != (nlen ^ 0xFFFFu)) { err = E_STORED; break; }
            if (p + 4u + len > it.src_len) { err = E_INPUT; break; }
            for (uint32_t o = 0; o < len; o += (uint32_t)kNewMax) {
                const uint32_t n = len - o < (uint32_t)kNewMax ? len - o : (uint32_t)kNewMax;
                if (produced < it.dst_cap || o + (uint32_t)kHist >= len) {           // (beyond the caller's capacity only the window matters)
                    for (uint32_t i = t; i < n; i += kT) S.ring[(produced + i) & kRingMask] = src[p + 4u + o + i];
                    __syncthreads();
                    flush(produced, n);
                    __syncthreads();
                }
                produced = produced + n < produced ? 0xFFFFFFFFu : produced + n;
            }
            pos = (p + 4u + len) * 8u;
            PROF(P_STORED);
        } else {
            // -------------------------------------------------------------- Huffman block: tables, then rounds
            const int hlit = (int)S.ctrl[C_HLIT], hdist = (int)S.ctrl[C_HDIST];
            if (!(btype == 1u && fixed_tables)) {
                build_tables(S, hlit, btype == 1u ? 288 : hlit, hdist, prof);
            }
            fixed_tables = btype == 1u;
            err = S.ctrl[C_ERR];
            if (err) break;
            LongLit long_lit; long_lit.load(S.lit);                                   // the code lengths behind the primary tables
            LongDist long_dist; long_dist.load(S.dist);
            PROF(P_T_LONG);
            bool in_block = true;
            const uint64_t block_begin = pos;
            while (in_block && !err) {
                if (!enough(pos)) { suspended = true; suspended_in_block = true; break; }
                // (the window the header was read from serves the block's first round: the lanes in front of `pos` are passed over like
                // lanes behind a long token)
                const bool reuse = win_base != ~(uint64_t)0 && pos >= win_base * 8u && pos - win_base * 8u < 16u * kSubBits;
                const uint64_t wbase = reuse ? win_base : load_window(pos);
                win_base = ~(uint64_t)0;                                              // (the tiles' from[] takes the window's place)
                PROF(P_WINDOW); PROF_COUNT(P_N_CHUNKS, 1);
                const uint32_t rel0 = (uint32_t)(pos - wbase * 8u);
                const uint32_t my_base = (uint32_t)t * kSubBits;
                // ---- tokens: speculative decode, then lanes fall into step with their predecessors
                // Lanes behind the block's end decode its successor's bits with this block's tables: waves of them cost the others their issue
                // slots (the passes of a round are arithmetic-bound with sixteen waves at work).  An encoder's blocks are of a kind -- the round
                // wakes as many waves as the previous block would need from here, a quarter more; a longer block just takes another round.
                uint32_t awake = kT;
                if (last_block_bits) {
                    const uint64_t so_far = pos - block_begin;
                    const uint64_t left = last_block_bits > so_far ? last_block_bits - so_far : 0;
                    const uint64_t want = (uint64_t)rel0 + left + (left >> 2) + 4u * kSubBits;
                    awake = want / kSubBits >= (uint64_t)kT ? (uint32_t)kT : (((uint32_t)(want / kSubBits) + 64u) & ~63u);
                    if (awake > (uint32_t)kT) awake = kT;
                }
                Lane L{ (uint32_t)t * kSubBits, (uint32_t)t * kSubBits, 0, 0, 0 };
                if ((uint32_t)t < awake) lane_trace<false>(S, long_lit, long_dist, L, t == 0 ? rel0 : my_base, my_base);
                // Every lane's state goes to LDS; then, turn by turn: a lane whose predecessor leaves somewhere else than the lane begins
                // -- and not on a bit of its map -- is a job; the jobs are walked by the first threads of the workgroup, packed (a
                // turn late in the round has a handful of jobs: one wave walks them, the other fifteen wait at the barrier instead of
                // each running the loop for one or two of its lanes).  Lanes behind the first stop wait until it is settled.
                if (t == 0) { S.ctrl[C_JOBS0] = 0; S.ctrl[C_STOP0] = kT; }
                S.lane_exit[t] = L.exit | L.flags << 24;
                S.lane_start[t] = L.start;
                S.lane_map[t] = make_uint4((uint32_t)L.first, (uint32_t)(L.first >> 32), (uint32_t)L.count, (uint32_t)(L.count >> 32));
                __syncthreads();
                PROF(P_SWEEP0);
                const uint32_t last_lane = awake < (uint32_t)kT ? awake : (uint32_t)(kT - 1);   // lanes at and behind it are not part of this round
                uint32_t first_stop = kT;                                              // the first lane of the chain with a stop (lanes behind it do not matter)
                for (int turn = 0; ; ++turn) {
                    PROF_COUNT(P_N_SWEEPS, 1);
                    // (three slots in turn for the job count / first stop: thread 0 prepares the next turn's, last read two barriers ago)
                    const int slot = turn % 3, next_slot = (turn + 1) % 3;
                    if (t == 0) { S.ctrl[C_JOBS0 + next_slot] = 0; S.ctrl[C_STOP0 + next_slot] = kT; }
                    const uint32_t prev = t ? S.lane_exit[t - 1] : rel0, mine = S.lane_exit[t];
                    bool job = t > 0 && (uint32_t)t < last_lane && !(prev >> 24) && prev != S.lane_start[t] && (uint32_t)t <= first_stop;     // (a stopped predecessor moves nobody)
                    if (job) {
                        const uint32_t rel = prev - my_base, w = rel >> 5;             // (>= 0: a predecessor leaves at or beyond its end)
                        if (w < 8u) {
                            const uint4 m = S.lane_map[t];
                            const uint64_t first = (uint64_t)m.y << 32 | m.x, count = (uint64_t)m.w << 32 | m.z;
                            if (((uint32_t)(first >> (6u * w)) & 63u) == (32u | (rel & 31u))) {      // on the known chain already: it begins later, that is all
                                const uint64_t keep = ~0ull << (6u * w), f = first & keep, c = count & keep;
                                S.lane_map[t] = make_uint4((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)c, (uint32_t)(c >> 32));
                                S.lane_start[t] = prev;
                                job = false;
                            }
                        }
                    }
                    const uint64_t jobs = __ballot(job);
                    if (jobs) {
                        uint32_t at = 0;
                        if (lane_in_wave == 0) at = atomicAdd(&S.ctrl[C_JOBS0 + slot], (uint32_t)__popcll(jobs));
                        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
                        if (job) S.jobs[at + (uint32_t)__popcll(jobs & ((1ull << lane_in_wave) - 1ull))] = (uint16_t)t;
                    }
                    const uint64_t stopped = __ballot((mine >> 24) != 0 && (uint32_t)t <= first_stop && (uint32_t)t < last_lane);
                    if (stopped && lane_in_wave == 0) atomicMin(&S.ctrl[C_STOP0 + slot], (uint32_t)(wave * 64) + (uint32_t)__builtin_ctzll(stopped));
                    __syncthreads();
                    PROF(P_S_DETECT);
                    const uint32_t njobs = S.ctrl[C_JOBS0 + slot], stop_now = S.ctrl[C_STOP0 + slot];
                    // settled: nobody has to walk, and the lanes that were held back this turn are the ones that will be held back for good
                    // (a stop that has just dissolved, or moved, lets other lanes speak up next turn)
                    const bool settled = !njobs && stop_now == first_stop;
                    first_stop = stop_now;
                    if (settled) break;
                    // (every turn settles at least the first unsettled lane, and a stop moves at most once per settled lane: 2 kT turns are the
                    // proven bound.  A kernel that spins is worse than a stream turned away: beyond twice that the stream counts as corrupt.)
                    if (turn > 4 * kT) { err = E_INPUT; break; }
                    if ((uint32_t)t < njobs) {
                        const uint32_t lane = S.jobs[t];
                        const uint4 m = S.lane_map[lane];
                        const uint32_t ex = S.lane_exit[lane];
                        Lane J{ S.lane_start[lane], ex & 0xFFFFFFu, ex >> 24, (uint64_t)m.y << 32 | m.x, (uint64_t)m.w << 32 | m.z };
                        lane_trace<true>(S, long_lit, long_dist, J, S.lane_exit[lane - 1] & 0xFFFFFFu, lane * kSubBits);
                        S.lane_exit[lane] = J.exit | J.flags << 24;
                        S.lane_start[lane] = J.start;
                        S.lane_map[lane] = make_uint4((uint32_t)J.first, (uint32_t)(J.first >> 32), (uint32_t)J.count, (uint32_t)(J.count >> 32));
                    }
                    __syncthreads();
                    PROF(turn == 0 ? P_S_TURN1 : P_SWEEPS);
                    PROF_COUNT(P_N_MATCHES, njobs);
                }
                if (err) break;
                { const uint4 m = S.lane_map[t]; L.start = S.lane_start[t]; L.first = (uint64_t)m.y << 32 | m.x; L.count = (uint64_t)m.w << 32 | m.z; }
                PROF(P_SWEEPS);
                uint32_t nvalid = first_stop < last_lane ? first_stop + 1u : last_lane;      // lanes 0 .. nvalid - 1 form the chain (never the window's last lane)
                // ---- slots for the tokens, the round cut where the list would overflow
                if (t == 0) S.ctrl[C_CUT] = kT;
                uint32_t ntok = 0;
                if ((uint32_t)t < nvalid) {
                    #pragma unroll
                    for (int w = 0; w < 8; ++w) ntok += (uint32_t)(L.count >> (6 * w)) & 63u;
                }
                const uint32_t tok_incl = block_inclusive_sum(S, ntok, S.cum_tok);
                if ((uint32_t)t < nvalid && tok_incl > (uint32_t)kTokCap) atomicMin(&S.ctrl[C_CUT], (uint32_t)t);
                __syncthreads();
                const bool was_cut = S.ctrl[C_CUT] < nvalid;
                if (was_cut) nvalid = S.ctrl[C_CUT];                                  // >= 1: one lane alone always fits
                const uint32_t last = S.lane_exit[nvalid - 1];
                const uint32_t last_exit = last & 0xFFFFFFu, last_flags = was_cut ? 0u : last >> 24;
                err = S.ctrl[C_ERR];
                if (!err && (last_flags & F_BAD)) err = E_CODE;
                if (!err && wbase * 8u + last_exit > src_bits) err = E_INPUT;
                PROF(P_SCAN);
                if (err) break;
                prefetch(wbase * 8u + last_exit);                                     // the next round's (or the next header's) window
                // ---- the tokens themselves, and where every lane's bytes go
                uint32_t out = 0;
                if ((uint32_t)t < nvalid) out = lane_emit(S, long_lit, long_dist, L, my_base, toks + (tok_incl - ntok), (uint32_t)t);
                const uint32_t out_incl = block_inclusive_sum(S, out, S.cum_out0 + 1);       // (its barriers also put the tokens in front of their readers)
                PROF(P_WRITE);
                // ---- bytes: tiles of at most kNewMax bytes, beginning and ending anywhere (offsets below count from the round's first byte)
                const uint32_t round_out = S.cum_out0[nvalid], produced0 = produced;
                for (uint32_t o0 = 0; o0 < round_out && !err; ) {
                    const uint32_t total = round_out - o0 < (uint32_t)kNewMax ? round_out - o0 : (uint32_t)kNewMax, o1 = o0 + total;
                    if (t == 0) { S.ctrl[C_CUT] = kT; S.ctrl[C_LAST] = 0; S.ctrl[C_NLONG] = 0; S.ctrl[C_OPEN0] = 0; }
                    __syncthreads();
                    {   // (one pair of atomics per wave, not per lane: several hundred lanes on two LDS words cost a tile 11 000 cycles)
                        const uint64_t mine = __ballot((uint32_t)t < nvalid && out_incl > o0 && out_incl - out < o1);
                        if (mine && lane_in_wave == 0) {
                            atomicMin(&S.ctrl[C_CUT], (uint32_t)(wave * 64) + (uint32_t)__builtin_ctzll(mine));
                            atomicMax(&S.ctrl[C_LAST], (uint32_t)(wave * 64) + 63u - (uint32_t)__builtin_clzll(mine));
                        }
                    }
                    __syncthreads();
                    const uint32_t a = S.ctrl[C_CUT], b = S.ctrl[C_LAST];             // the lanes with bytes in the tile: all their tokens are looked at
                    PROF(P_F_SETUP);
                    const uint32_t tok0 = a ? S.cum_tok[a - 1] : 0u, tok1 = S.cum_tok[b];
                    const bool sink = produced >= it.dst_cap;                         // the caller's buffer is full: decode on, write nothing
                    if (!sink) {
                        // every byte a literal until a match says otherwise
                        for (uint32_t u = (uint32_t)t; u < (total + 7u) >> 3; u += kT) {
                            const uint32_t self0 = u * 8u + kHist, s0 = self0 | (self0 + 1u) << 16;
                            *reinterpret_cast<u32x4*>(&S.from[u * 8u]) = u32x4{ s0, s0 + 0x00020002u, s0 + 0x00040004u, s0 + 0x00060006u };
                        }
                        __syncthreads();
                    }
                    PROF(P_F_INIT);
                    if (!sink || produced0 < (uint32_t)kHist) {
                        for (uint32_t kb = tok0; kb < tok1; kb += 4u * kT) {             // (the same trips for every thread: the neighbours' tokens travel by DPP)
                            uint2 four[4];                                             // four loads in flight (an absent token reads as an empty literal)
                            #pragma unroll
                            for (int u = 0; u < 4; ++u) { const uint32_t k = kb + (uint32_t)t + (uint32_t)u * kT; four[u] = k < tok1 ? toks[k] : make_uint2(0u, 0u); }
                            uint32_t lane_at[4];
                            #pragma unroll
                            for (int u = 0; u < 4; ++u) lane_at[u] = S.cum_out0[(four[u].y >> 16) & 0x3FFu];      // (four look-ups in flight)
                            #pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const uint2 tk = four[u];
                                const uint32_t at = lane_at[u] + (tk.y & 0xFFFFu);          // the token's first byte
                                const bool match = tk.y >> 31;
                                const uint32_t len = tk.x & 0xFFFFu, dist = tk.x >> 16;
                                // A match right behind a match of the same distance goes on with its period (a run cut into 258-byte pieces,
                                // the pixels of a flat row): all of them copy from in front of the FIRST one, so the pieces of a run are one
                                // link deep instead of one link per piece.  The first one is looked for among the wave's 64 tokens.
                                if (!__ballot(match)) {                                 // a wave of literals: nothing but one or two bytes into the ring
                                    if (!sink) {
                                        const uint32_t n = tk.x & 3u, rel = at - o0;       // (rel >= total: in front of or behind the tile)
                                        if (n && rel < total) S.ring[(produced + rel) & kRingMask] = (uint8_t)(tk.x >> 16);
                                        if (n > 1u && rel + 1u < total) S.ring[(produced + rel + 1u) & kRingMask] = (uint8_t)(tk.x >> 24);
                                    }
                                    continue;
                                }
                                bool goes_on = false;
                                uint32_t first = at + 1u;                                // -> the nearest token in front that does not go on, + 1
                                {
                                    const uint32_t before_x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tk.x, 0x138, 0xF, 0xF, false);      // wave_shr:1 (lane 0: no token)
                                    const uint32_t before_y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tk.y, 0x138, 0xF, 0xF, false);
                                    goes_on = match && (before_y >> 31) && (before_x >> 16) == dist;
                                    first = goes_on ? 0u : at + 1u;
                                    if (__ballot(goes_on)) {
#define RUN_STEP(CTRL, ROWS) { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)first, CTRL, ROWS, 0xF, false); first = o > first ? o : first; }
                                        RUN_STEP(0x111, 0xF) RUN_STEP(0x112, 0xF) RUN_STEP(0x114, 0xF) RUN_STEP(0x118, 0xF) RUN_STEP(0x142, 0xA) RUN_STEP(0x143, 0xC)
#undef RUN_STEP
                                    }
                                }
                                const uint32_t root = goes_on ? first - 1u : at;          // where the run begins
                                // Byte x of a run of period dist equals byte p - dist + (x - p) mod dist for any point p of the run at or in front
                                // of x: p = the run's first byte, or the tile's if the run began in front of it (the dist bytes in front of p
                                // are bytes of this tile or window history either way).  Pointing there at once keeps the chains of
                                // overlapping copies one link long.  Long matches are left to a whole wave (their list slots: one atomic per wave).
                                const bool too_far = match && produced0 < (uint32_t)kHist && dist > produced0 + at;      // reaches before the first output byte
                                if (too_far) atomicMax(&S.ctrl[C_ERR], (uint32_t)E_DISTANCE);
                                const bool inside = match && !too_far && !sink && at < o1 && at + len > o0;
                                const uint32_t p = root > o0 ? root : o0;
                                const uint32_t x0 = at > o0 ? at : o0, x1 = at + len < o1 ? at + len : o1;      // the bytes of the match inside the tile
                                const uint32_t q = p - o0 + kHist - dist, since = x0 - p, n = x1 - x0;
                                const bool is_long = inside && n >= (uint32_t)kLongMin;
                                const uint64_t longs = __ballot(is_long);
                                uint32_t slot = kLongCap;
                                if (longs) {
                                    const int leader = __builtin_ctzll(longs);
                                    uint32_t base = 0;
                                    if (lane_in_wave == leader) base = atomicAdd(&S.ctrl[C_NLONG], (uint32_t)__popcll(longs));
                                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
                                    if (is_long) slot = base + (uint32_t)__popcll(longs & ((1ull << lane_in_wave) - 1ull));
                                }
                                if (inside) {
                                    if (slot < (uint32_t)kLongCap) S.longm[slot] = make_uint2((x0 - o0) | n << 15, since | dist << 16);
                                    else for (uint32_t i = 0, m = since % dist; i < n; ++i) { S.from[x0 - o0 + i] = (uint16_t)(q + m); if (++m == dist) m = 0; }
                                } else if (!match && !sink) {
                                    const uint32_t nl = tk.x & 3u;
                                    if (nl && at >= o0 && at < o1) S.ring[(produced0 + at) & kRingMask] = (uint8_t)(tk.x >> 16);
                                    if (nl > 1u && at + 1u >= o0 && at + 1u < o1) S.ring[(produced0 + at + 1u) & kRingMask] = (uint8_t)(tk.x >> 24);
                                }
                            }
                        }
                        __syncthreads();
                        err = S.ctrl[C_ERR];
                    }
                    PROF(P_STORED);                                                    // (the slot of the stored blocks doubles as "tokens into the tile")
                    if (!sink && !err) {
                        expand_long_matches(S);
                        __syncthreads();
                        PROF(P_R_EXPAND);
                        { const int rounds = resolve_copies(S, produced, total, prof); (void)rounds; PROF_COUNT(P_N_ROUNDS, rounds); }
                        __syncthreads();
                        PROF(P_MATCH);
                        flush(produced, total);
                        __syncthreads();
                        PROF(P_FLUSH);
                    }
                    produced = produced + total < produced ? 0xFFFFFFFFu : produced + total;           // (saturating: only its size matters beyond 32 KiB)
                    o0 = o1;
                }
                pos = wbase * 8u + last_exit;
                if (last_flags & F_EOB) { in_block = false; last_block_bits = pos - block_begin; }
            }
        }
        if (suspended) break;                                                       // (inside the block: it goes on next launch)
        if (bfinal) done = true;
    }
    if (state) {                                      // where the next launch goes on
        __syncthreads();
        if (suspended_in_block) for (int k = t; k < 320; k += kT) state->lens[k] = S.lens[k];
        if (t == 0) {
            state->pos = pos; state->produced = produced; state->blocks = blocks; state->err = err;
            state->flags = ST_VALID | (done ? ST_DONE : 0u) | (suspended_in_block ? ST_IN_BLOCK : 0u);
            state->btype = S.ctrl[C_BTYPE]; state->bfinal = S.ctrl[C_FINAL]; state->hlit = S.ctrl[C_HLIT]; state->hdist = S.ctrl[C_HDIST];
        }
    }
    PROF_FLUSH;
    __syncthreads();                                  // every thread's token reads are done: the list may change hands
    if (t == 0) {
        out_len[blockIdx.x] = produced < it.dst_cap ? produced : it.dst_cap;
        status[blockIdx.x] = err;
        __hip_atomic_store(&slot_busy[my_slot], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

} // namespace

// The per-(thread, device, stream) scratch of the launches: token lists (256 KiB per stream), the streams' states between the launches
// of a sliced call, the descriptors and the bytes-there table.  Calls on one stream are ordered, calls on different streams never
// share it; growing it waits for its own stream only.
namespace {
struct InflateScratch { uint2* toks; uint32_t* busy; uint32_t n_slots; InfState* states; InfItem* items; uint32_t* avail; };
int inflate_scratch(int count, hipStream_t stream, InflateScratch& out)
{
    // per call: the attribute belongs to the current device's copy of the kernel, and a process may drive several devices
    if (hipFuncSetAttribute((const void*)k_inflate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Shared)) != hipSuccess) return set_error(GAMUT_HIP_ERR_HIP, "inflate: %zu bytes of LDS are not available", sizeof(Shared));
    struct StreamTable { int device; hipStream_t stream; void* p; size_t cap; uint32_t n_slots; uint64_t used; };
    static thread_local std::vector<StreamTable> tables;
    static thread_local uint64_t tick = 0;
    StreamTable* e = nullptr;
    const int device = current_device();                    // (the null stream is one handle for every device)
    for (StreamTable& c : tables) if (c.stream == stream && c.device == device) { e = &c; break; }
    if (!e) {
        // A caller that makes a stream per batch would add an entry per batch: beyond 8 entries the least recently used ones go.  Their
        // stream handles may be dead by now, so nothing stream-specific can be asked of them: the device is drained once instead.
        if (tables.size() >= 8) {
            std::sort(tables.begin(), tables.end(), [](const StreamTable& a, const StreamTable& b) { return a.used > b.used; });
            while (tables.size() > 4) {                        // (an entry's memory belongs to ITS device: that is the one to drain)
                if (tables.back().p) { (void)hipSetDevice(tables.back().device); (void)hipDeviceSynchronize(); (void)hipFree(tables.back().p); }
                tables.pop_back();
            }
            (void)hipSetDevice(device);
        }
        tables.push_back(StreamTable{ device, stream, nullptr, 0, 0, 0 });
        e = &tables.back();
    }
    e->used = ++tick;
    // token lists: one per workgroup that can be resident (one per compute unit: the kernel takes 157 KB of LDS), twice that for slack
    // -- 512 x 256 KiB = 128 MiB on MI355X however many streams the batch has (a list per stream was 2.5 GB for 8192 PNG files)
    static thread_local PerDevice<int> cus_pd;
    int& cus = cus_pd.cur();
    if (!cus) { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) n = 256; cus = n; }
    const uint32_t resident = (uint32_t)cus;
    const uint32_t n_slots = (uint32_t)count < 2 * resident ? (uint32_t)count : 2 * resident;
    auto round_up = [](size_t n) { return (n + 255) & ~(size_t)255; };
    const size_t tok_bytes = (size_t)(2 * resident) * kTokCap * sizeof(uint2), busy_bytes = round_up((size_t)(2 * resident) * 4),
                 state_bytes = round_up((size_t)count * sizeof(InfState)),
                 item_bytes = round_up((size_t)count * sizeof(InfItem)), avail_bytes = round_up((size_t)count * 4);
    // (the tables lie BEHIND the token lists the entry already has, which may be more than this batch needs: that is what must fit)
    const size_t need_toks = (size_t)(n_slots > e->n_slots ? n_slots : e->n_slots) * kTokCap * sizeof(uint2);
    const size_t bytes = need_toks + busy_bytes + state_bytes + item_bytes + avail_bytes;
    if (bytes > e->cap || n_slots > e->n_slots) {
        if (e->p) { (void)hipStreamSynchronize(stream); (void)hipFree(e->p); e->p = nullptr; e->cap = 0; e->n_slots = 0; }
        // the per-stream tables grow with the batch (+ a quarter); the token lists are sized once: for a full chip as soon as a batch has
        // more streams than a few, so that later, larger batches do not reallocate
        const uint32_t slots_now = n_slots > 16 ? 2 * resident : n_slots;
        const size_t small = busy_bytes + state_bytes + item_bytes + avail_bytes;
        const size_t want = (slots_now == 2 * resident ? tok_bytes : (size_t)slots_now * kTokCap * sizeof(uint2)) + small + small / 4 + 4096;
        if (hipMalloc(&e->p, want) != hipSuccess) { (void)hipGetLastError(); e->p = nullptr; return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "inflate: %zu bytes of scratch are not available", want); }
        e->cap = want; e->n_slots = slots_now;
        // the flags start at zero and every workgroup puts its own back: cleared here only
        if (hipMemsetAsync(static_cast<uint8_t*>(e->p) + (size_t)slots_now * kTokCap * sizeof(uint2), 0, busy_bytes, stream) != hipSuccess) return set_error(GAMUT_HIP_ERR_HIP, "inflate: scratch clear failed");
    }
    uint8_t* const base = static_cast<uint8_t*>(e->p);
    const size_t toks_now = (size_t)e->n_slots * kTokCap * sizeof(uint2);
    out.toks = reinterpret_cast<uint2*>(base);
    out.busy = reinterpret_cast<uint32_t*>(base + toks_now);
    out.n_slots = e->n_slots;
    out.states = reinterpret_cast<InfState*>(base + toks_now + busy_bytes);
    out.items = reinterpret_cast<InfItem*>(base + toks_now + busy_bytes + state_bytes);
    out.avail = reinterpret_cast<uint32_t*>(base + toks_now + busy_bytes + state_bytes + item_bytes);
    return GAMUT_HIP_OK;
}
int inflate_items(const gamut_hip_inflate_desc* descs, int count, std::vector<InfItem>& items)
{
    static_assert(sizeof(InfItem) == sizeof(gamut_hip_inflate_desc), "descriptor layout");
    items.resize((size_t)count);                               // pageable: the upload has read it when hipMemcpyAsync returns
    for (int i = 0; i < count; ++i) {
        if (!descs[i].src || (!descs[i].dst && descs[i].dst_cap)) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "inflate: stream %d: null pointer", i);
        items[(size_t)i] = InfItem{ descs[i].src, descs[i].dst, descs[i].src_len, descs[i].dst_cap };
    }
    return GAMUT_HIP_OK;
}
} // namespace

int inflate_launch(const gamut_hip_inflate_desc* descs, int count, uint32_t* out_len_dev, uint32_t* status_dev, hipStream_t stream)
{
    std::vector<InfItem> items;
    if (int rc = inflate_items(descs, count, items)) return rc;
    InflateScratch sc;
    if (int rc = inflate_scratch(count, stream, sc)) return rc;
    // the lists' busy flags: every workgroup puts its own back, but a launch that died half way (a fault the process survived) would
    // leave some set for ever -- fewer lists for the launches after it on this stream, or a spin without end.  2 KB, stream-ordered.
    GAMUT_HIP_CHECK(hipMemsetAsync(sc.busy, 0, (size_t)sc.n_slots * 4, stream));
    GAMUT_HIP_CHECK(hipMemcpyAsync(sc.items, items.data(), items.size() * sizeof(InfItem), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_inflate, dim3((unsigned)count), dim3(kT), sizeof(Shared), stream, (const InfItem*)sc.items, count, sc.toks, sc.busy, sc.n_slots, out_len_dev, status_dev, (InfState*)nullptr, (const uint32_t*)nullptr);
    return launch_status("inflate");
}

// A sliced call: the streams are still on their way up.  begin once (descriptors up, states cleared), then one step per slice -- the
// caller has made `stream` wait for the upload of the bytes avail_host[i] promises -- the last one with every stream whole.
int inflate_sliced_begin(const gamut_hip_inflate_desc* descs, int count, hipStream_t stream)
{
    std::vector<InfItem> items;
    if (int rc = inflate_items(descs, count, items)) return rc;
    InflateScratch sc;
    if (int rc = inflate_scratch(count, stream, sc)) return rc;
    // the lists' busy flags: every workgroup puts its own back, but a launch that died half way (a fault the process survived) would
    // leave some set for ever -- fewer lists for the launches after it on this stream, or a spin without end.  2 KB, stream-ordered.
    GAMUT_HIP_CHECK(hipMemsetAsync(sc.busy, 0, (size_t)sc.n_slots * 4, stream));
    GAMUT_HIP_CHECK(hipMemcpyAsync(sc.items, items.data(), items.size() * sizeof(InfItem), hipMemcpyHostToDevice, stream));
    GAMUT_HIP_CHECK(hipMemsetAsync(sc.states, 0, (size_t)count * sizeof(InfState), stream));
    return GAMUT_HIP_OK;
}
int inflate_sliced_step(int count, const uint32_t* avail_host /* pinned or pageable: read before the call returns */, uint32_t* out_len_dev, uint32_t* status_dev, hipStream_t stream)
{
    InflateScratch sc;
    if (int rc = inflate_scratch(count, stream, sc)) return rc;     // (the buffer of _begin: same thread, device, stream, count)
    GAMUT_HIP_CHECK(hipMemcpyAsync(sc.avail, avail_host, (size_t)count * 4, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_inflate, dim3((unsigned)count), dim3(kT), sizeof(Shared), stream, (const InfItem*)sc.items, count, sc.toks, sc.busy, sc.n_slots, out_len_dev, status_dev, sc.states, (const uint32_t*)sc.avail);
    return launch_status("inflate");
}

} // namespace gamut

using namespace gamut;

#if INFLATE_PROFILE
extern "C" int gamut_hip_inflate_profile(unsigned long long* out, int reset)      // measurement builds only
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_inflate_prof), sizeof(unsigned long long) * P_N) != hipSuccess) return 1;
    if (reset) { unsigned long long z[P_N] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_inflate_prof), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

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

// The same in steps: every launch sees `slice_bytes` more of every stream than the one before (all of them are in HBM already here; the
// PNG batch path runs the launches behind the upload, slice by slice) -- a stream stops in front of a round it cannot finish within the
// bytes it may read and goes on in the next launch from its saved state.  Same results as gamut_hip_inflate_batch_device.
extern "C" int gamut_hip_inflate_batch_device_sliced(const gamut_hip_inflate_desc* descs, int count, uint32_t* out_len_dev, uint32_t* status_dev, uint32_t slice_bytes, void* stream)
{
    clear_error();
    if (count < 0 || (count > 0 && (!descs || !out_len_dev || !status_dev)) || !slice_bytes) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "inflate_batch_device_sliced: bad arguments");
    if (count == 0) return GAMUT_HIP_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_error(GAMUT_HIP_ERR_NO_DEVICE, "no HIP device available (libgamut_hip has no CPU fallback)");
    try {
        hipStream_t st = pick_stream(stream);
        if (int rc = inflate_sliced_begin(descs, count, st)) return rc;
        uint32_t longest = 0;
        for (int i = 0; i < count; ++i) longest = descs[i].src_len > longest ? descs[i].src_len : longest;
        std::vector<std::vector<uint32_t>> tables;                  // (one per launch, alive until the stream has been waited for)
        for (uint64_t upto = slice_bytes; ; upto += slice_bytes) {
            tables.emplace_back((size_t)count);
            std::vector<uint32_t>& avail = tables.back();
            for (int i = 0; i < count; ++i) avail[(size_t)i] = (uint32_t)(upto < descs[i].src_len ? upto : descs[i].src_len);
            if (int rc = inflate_sliced_step(count, avail.data(), out_len_dev, status_dev, st)) { (void)hipStreamSynchronize(st); return rc; }
            if (upto >= longest) break;
        }
        GAMUT_HIP_CHECK(hipStreamSynchronize(st));                   // (the tables go away with this frame; the other entry point is asynchronous)
        return GAMUT_HIP_OK;
    } catch (...) { return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "inflate_batch_device_sliced: out of host memory"); }
}
