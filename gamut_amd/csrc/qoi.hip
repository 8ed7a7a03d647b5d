// qoi.hip -- QOI decode (needed for the mixed JPEG/PNG/QOI batch of BASELINE.json config 5).
//
// Replaces qoi_decode (source/gamut/codecs/qoi.d:448-550): one workgroup per stream, see k_qoi_decode.  The host only
// validates the 14-byte header (same checks as :472-480) and uploads the streams as they are.
#include "common.hpp"
#include <atomic>
#include <type_traits>
#include <vector>

namespace gamut {
namespace {

constexpr uint32_t kQoiMagic = 0x716F6966u, kQoiPixelsMax = 400000000u;      // qoi.d:244, :251
constexpr int kQoiHeader = 14, kQoiPadding = 8;                                // :245, :268

struct QoiItem { uint64_t begin; int64_t out_off; uint32_t size, npx; int32_t channels, pad; };

// One WORKGROUP of four waves per stream.  The format is a byte-serial state machine (previous pixel, 64-entry colour hash, run
// counter), but only one link of the chain is really serial -- a QOI_OP_INDEX needs the hash table as all earlier pixels left
// it.  The rest is parallel over the 256 lanes, 2 KiB of the stream at a time:
//   A. op boundaries: an op's length follows from its first byte alone, so each lane walks its own 8 bytes for every possible
//      entry offset 0..4 (an op of the previous lane spills at most 4 bytes) with plain bit operations; the five exit offsets
//      are a map {0..4} -> {0..4} packed in 15 bits, maps compose, and a scan over the lanes (DPP inside a wave, LDS across the
//      four) gives every lane its true entry offset; the op starts are compacted into a list in LDS;
//   B. 64 ops at a time, one per lane, the groups dealt to the four waves: an op is a bytewise function x -> (x & ~M) + V of
//      the previous pixel (DIFF / LUMA / RUN: M = 0, V = deltas; RGB / RGBA: M = the bytes set, V = their values); such
//      functions compose, so an inclusive scan over the lanes (DPP row shifts / broadcasts, no LDS) gives every op's pixel
//      relative to the pixel before the group.  An INDEX op is "all bytes set to an unknown U"; the scan carries which bytes
//      still hang on the most recent INDEX.  Run lengths are prefix-summed in the same steps.  Results go to LDS;
//   C. wave 0 then walks the groups in order -- the only serial part: pixels before the first INDEX op of a group at once, the
//      INDEX ops one after the other (wave-uniform loop): the slot's value is the newest earlier pixel of the group with that
//      hash (ballot + readlane), else the table as it stood before the group (a register per slot, lane s = slot s); the lanes
//      up to the next INDEX op then get their pixels and hashes;
//   D. the table takes the group's pixels with one LDS ds_max_u64 per lane on (op number << 32 | pixel): the newest writer of
//      a slot wins without a second pass; pixels go to an LDS buffer by run length and leave in 1 KiB rows.
// W = 4 waves per stream halves a stream's time (18.6 instead of 31.6 ms per 1080p stream) but costs more instructions in all
// (every wave scans, one resolves): batches of more streams than the chip has SIMDs to spare run W = 1, one wave per stream.
constexpr int kQoiWin = 2048;                                                       // window bytes
// W = 1: pixels wait in an LDS buffer until whole 1 KiB rows can leave.  A group of 64 ops makes up to 64 * 62 = 3968 pixels, but
// only runs do that: a group of more than kQoiBufGroup pixels writes them to the image itself, and the buffer stays small -- the
// workgroup's LDS (10 KB instead of 24) is what decides how many streams a compute unit holds at once: 2730 streams (BASELINE.json
// config 5 on one GPU) are all resident instead of taking two rounds.
constexpr int kQoiBufGroup = 512, kQoiOutCap = 256 + kQoiBufGroup + 64;
constexpr int kQoiWideBelow = 768;                                                  // streams per launch below which W = 4
constexpr int kQoiSlack = GAMUT_HIP_QOI_SLACK;                                      // readable bytes guaranteed after every stream
constexpr uint32_t kQoiMapId = 0u | 1u << 3 | 2u << 6 | 3u << 9 | 4u << 12;         // the identity of the exit-offset maps

__device__ __forceinline__ uint32_t qoi_add_bytes(uint32_t x, uint32_t y)       // bytewise (x + y) mod 256
{
    return ((x & 0x7f7f7f7fu) + (y & 0x7f7f7f7fu)) ^ ((x ^ y) & 0x80808080u);
}
__device__ __forceinline__ uint32_t qoi_hash(uint32_t px) { return __builtin_amdgcn_udot4(px, 0x0B070503u, 0u, false) & 63u; }   // QOI_COLOR_HASH :239-242

// value of the lane CTRL names (row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143, wave_shr:1 = 0x138); lanes
// without a source, or in rows outside ROWMASK, get `idle`
template <int CTRL, int ROWMASK> __device__ __forceinline__ uint32_t qoi_dpp(uint32_t v, uint32_t idle = 0)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)idle, (int)v, CTRL, ROWMASK, 0xF, false);
}
__device__ __forceinline__ uint32_t qoi_readlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
// LDS hand-offs inside one wave: its LDS operations execute in program order, this only stops the compiler from moving them
__device__ __forceinline__ void qoi_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// exit-offset maps: entry offset e (0..4) of a stretch of bytes -> entry offset of the stretch after it, 3 bits each
__device__ __forceinline__ uint32_t qoi_map_apply(uint32_t m, uint32_t e) { return (m >> (3u * e)) & 7u; }
__device__ __forceinline__ uint32_t qoi_map_then(uint32_t a, uint32_t b)          // a first, then b
{
    uint32_t r = 0;
    #pragma unroll
    for (int k = 0; k < 5; ++k) r |= qoi_map_apply(b, qoi_map_apply(a, (uint32_t)k)) << (3 * k);
    return r;
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ void qoi_map_scan_step(uint32_t& m) { m = qoi_map_then(qoi_dpp<CTRL, ROWMASK>(m, kQoiMapId), m); }
// The inclusive scan of `then` over the lanes of a wave (lanes 0 .. this one, composed).  Composing two 15-bit maps is five dependent
// extract-shift-extract chains, ~25 instructions, six times per window; with a map held as BYTES (entries 0..3 in one register, entry 4 in
// another) it is a table look-up the hardware has: v_perm_b32 picks, per selector byte 0..7, a byte of its two sources -- `a then b` is
// perm({b4, b3 b2 b1 b0}, selectors = a's bytes): two permutes per step.  (Bytes 1-3 of the second register carry whatever the look-up puts
// there -- always some entry of a map, 0..4: valid selectors, never read as an entry.)
__device__ __forceinline__ void qoi_map_scan64(uint32_t& map)
{
    uint32_t lo = (map & 7u) | (map >> 3 & 7u) << 8 | (map >> 6 & 7u) << 16 | (map >> 9 & 7u) << 24, hi = map >> 12 & 7u;
#define GAMUT_QOI_MAP_STEP(CTRL, ROWMASK) { const uint32_t pl = qoi_dpp<CTRL, ROWMASK>(lo, 0x03020100u), ph = qoi_dpp<CTRL, ROWMASK>(hi, 4u); \
                                            const uint32_t nl = __builtin_amdgcn_perm(hi, lo, pl), nh = __builtin_amdgcn_perm(hi, lo, ph); lo = nl; hi = nh; }
    GAMUT_QOI_MAP_STEP(0x111, 0xF) GAMUT_QOI_MAP_STEP(0x112, 0xF) GAMUT_QOI_MAP_STEP(0x114, 0xF) GAMUT_QOI_MAP_STEP(0x118, 0xF)
    GAMUT_QOI_MAP_STEP(0x142, 0xA) GAMUT_QOI_MAP_STEP(0x143, 0xC)
#undef GAMUT_QOI_MAP_STEP
    map = (lo & 7u) | (lo >> 8 & 7u) << 3 | (lo >> 16 & 7u) << 6 | (lo >> 24 & 7u) << 9 | (hi & 7u) << 12;
}
// n += the lane CTRL names (nothing for lanes without a source, or in rows outside ROWMASK): one v_add_u32_dpp in place.  Through
// the builtin the compiler zeroes a temporary, moves into it and adds: three instructions, six times per scan.  (s_nop 1: a DPP
// read of a register the previous vector instruction wrote needs two wait states, and the compiler does not look into asm.)
template <int CTRL, int ROWMASK> __device__ __forceinline__ void qoi_add_scan_step(uint32_t& n)
{
    if constexpr (CTRL == 0x111) asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(n));
    else if constexpr (CTRL == 0x112) asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(n));
    else if constexpr (CTRL == 0x114) asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(n));
    else if constexpr (CTRL == 0x118) asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(n));
    else if constexpr (CTRL == 0x142 && ROWMASK == 0xA) asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(n));
    else if constexpr (CTRL == 0x143 && ROWMASK == 0xC) asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(n));
    else n += qoi_dpp<CTRL, ROWMASK>(n);
}
// the value of the lane CTRL names, 0 where there is none: for the row shifts (every row takes part) bound_ctrl supplies the zeros
// and no temporary has to be cleared first
template <int CTRL, int ROWMASK> __device__ __forceinline__ uint32_t qoi_dpp0(uint32_t v)
{
    if constexpr (ROWMASK == 0xF && CTRL >= 0x111 && CTRL <= 0x11F)
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
    else
        return qoi_dpp<CTRL, ROWMASK>(v);
}

struct QoiFn { uint32_t M, V, U; };       // x -> ((x & ~M) | (index value & U)) + V bytewise; U is a subset of M
__device__ __forceinline__ QoiFn qoi_then(QoiFn a, QoiFn b)                       // a first, then b
{
    QoiFn r; r.M = a.M | b.M; r.V = qoi_add_bytes(a.V & ~b.M, b.V); r.U = b.U | (a.U & ~b.M); return r;
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ void qoi_scan_step(QoiFn& f, uint32_t& n)
{
    const QoiFn p = { qoi_dpp0<CTRL, ROWMASK>(f.M), qoi_dpp0<CTRL, ROWMASK>(f.V), qoi_dpp0<CTRL, ROWMASK>(f.U) };   // (0, 0, 0) = identity
    f = qoi_then(p, f);
    qoi_add_scan_step<CTRL, ROWMASK>(n);
}

// ---- the same functions WITHOUT composing them (round 6) ----------------------------------------------------------------------
// qoi_scan_step composes (M, V, U) triples: per step three DPP moves, a bytewise add (five instructions: the part has no packed
// byte add) and the mask logic, 14 vector instructions x 6 steps -- 83 of the ~175 a group of 64 ops costs, and a batch's time
// is its instruction count.  But a group's functions have a simple shape: the colour bytes of op i are EITHER the deltas of ops
// 0..i added to the pixel before the group (no RGB / RGBA / INDEX op up to i), OR the value the most recent such op j set (from
// the stream, or from the index) plus the deltas of ops j+1..i.  Deltas add; so
//   * P = inclusive prefix sums of the deltas, as PLAIN integer sums: R and B in the 16-bit halves of one register, G in another
//     (64 values below 271 never carry out of a half; masked to bytes once, at the end) -- one v_add_u32_dpp per step and register;
//   * jm = 1 + the lane of the most recent absolute op (0: none) -- a v_max_u32_dpp per step;
//   * op i's value = A[jm] - P[jm] + P[i], A = the bytes an RGB / RGBA op carries (0 for INDEX: the index value is added when it
//     is known): the lane's own P, and Q = A - P fetched from lane jm - 1 by ds_bpermute_b32 (the crossbar, no memory).
// Four chains (with the run lengths') interleaved: a DPP read of a register written four instructions earlier needs no wait states.
// 24 + ~20 vector instructions instead of 83 + 12.  Alpha only changes at RGBA and INDEX ops: groups without either (six in ten of
// a photograph's) skip it, groups with INDEX ops only mark the lanes from the first one on, RGBA ops take a scan of their own.
// The result is the same (M, V, U) the composition gives, bit for bit (U a subset of M, V's alpha 0 unless an RGBA op set it).
#define GAMUT_QOI_SCAN4(CTRL) asm volatile("v_add_u32_dpp %0, %0, %0 " CTRL "\n\tv_add_u32_dpp %1, %1, %1 " CTRL "\n\tv_add_u32_dpp %2, %2, %2 " CTRL "\n\tv_max_u32_dpp %3, %3, %3 " CTRL \
                                           : "+v"(pe), "+v"(po), "+v"(n), "+v"(jm))
template <int CTRL, int ROWMASK> __device__ __forceinline__ void qoi_max_scan_step(uint32_t& v) { const uint32_t p = qoi_dpp<CTRL, ROWMASK>(v, 0u); v = p > v ? p : v; }
// lane: 0..63; dE = R | B << 16 and dO = G: the deltas of a DIFF / LUMA op (0 for every other op); m_stream = all ones for an RGB / RGBA
// op (bytes 1..4 of the op are v_abs); n: in = the op's run length, out = the inclusive prefix sum
__device__ __forceinline__ QoiFn qoi_group_scan(int lane, uint32_t dE, uint32_t dO, uint32_t m_stream, bool is_index, bool is_rgba, uint32_t v_abs, uint32_t& n)
{
    uint32_t pe = dE, po = dO, jm = (m_stream != 0u || is_index) ? (uint32_t)lane + 1u : 0u;
    asm volatile("s_nop 1");
    GAMUT_QOI_SCAN4("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    GAMUT_QOI_SCAN4("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    GAMUT_QOI_SCAN4("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    GAMUT_QOI_SCAN4("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    GAMUT_QOI_SCAN4("row_bcast:15 row_mask:0xa bank_mask:0xf");
    GAMUT_QOI_SCAN4("row_bcast:31 row_mask:0xc bank_mask:0xf");
    const uint32_t qE = (v_abs & m_stream & 0x00FF00FFu) - pe, qO = ((v_abs >> 8) & m_stream & 0xFFu) - po;
    const int addr = (int)(jm << 2) - 4;
    const bool none = jm == 0u;
    uint32_t fE = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)qE), fO = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)qO);
    fE = none ? 0u : fE; fO = none ? 0u : fO;
    QoiFn f;
    f.V = ((fE + pe) & 0x00FF00FFu) | ((fO + po) & 0xFFu) << 8;
    f.M = none ? 0u : 0x00FFFFFFu;
    f.U = 0u;
    const uint64_t bal_idx = __ballot(is_index), bal_rgba = __ballot(is_rgba);
    if (bal_idx | bal_rgba) {                                  // (wave-uniform)
        const uint32_t src_is_index = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, is_index ? 0x00FFFFFF : 0);
        f.U = none ? 0u : src_is_index;
        if (bal_rgba == 0ull) {                                // INDEX ops only: from the first one on, alpha is an index value's
            const bool behind = lane >= __builtin_ctzll(bal_idx);
            f.M |= behind ? 0xFF000000u : 0u;
            f.U |= behind ? 0xFF000000u : 0u;
        } else {                                               // the most recent op that sets alpha: an RGBA op's own byte, or an index value's
            uint32_t ja = (is_index || is_rgba) ? (uint32_t)lane + 1u : 0u;
            qoi_max_scan_step<0x111, 0xF>(ja); qoi_max_scan_step<0x112, 0xF>(ja); qoi_max_scan_step<0x114, 0xF>(ja); qoi_max_scan_step<0x118, 0xF>(ja);
            qoi_max_scan_step<0x142, 0xA>(ja); qoi_max_scan_step<0x143, 0xC>(ja);
            const uint32_t wa = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ja << 2) - 4, is_index ? 1 : (int)(v_abs & 0xFF000000u));
            if (ja != 0u) {
                f.M |= 0xFF000000u;
                if (wa & 1u) f.U |= 0xFF000000u; else f.V |= wa & 0xFF000000u;
            }
        }
    }
    return f;
}

#ifndef QOI_GROUP_SCAN            // 1: qoi_group_scan, 0: the composition scan (qoi_scan_step) -- the A/B knob (tools/variant.sh)
#define QOI_GROUP_SCAN 1
#endif
#ifndef QOI_PREFETCH              // 1: the one-wave kernel fetches a group's op offsets / bytes / table entries a group ahead.  Measured and left off:
#define QOI_PREFETCH 0            // 2730 streams 32.9 -> 33.6 ms (profiles/r06_qoi_prefetch_ab.txt) -- the waits it removes were covered by the SIMD's other waves
#endif
#ifndef QOI_DIRECT_RGBA           // tuning knob (tools/variant.sh): 0 = RGBA outputs through the LDS pixel buffer like RGB ones
#define QOI_DIRECT_RGBA 1
#endif
#ifndef QOI_PROFILE               // measurement only (tools/variant.sh qoi:prof:-DQOI_PROFILE=1): cycles per phase of wave 0, summed over streams
#define QOI_PROFILE 0
#endif
#if QOI_PROFILE
__device__ unsigned long long g_qoi_prof[8];
#define QPROF(slot) do { const unsigned long long now_ = clock64(); qprof[slot] += now_ - qprof_t0; qprof_t0 = now_; } while (0)
#else
#define QPROF(slot)
#endif

template <int kQoiWaves>
__global__ __launch_bounds__(kQoiWaves * 64) void k_qoi_decode(const QoiItem* items, int n_items, const uint8_t* blob, uint8_t* out)
{
    constexpr int kQoiT = kQoiWaves * 64, kQoiLaneBytes = kQoiWin / kQoiT;           // 8 bytes of the window per lane (W = 4), or 32
    typedef typename std::conditional<(kQoiLaneBytes > 16), uint64_t, uint32_t>::type Starts;   // bit i: an op starts at byte i (bits up to lane bytes + 4)
    __shared__ uint32_t win[kQoiWin / 4 + 4];                 // the window + 8 bytes of the next one (an op reads up to 4 bytes past its start)
    __shared__ uint16_t ops[kQoiWin];                         // op starts of the window, in order
    __shared__ uint4 prep[kQoiWaves > 1 ? kQoiWin : 1];       // W = 4, per op: M, V, U, run-length prefix | own run length << 12 | first byte << 18
    __shared__ unsigned long long table[64];                  // op number << 32 | pixel   (qoi_rgba_t[64] index, :453)
    __shared__ __attribute__((aligned(16))) uint32_t obuf[kQoiWaves > 1 ? 4 : kQoiOutCap];     // W = 1: pixels on their way out
    __shared__ uint32_t xs[kQoiWaves > 1 ? kQoiWin : 1], gpos[kQoiWin / 64], gdone;           // W = 4: per op its pixel; per group its first pixel's index
    __shared__ uint32_t wave_map[kQoiWaves], wave_cnt[kQoiWaves], go_on;
    // an op's function by its first byte: x = V (DIFF: the deltas; LUMA: vg - 8, vg, vg - 8 -- the second byte's nibbles are added per
    // op), y = run length | is_index << 8 | sets bytes from the stream (RGB / RGBA) << 9 | M sets the colour bytes << 10 | M sets all
    // bytes << 11 | a LUMA op's nibble mask in bits 16-19 -- one 8-byte LDS read instead of ~25 instructions of field extraction and
    // selects per op; entry 256 = no op (a lane past the window's last op).  (8 bytes, not M and U spelled out in 16: with 14.3 KB
    // of LDS per stream a CU holds 10 streams, 2560 in all, and config 5's 2730 took a second round: 44.5 -> 61 ms.)
    __shared__ uint2 lut[257];
    // W = 1: 5 * (op length - 1) by the op's first byte (0 / 5 / 15 / 20), for the boundary walk below -- 256 bytes = one row of the 64 banks:
    // a wave's 64 byte reads never conflict
    __shared__ __attribute__((aligned(256))) uint8_t sh5tab[256];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    struct __attribute__((packed, aligned(1))) AnyVec { u32x4 v; };               // 16 bytes at any address
    struct __attribute__((packed, aligned(1))) AnyU32 { uint32_t v; };
    struct __attribute__((packed, aligned(1))) AnyU64 { uint64_t v; };
    struct __attribute__((packed, aligned(1))) Any12 { uint32_t a, b, c; };

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if ((int)blockIdx.x >= n_items) return;
    const QoiItem it = items[blockIdx.x];
    const bool rgba = it.channels == 4;
    const bool rgba_out = rgba && QOI_DIRECT_RGBA;            // W = 1: RGBA pixels are stored from the registers (emit_buffered)
    uint8_t* pixels = out + it.out_off;
    const uint8_t* stream = blob + it.begin + kQoiHeader;     // chunks start here
    // bytes [14, size - 8) are chunks (p < chunks_len, :498); a chunk may read up to 4 bytes further (padding / slack)
    const int chunk_bytes = (int)it.size - kQoiPadding - kQoiHeader > 0 ? (int)it.size - kQoiPadding - kQoiHeader : 0;
    const int avail = (int)it.size - kQoiHeader + kQoiSlack;  // bytes that may be read from `stream`
    const uint32_t npx_total = it.npx;

    if (t < 64) table[t] = 0;                                 // memset(index, 0) :491
    if (t == 0) go_on = 1;
    for (uint32_t b1 = (uint32_t)t; b1 < 257u; b1 += (uint32_t)kQoiT) {
        const uint32_t top = b1 >> 6, vg = (b1 & 63u) - 32u;
        const bool is_rgb = b1 == 0xFEu, is_rgba = b1 == 0xFFu, none = b1 == 256u;
        const uint32_t v_diff = ((((b1 >> 4) & 3u) - 2u) & 255u) | ((((b1 >> 2) & 3u) - 2u) & 255u) << 8 | (((b1 & 3u) - 2u) & 255u) << 16;
        const uint32_t v_luma = ((vg - 8u) & 255u) | (vg & 255u) << 8 | ((vg - 8u) & 255u) << 16;
        uint2 e;
        e.x = none ? 0u : top == 1u ? v_diff : top == 2u ? v_luma : 0u;
#if QOI_GROUP_SCAN
        const uint32_t g_delta = (e.x >> 8) & 255u;           // the deltas as qoi_group_scan sums them: R | B << 16 in x, G in bits 20-27 of y
        e.x &= 0x00FF00FFu;
#else
        const uint32_t g_delta = 0u;
#endif
        e.y = g_delta << 20 | (none ? 0u : (top == 3u && !is_rgb && !is_rgba) ? 1u + (b1 & 63u) : 1u) | ((!none && top == 0u) ? 1u << 8 : 0u) | ((!none && top == 2u) ? 0x000F0000u : 0u) |
              ((!none && (is_rgb || is_rgba)) ? 1u << 9 : 0u) | ((!none && is_rgb) ? 1u << 10 : 0u) | ((!none && (is_rgba || top == 0u)) ? 1u << 11 : 0u);
        lut[b1] = e;
    }
    for (uint32_t b1 = (uint32_t)t; b1 < 256u; b1 += (uint32_t)kQoiT)
            sh5tab[b1] = (uint8_t)(5u * ((b1 >= 0xFEu ? b1 - 0xFAu : ((b1 >> 6) == 2u ? 2u : 1u)) - 1u));
    uint32_t carry = 0xFF000000u;                             // r = g = b = 0, a = 255 :492-495                (wave 0's state from here ...)
    uint32_t produced = 0, ops_done = 0;                      // pixels decoded, ops decoded
    uint32_t fill = 0; size_t flushed = 0;                    // pixels waiting in obuf, pixels already in the image
    uint32_t tab = 0;                                         // lane s: slot s of the table as the groups so far left it (read back right after every update:
    //                                                           the round trip runs beside the next group's parsing instead of in front of its INDEX ops)  (... to here)
    uint32_t entry = 0;                                       // offset of the first op start in the next window (every thread keeps it)

    struct Mine { uint64_t q[kQoiLaneBytes / 8]; };
    auto fetch = [&](int pos, Mine& mine, uint64_t& tail) {                       // this lane's bytes of the window at `pos` (+ 8 more for thread 0)
        const int at = pos + t * kQoiLaneBytes;
        tail = 0;
        #pragma unroll
        for (int k = 0; k < kQoiLaneBytes / 8; ++k) mine.q[k] = at + kQoiLaneBytes <= avail ? reinterpret_cast<const AnyU64*>(stream + at)[k].v : 0ull;
        if (t == 0 && pos + kQoiWin + 8 <= avail) tail = reinterpret_cast<const AnyU64*>(stream + pos + kQoiWin)->v;
    };
    auto flush_rows = [&]() {                                 // (wave 0) whole rows of 256 pixels leave; the rest moves to the front
        const uint32_t n = fill & ~255u;
        for (uint32_t i = 0; i < n; i += 256) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(&obuf[i + lane * 4]);
            const size_t px = flushed + i + (size_t)lane * 4;
            if (rgba) reinterpret_cast<AnyVec*>(pixels + px * 4)->v = v;
            else {
                const uint32_t p0 = v.x & 0xFFFFFFu, p1 = v.y & 0xFFFFFFu, p2 = v.z & 0xFFFFFFu, p3 = v.w & 0xFFFFFFu;
                Any12 w; w.a = p0 | p1 << 24; w.b = p1 >> 8 | p2 << 16; w.c = p2 >> 16 | p3 << 8;
                *reinterpret_cast<Any12*>(pixels + px * 3) = w;
            }
        }
        const uint32_t rest = fill - n;
        u32x4 keep = {0, 0, 0, 0};
        if ((uint32_t)lane * 4 < rest) keep = *reinterpret_cast<const u32x4*>(&obuf[n + lane * 4]);
        qoi_wave_sync();
        if ((uint32_t)lane * 4 < rest) *reinterpret_cast<u32x4*>(&obuf[lane * 4]) = keep;
        qoi_wave_sync();
        flushed += n; fill = rest;
    };
    auto store_px = [&](size_t px, uint32_t v) {
        if (rgba) reinterpret_cast<AnyU32*>(pixels + px * 4)->v = v;
        else { uint8_t* o = pixels + px * 3; o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16); }
    };

    Mine dmine; uint64_t dtail;
    fetch(0, dmine, dtail);
    __syncthreads();
#if QOI_PROFILE
    unsigned long long qprof[8] = {}, qprof_t0 = clock64();
#endif
    for (int pos = 0; pos < chunk_bytes && go_on; pos += kQoiWin) {
        // ---- the window into LDS; the next one into registers
        #pragma unroll
        for (int k = 0; k < kQoiLaneBytes / 8; ++k) { win[t * (kQoiLaneBytes / 4) + 2 * k] = (uint32_t)dmine.q[k]; win[t * (kQoiLaneBytes / 4) + 2 * k + 1] = (uint32_t)(dmine.q[k] >> 32); }
        if (t == 0) { win[kQoiWin / 4] = (uint32_t)dtail; win[kQoiWin / 4 + 1] = (uint32_t)(dtail >> 32); }
        // ---- A. op starts among this lane's 8 bytes, for entry offsets 0..4 (bit i of s[e]: an op starts at byte i)
        Starts s[5] = { 1, 2, 4, 8, 16 };
        uint32_t map = 0;                                     // entry offset e -> offset of the first op start in the next lane's bytes
        constexpr int kRec = (kQoiLaneBytes + 5) / 6;         // six byte positions per record register
        uint32_t rec[kRec] = {};
        if constexpr (true) {
            // One wave per stream, 32 bytes per lane: five 64-bit bit sets walked side by side cost 30 vector instructions per byte (a
            // variable 64-bit shift and two halves to OR per set) -- 800 per window and lane, a fifth of everything the kernel does (round
            // 6, from the ISA: a batch's time is its instruction count).  The same walk turned round: ONE register holds, for the next
            // five byte positions, WHICH of the five entry offsets have an op starting there (5 slots x 5 bits).  Per byte: the set for
            // this position (`cur`) is recorded (position i -> bits 5 (i mod 6) .. + 4 of rec[i / 6]), the window slides by one slot and
            // `cur` is added to the slot of position i + length -- five instructions and one byte read from the 256-byte table, whatever
            // the number of entry offsets.  What is pending after the last byte is the exit map; the true entry's starts are picked out
            // of the record once the scan over the lanes has said which entry is true (below).
            uint32_t w = 0x01041041u;                         // slot e = { e }: with entry offset e the first op starts at byte e
            #pragma unroll
            for (int i = 0; i < kQoiLaneBytes; ++i) {
                const uint32_t b = (uint32_t)(dmine.q[i >> 3] >> (8 * (i & 7))) & 255u;
                const uint32_t sh = sh5tab[b];
                const uint32_t cur = w & 31u;
                rec[i / 6] |= cur << (5 * (i % 6));
                // (one v_lshl_or_b32; left to itself the compiler computes the next `cur` from the two halves in parallel -- a shorter chain, a
                //  fourth instruction per byte)
                { const uint32_t ws = w >> 5; asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(w) : "v"(cur), "v"(sh), "v"(ws)); }
            }
            #pragma unroll
            for (int e = 0; e < 5; ++e) {                     // entry e's chain is pending in exactly one slot k (its first start at or behind the lane's last byte): bit 5 k + e
                const uint32_t at = (uint32_t)__builtin_ctz(w & (0x00108421u << e)) - (uint32_t)e;      // 5 k
                map |= ((at * 13u) >> 6) << (3 * e);          // k = at / 5 for at = 0, 5, ... 20
            }
        } else {
        #pragma unroll
        for (int i = 0; i < kQoiLaneBytes; ++i) {
            const uint32_t b = (uint32_t)(dmine.q[i >> 3] >> (8 * (i & 7))) & 255u;
            const uint32_t len = b >= 0xFEu ? b - 0xFAu : ((b >> 6) == 2u ? 2u : 1u);          // RGB 4, RGBA 5, LUMA 2, the others 1
            #pragma unroll
            for (int e = 0; e < 5; ++e) s[e] |= (s[e] & ((Starts)1 << i)) << len;
        }
        #pragma unroll
        for (int e = 0; e < 5; ++e) map |= (uint32_t)__builtin_ctz((uint32_t)(s[e] >> kQoiLaneBytes)) << (3 * e);
        }
        fetch(pos + kQoiWin, dmine, dtail);
        qoi_map_scan64(map);                                  // lanes 0 .. this one, composed
        if (lane == 63) wave_map[wave] = map;
        QPROF(0);
        __syncthreads();
        QPROF(1);
        uint32_t wave_entry = entry;
        #pragma unroll
        for (int w = 0; w < kQoiWaves; ++w) {
            const uint32_t after = qoi_map_apply(wave_map[w], entry);
            if (w < wave) wave_entry = after;
            entry = after;                                    // (after the loop: the next window's entry offset)
        }
        const uint32_t my_entry = qoi_map_apply(qoi_dpp<0x138, 0xF>(map, kQoiMapId), wave_entry);
        uint32_t starts;
        if constexpr (true) {
            // the true entry's bits of the record (5 k + e in every register, k = 0..5), squeezed to one bit per byte: three bits at a time by a
            // multiplication whose cross terms fall outside the field (x = b0 | b5 | b10: x * 0x111 has b0, b5, b10 at bits 8, 9, 10)
            starts = 0;
            #pragma unroll
            for (int j = 0; j < kRec; ++j) {
                const uint32_t x = rec[j] >> my_entry;
                const uint32_t lo3 = (((x & 0x421u) * 0x111u) >> 8) & 7u, hi3 = ((((x >> 15) & 0x421u) * 0x111u) >> 8) & 7u;
                starts |= (lo3 | hi3 << 3) << (6 * j);        // (the last register holds 2 positions -- the rest of it was never written)
            }
        } else
        starts = (uint32_t)(my_entry == 0 ? s[0] : my_entry == 1 ? s[1] : my_entry == 2 ? s[2] : my_entry == 3 ? s[3] : s[4]);
        {
            const int room = chunk_bytes - (pos + t * kQoiLaneBytes);                // op starts only in this lane's bytes and below chunks_len
            const int n = room < kQoiLaneBytes ? room : kQoiLaneBytes;
            starts &= n >= 32 ? ~0u : n <= 0 ? 0u : (1u << n) - 1u;
        }
        // ---- compact the op starts into ops[]
        const uint32_t mine_n = (uint32_t)__builtin_popcount(starts);
        uint32_t incl = mine_n;
        qoi_add_scan_step<0x111, 0xF>(incl); qoi_add_scan_step<0x112, 0xF>(incl); qoi_add_scan_step<0x114, 0xF>(incl); qoi_add_scan_step<0x118, 0xF>(incl);
        qoi_add_scan_step<0x142, 0xA>(incl); qoi_add_scan_step<0x143, 0xC>(incl);
        if (lane == 63) wave_cnt[wave] = incl;
        __syncthreads();
        uint32_t at = incl - mine_n, nops = 0;
        #pragma unroll
        for (int w = 0; w < kQoiWaves; ++w) { const uint32_t c = wave_cnt[w]; if (w < wave) at += c; nops += c; }
        for (uint32_t m = starts; m; m &= m - 1) ops[at++] = (uint16_t)(t * kQoiLaneBytes + __builtin_ctz(m));
        QPROF(2);
        __syncthreads();
        QPROF(1);
        // ---- B. the ops as functions of the previous pixel, 64 at a time (W = 4: groups dealt to the waves, results in LDS)
        struct Group { QoiFn f; uint32_t run_incl, npx, b1; bool is_index; };
        // The three look-ups a group starts with depend on each other (op offset -> its bytes in the window -> the table entry of its first
        // byte): three LDS round trips before the first useful instruction; rocprofv3 has the waves of the one-wave kernel waiting on a counter
        // a third of their time (SQ_WAIT_ANY 14.5 G of SQ_WAVE_CYCLES 42 G, profiles/r06_qoi_group_scan_pmc.txt).  They are stages so that the
        // one-wave loop CAN fetch each a group ahead of its use (QOI_PREFETCH) -- which turned out slower: with three waves on the SIMDs that
        // decide the launch's time, one wave's wait is another's issue slot, and the staging costs eight instructions per group.
        struct Raw { uint32_t lo, hi; uint2 e; };
        auto fetch_op = [&](uint32_t g) -> uint32_t { return g + lane < nops ? ops[g + lane] : 0u; };         // (lanes past the last op: offset 0, entry 256)
        struct Win3 { uint32_t w0, w1, w2, sh; };
        auto fetch_win = [&](uint32_t o) -> Win3 { return Win3{ win[o >> 2], win[(o >> 2) + 1], win[(o >> 2) + 2], 8 * (o & 3) }; };
        auto fetch_lut = [&](uint32_t g, const Win3& w) -> Raw {
            Raw r;
            r.lo = __builtin_amdgcn_alignbit(w.w1, w.w0, w.sh); r.hi = __builtin_amdgcn_alignbit(w.w2, w.w1, w.sh);
            r.e = lut[g + lane < nops ? (r.lo & 255u) : 256u];
            return r;
        };
        auto scan_group = [&](const Raw& r) -> Group {
            const uint32_t lo = r.lo, hi = r.hi;
            const uint32_t b1 = lo & 255u, b2 = (lo >> 8) & 255u;
            const uint2 e = r.e;
            const uint32_t v_abs = __builtin_amdgcn_alignbit(hi, lo, 8);                          // stream bytes 1..4
            const uint32_t nib = ((b2 >> 4) | (b2 & 15u) << 16) & (e.y >> 16 | (e.y & 0x000F0000u));  // LUMA: dr - dg + 8, db - dg + 8 (:528-530); else 0
            Group G;
            G.is_index = (e.y >> 8 & 1u) != 0;
            G.b1 = b1;
            G.npx = e.y & 0xFFu;
            G.run_incl = G.npx;
#if QOI_GROUP_SCAN
            // (the nibbles go on top of the halves of x: at most 255 + 15, the halves do not meet)
            G.f = qoi_group_scan(lane, e.x + nib, (e.y >> 20) & 255u, (uint32_t)((int32_t)(e.y << 22) >> 31), G.is_index, (e.y & 0xA00u) == 0xA00u, v_abs, G.run_incl);
#else
            G.f.M = (uint32_t)((int32_t)(e.y << 20) >> 31) | ((uint32_t)((int32_t)(e.y << 21) >> 31) & 0x00FFFFFFu);      // bit 11: all, bit 10: colour
            G.f.U = (uint32_t)((int32_t)(e.y << 23) >> 31);                                                             // bit 8: an INDEX op
            G.f.V = (e.y >> 9 & 1u) ? (v_abs & G.f.M) : qoi_add_bytes(e.x, nib);                  // RGB / RGBA: the bytes the op sets
            qoi_scan_step<0x111, 0xF>(G.f, G.run_incl); qoi_scan_step<0x112, 0xF>(G.f, G.run_incl); qoi_scan_step<0x114, 0xF>(G.f, G.run_incl);
            qoi_scan_step<0x118, 0xF>(G.f, G.run_incl); qoi_scan_step<0x142, 0xA>(G.f, G.run_incl); qoi_scan_step<0x143, 0xC>(G.f, G.run_incl);
#endif
            return G;
        };
        auto parse_group = [&](uint32_t g) -> Group { return scan_group(fetch_lut(g, fetch_win(fetch_op(g)))); };
        // ---- C. pixels and table of one group (in stream order: wave 0) -> the group's pixels, its first pixel's index
        auto resolve_group = [&](uint32_t g, const Group& G, uint32_t& first_px) -> uint32_t {
            const uint32_t cnt = nops - g < 64u ? nops - g : 64u;
            const bool active = (uint32_t)lane < cnt;
            const QoiFn f = G.f;
            // pixels: lanes before the first INDEX op at once, then INDEX op by INDEX op
            uint32_t x = qoi_add_bytes(carry & ~f.M, f.V);
            uint64_t todo = __ballot(G.is_index);
            uint32_t h = qoi_hash(x);
            while (todo) {
                const int j = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int jn = todo ? __builtin_ctzll(todo) : 64;
                const uint32_t slot = qoi_readlane(G.b1, j) & 63u;
                const uint64_t m = __ballot(h == slot) & ((1ull << j) - 1ull);
                const uint32_t base = m ? qoi_readlane(x, 63 - __builtin_clzll(m)) : qoi_readlane(tab, (int)slot);
                const uint32_t nx = qoi_add_bytes(base & f.U, f.V);       // from an INDEX op on, every byte is set (M = all)
                const bool in = lane >= j && lane < jn;
                x = in ? nx : x;
                h = in ? qoi_hash(nx) : h;
            }
            if (active) atomicMax(&table[h], (unsigned long long)(ops_done + 1u + (uint32_t)lane) << 32 | x);
            tab = (uint32_t)table[lane];                                  // (LDS operations of a wave execute in order: this sees the update)
            carry = qoi_readlane(x, (int)cnt - 1);
            first_px = produced;
            const uint32_t total = qoi_readlane(G.run_incl, (int)cnt - 1), room = npx_total - produced;
            produced += total < room ? total : room;
            ops_done += cnt;
            return x;
        };
        // ---- D (one wave per stream): the group's pixels through the LDS buffer, out in 1 KiB rows
        auto emit_buffered = [&](const Group& G, uint32_t x, uint32_t first_px) {
            const uint32_t off = G.run_incl - G.npx, room = npx_total - first_px;   // (room >= 1: the loop stops once the image is full)
            const uint32_t npx = off >= room ? 0u : (G.npx < room - off ? G.npx : room - off);
            const uint32_t total = produced - first_px;                            // wave-uniform
            // RGBA outputs leave from the registers: a pixel is a dword, 64 consecutive ops are 64 consecutive dwords (runs aside), and the
            // lines they share with the groups before and after meet in L2 -- no LDS write, no read back, no rows to shuffle: a fifth fewer
            // instructions per group, which is what a batch's time is made of.  (RGB outputs are 3 bytes per pixel: they keep the buffer,
            // which packs them into whole dwords.)  Long runs of either kind go straight to the image too.
            if (rgba_out) {
                if (npx) store_px((size_t)first_px + off, x);
                if (__any(npx > 1u)) for (uint32_t r = 1; __any(r < npx); ++r) if (r < npx) store_px((size_t)first_px + off + r, x);
                flushed = produced;
                return;
            }
            if (total > (uint32_t)kQoiBufGroup) {             // long runs: what waits in the buffer leaves, then the group's pixels go straight to the image
                for (uint32_t i = lane; i < fill; i += 64) store_px(flushed + i, obuf[i]);
                for (uint32_t r = 0; __any(r < npx); ++r) if (r < npx) store_px((size_t)first_px + off + r, x);
                flushed = produced; fill = 0;
                qoi_wave_sync();
                return;
            }
            if (npx) obuf[fill + off] = x;
            if (__any(npx > 1u)) {
                for (uint32_t r = 1; __any(r < npx); ++r) if (r < npx) obuf[fill + off + r] = x;
            }
            fill += total;
            qoi_wave_sync();
            if (fill >= 256u) flush_rows();
        };
        if constexpr (kQoiWaves == 1) {
#if QOI_PREFETCH
            Raw cur = fetch_lut(0, fetch_win(fetch_op(0)));
            uint32_t o_next = fetch_op(64);
            for (uint32_t g = 0; g < nops && produced < npx_total; g += 64) {
                const Win3 w_next = fetch_win(o_next);                 // (its offsets were asked for a group ago)
                const Group G = scan_group(cur);
                QPROF(3);
                cur = fetch_lut(g + 64, w_next);
                uint32_t first_px;
                const uint32_t x = resolve_group(g, G, first_px);
                QPROF(5);
                o_next = fetch_op(g + 128);
                emit_buffered(G, x, first_px);
                QPROF(6);
            }
#else
            for (uint32_t g = 0; g < nops && produced < npx_total; g += 64) {
                const Group G = parse_group(g);
                QPROF(3);
                uint32_t first_px;
                const uint32_t x = resolve_group(g, G, first_px);
                QPROF(5);
                emit_buffered(G, x, first_px);
                QPROF(6);
            }
#endif
        } else {
            // (a full group's run lengths can sum to 64 * 62 = 3968 < 4096: the 12-bit prefix never wraps)
            for (uint32_t g = (uint32_t)wave * 64u; g < nops; g += kQoiT) {
                const Group G = parse_group(g);
                prep[g + lane] = make_uint4(G.f.M, G.f.V, G.f.U, G.run_incl | G.npx << 12 | G.b1 << 18 | (G.is_index ? 1u << 26 : 0u));
            }
            QPROF(3);
            __syncthreads();
            QPROF(1);
            if (wave == 0) {                                  // the serial part: nothing but the pixels and the table
                uint4 pr = prep[lane];
                uint32_t g = 0;
                for (; g < nops && produced < npx_total; g += 64) {
                    Group G;
                    G.f = QoiFn{ pr.x, pr.y, pr.z };
                    G.run_incl = pr.w & 0xFFFu; G.npx = (pr.w >> 12) & 63u; G.b1 = (pr.w >> 18) & 255u; G.is_index = ((pr.w >> 26) & 1u) != 0;
                    if (g + 64 < nops) pr = prep[g + 64 + lane];                   // (the next group's, while this one is worked on)
                    uint32_t first_px;
                    xs[g + lane] = resolve_group(g, G, first_px);
                    if (lane == 0) gpos[g >> 6] = first_px;
                }
                if (lane == 0) gdone = g;                     // groups from here on were not resolved: the image was full before them
            }
            QPROF(4);
            __syncthreads();
            QPROF(1);
            // D (four waves per stream): every wave writes the pixels of its groups straight to the image
            const uint32_t resolved = gdone;
            for (uint32_t g = (uint32_t)wave * 64u; g < resolved; g += kQoiT) {
                const uint32_t first_px = gpos[g >> 6];
                if (g + lane >= nops) continue;
                const uint4 pr = prep[g + lane];
                const uint32_t run_incl = pr.w & 0xFFFu, npx = (pr.w >> 12) & 63u, x = xs[g + lane];
                const size_t at = (size_t)first_px + (run_incl - npx);
                for (uint32_t r = 0; r < npx && at + r < npx_total; ++r) store_px(at + r, x);
            }
        }
        if (wave == 0 && lane == 0 && produced >= npx_total) go_on = 0;
        QPROF(4);
        __syncthreads();
        QPROF(1);
    }
#if QOI_PROFILE
    if (t == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_qoi_prof[k], qprof[k]);
#endif
    // what is left in the buffer, then the tail of a stream that ended early: the last pixel repeats (:496-497, run / p >= chunks_len)
    if (wave == 0) {
        for (uint32_t i = lane; i < fill; i += 64) store_px(flushed + i, obuf[i]);
        for (size_t i = (size_t)produced + lane; i < npx_total; i += 64) store_px(i, carry);
    }
}

// ---- the same decode as a PIPELINE: five waves per stream (batches below kQoiWideBelow streams) --------------------------------
// In k_qoi_decode<4> the four waves prepare a window together (A, compaction, B), then three of them wait while wave 0 walks its
// groups (C), then all write its pixels (D): the serial walk and the parallel phases take turns.  Here wave 0 does nothing but C,
// and waves 1-4 prepare window w + 1 and write the pixels of window w - 1 meanwhile (two sets of buffers): a stream takes
// max(C, A + B + D) instead of their sum.  The hardware barrier counts every wave of the workgroup, so the four producers meet at a
// counter in LDS instead, and windows change hands through two monotonic flags per buffer set (ready: prepared up to window w;
// done: walked up to window w).  Every wait is bounded: a flag that never comes (a bug) ends the wait, never hangs the GPU.
// A prepared op is 8 bytes here (M and U only take three values each: nothing / colour bytes / all, and nothing / alpha / all).
constexpr int kPipeT = 256;                                   // producer threads
struct QoiPipeBuf {
    uint32_t win[kQoiWin / 4 + 4];
    uint16_t ops[kQoiWin];
    uint2    prep[kQoiWin];                                    // V | run-length prefix, own run length << 12, first byte << 18, is_index << 26, M code << 27, U code << 29
    uint32_t xs[kQoiWin];
    uint32_t gpos[kQoiWin / 64];
    uint32_t nops, gdone;
};
__device__ __forceinline__ void qoi_flag_set(uint32_t* f, uint32_t v, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void qoi_flag_wait(uint32_t* f, uint32_t v)
{
    for (uint32_t spins = 0; __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v && spins < 2000000u; ++spins) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void qoi_producers_meet(uint32_t* counter, uint32_t& target, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    target += 4u;
    qoi_flag_wait(counter, target);
}

__global__ __launch_bounds__(320) void k_qoi_pipe(const QoiItem* items, int n_items, const uint8_t* blob, uint8_t* out)
{
    constexpr int kLaneBytes = kQoiWin / kPipeT;              // 8
    __shared__ QoiPipeBuf pb[2];
    __shared__ unsigned long long table[64];
    __shared__ uint32_t wave_map[4], wave_cnt[4], f_ready[2], f_done[2], f_meet;
    struct __attribute__((packed, aligned(1))) AnyU32 { uint32_t v; };
    struct __attribute__((packed, aligned(1))) AnyU64 { uint64_t v; };

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if ((int)blockIdx.x >= n_items) return;
    const QoiItem it = items[blockIdx.x];
    const bool rgba = it.channels == 4;
    uint8_t* pixels = out + it.out_off;
    const uint8_t* stream = blob + it.begin + kQoiHeader;
    const int chunk_bytes = (int)it.size - kQoiPadding - kQoiHeader > 0 ? (int)it.size - kQoiPadding - kQoiHeader : 0;
    const int avail = (int)it.size - kQoiHeader + kQoiSlack;
    const uint32_t npx_total = it.npx;
    if (t < 64) table[t] = 0;
    if (t == 0) { f_ready[0] = f_ready[1] = f_done[0] = f_done[1] = f_meet = 0; }
    __syncthreads();
    auto store_px = [&](size_t px, uint32_t v) {
        if (rgba) reinterpret_cast<AnyU32*>(pixels + px * 4)->v = v;
        else { uint8_t* o = pixels + px * 3; o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16); }
    };

    if (wave == 0) {
        // ---- C: the walk (see k_qoi_decode::resolve_group) ------------------------------------------------------------------------
        uint32_t carry = 0xFF000000u, produced = 0, ops_done = 0;
        uint32_t tab = 0;                                     // lane s: slot s of the table, read back after every update
        uint32_t w = 0;
        for (int pos = 0; pos < chunk_bytes; pos += kQoiWin, ++w) {
            QoiPipeBuf& B = pb[w & 1u];
            qoi_flag_wait(&f_ready[w & 1u], w + 1u);
            const uint32_t nops = B.nops;
            uint32_t g = 0;
            if (produced < npx_total && nops) {
                uint2 pr = B.prep[lane];
                for (; g < nops && produced < npx_total; g += 64) {
                    const uint32_t mc = (pr.y >> 27) & 3u, uc = (pr.y >> 29) & 3u;
                    const QoiFn f = { mc == 0u ? 0u : mc == 1u ? 0x00FFFFFFu : 0xFFFFFFFFu, pr.x, uc == 0u ? 0u : uc == 1u ? 0xFF000000u : 0xFFFFFFFFu };
                    const uint32_t run_incl = pr.y & 0xFFFu, b1 = (pr.y >> 18) & 255u;
                    const bool is_index = ((pr.y >> 26) & 1u) != 0;
                    if (g + 64 < nops) pr = B.prep[g + 64 + lane];
                    const uint32_t cnt = nops - g < 64u ? nops - g : 64u;
                    const bool active = (uint32_t)lane < cnt;
                    uint32_t x = qoi_add_bytes(carry & ~f.M, f.V);
                    uint64_t todo = __ballot(is_index && active);
                    uint32_t h = qoi_hash(x);
                    while (todo) {
                        const int j = __builtin_ctzll(todo);
                        todo &= todo - 1;
                        const int jn = todo ? __builtin_ctzll(todo) : 64;
                        const uint32_t slot = qoi_readlane(b1, j) & 63u;
                        const uint64_t m = __ballot(h == slot) & ((1ull << j) - 1ull);
                        const uint32_t base = m ? qoi_readlane(x, 63 - __builtin_clzll(m)) : qoi_readlane(tab, (int)slot);
                        const uint32_t nx = qoi_add_bytes(base & f.U, f.V);
                        const bool in = lane >= j && lane < jn;
                        x = in ? nx : x;
                        h = in ? qoi_hash(nx) : h;
                    }
                    if (active) atomicMax(&table[h], (unsigned long long)(ops_done + 1u + (uint32_t)lane) << 32 | x);
                    tab = (uint32_t)table[lane];
                    carry = qoi_readlane(x, (int)cnt - 1);
                    const uint32_t first_px = produced;
                    const uint32_t total = qoi_readlane(run_incl, (int)cnt - 1), room = npx_total - produced;
                    produced += total < room ? total : room;
                    ops_done += cnt;
                    B.xs[g + lane] = x;
                    if (lane == 0) B.gpos[g >> 6] = first_px;
                }
            }
            if (lane == 0) B.gdone = g;
            qoi_flag_set(&f_done[w & 1u], w + 1u, lane);
        }
        for (size_t i = (size_t)produced + lane; i < npx_total; i += 64) store_px(i, carry);       // a stream that ended early: the last pixel repeats (:496-497)
        return;
    }

    // ---- producers: A, compaction, B for window w; D for window w - 1 ------------------------------------------------------------
    const int tp = t - 64, pwave = wave - 1;
    uint32_t meet = 0, entry = 0;
    struct Mine { uint64_t q[kLaneBytes / 8]; };
    auto fetch = [&](int pos, Mine& mine, uint64_t& tail) {
        const int at = pos + tp * kLaneBytes;
        tail = 0;
        #pragma unroll
        for (int k = 0; k < kLaneBytes / 8; ++k) mine.q[k] = at + kLaneBytes <= avail ? reinterpret_cast<const AnyU64*>(stream + at)[k].v : 0ull;
        if (tp == 0 && pos + kQoiWin + 8 <= avail) tail = reinterpret_cast<const AnyU64*>(stream + pos + kQoiWin)->v;
    };
    auto write_pixels = [&](uint32_t v) {                     // D of window v
        QoiPipeBuf& B = pb[v & 1u];
        qoi_flag_wait(&f_done[v & 1u], v + 1u);
        const uint32_t resolved = B.gdone, nops = B.nops;
        for (uint32_t g = (uint32_t)pwave * 64u; g < resolved; g += kPipeT) {
            if (g + lane >= nops) continue;
            const uint32_t first_px = B.gpos[g >> 6];
            const uint2 pr = B.prep[g + lane];
            const uint32_t run_incl = pr.y & 0xFFFu, npx = (pr.y >> 12) & 63u, x = B.xs[g + lane];
            const size_t at = (size_t)first_px + (run_incl - npx);
            for (uint32_t r = 0; r < npx && at + r < npx_total; ++r) store_px(at + r, x);
        }
    };
    Mine dmine; uint64_t dtail;
    fetch(0, dmine, dtail);
    uint32_t w = 0;
    for (int pos = 0; pos < chunk_bytes; pos += kQoiWin, ++w) {
        QoiPipeBuf& B = pb[w & 1u];
        uint32_t* win = B.win; uint16_t* ops = B.ops;
        #pragma unroll
        for (int k = 0; k < kLaneBytes / 8; ++k) { win[tp * (kLaneBytes / 4) + 2 * k] = (uint32_t)dmine.q[k]; win[tp * (kLaneBytes / 4) + 2 * k + 1] = (uint32_t)(dmine.q[k] >> 32); }
        if (tp == 0) { win[kQoiWin / 4] = (uint32_t)dtail; win[kQoiWin / 4 + 1] = (uint32_t)(dtail >> 32); }
        // A. op starts among this lane's 8 bytes, for entry offsets 0..4
        uint32_t s[5] = { 1, 2, 4, 8, 16 };
        #pragma unroll
        for (int i = 0; i < kLaneBytes; ++i) {
            const uint32_t b = (uint32_t)(dmine.q[i >> 3] >> (8 * (i & 7))) & 255u;
            const uint32_t len = b >= 0xFEu ? b - 0xFAu : ((b >> 6) == 2u ? 2u : 1u);
            #pragma unroll
            for (int e = 0; e < 5; ++e) s[e] |= (s[e] & (1u << i)) << len;
        }
        uint32_t map = 0;
        #pragma unroll
        for (int e = 0; e < 5; ++e) map |= (uint32_t)__builtin_ctz(s[e] >> kLaneBytes) << (3 * e);
        fetch(pos + kQoiWin, dmine, dtail);
        qoi_map_scan64(map);
        if (lane == 63) wave_map[pwave] = map;
        qoi_producers_meet(&f_meet, meet, lane);
        uint32_t wave_entry = entry;
        #pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t after = qoi_map_apply(wave_map[k], entry);
            if (k < pwave) wave_entry = after;
            entry = after;
        }
        const uint32_t my_entry = qoi_map_apply(qoi_dpp<0x138, 0xF>(map, kQoiMapId), wave_entry);
        uint32_t starts = my_entry == 0 ? s[0] : my_entry == 1 ? s[1] : my_entry == 2 ? s[2] : my_entry == 3 ? s[3] : s[4];
        {
            const int room = chunk_bytes - (pos + tp * kLaneBytes);
            const int n = room < kLaneBytes ? room : kLaneBytes;
            starts &= n >= 32 ? ~0u : n <= 0 ? 0u : (1u << n) - 1u;
        }
        const uint32_t mine_n = (uint32_t)__builtin_popcount(starts);
        uint32_t incl = mine_n;
        qoi_add_scan_step<0x111, 0xF>(incl); qoi_add_scan_step<0x112, 0xF>(incl); qoi_add_scan_step<0x114, 0xF>(incl); qoi_add_scan_step<0x118, 0xF>(incl);
        qoi_add_scan_step<0x142, 0xA>(incl); qoi_add_scan_step<0x143, 0xC>(incl);
        if (lane == 63) wave_cnt[pwave] = incl;
        qoi_producers_meet(&f_meet, meet, lane);
        uint32_t at = incl - mine_n, nops = 0;
        #pragma unroll
        for (int k = 0; k < 4; ++k) { const uint32_t c = wave_cnt[k]; if (k < pwave) at += c; nops += c; }
        for (uint32_t m = starts; m; m &= m - 1) ops[at++] = (uint16_t)(tp * kLaneBytes + __builtin_ctz(m));
        qoi_producers_meet(&f_meet, meet, lane);
        // B. the ops as functions of the previous pixel, 64 at a time, the groups dealt to the producer waves
        for (uint32_t g = (uint32_t)pwave * 64u; g < nops; g += kPipeT) {
            const bool active = g + lane < nops;
            uint32_t lo = 0, hi = 0;
            if (active) {
                const uint32_t o = ops[g + lane];
                const uint32_t w0 = win[o >> 2], w1 = win[(o >> 2) + 1], w2 = win[(o >> 2) + 2];
                const uint32_t sh = 8 * (o & 3);
                lo = __builtin_amdgcn_alignbit(w1, w0, sh); hi = __builtin_amdgcn_alignbit(w2, w1, sh);
            }
            const uint32_t b1 = lo & 255u, top = b1 >> 6, b2 = (lo >> 8) & 255u;
            const bool is_rgb = b1 == 0xFEu, is_rgba = b1 == 0xFFu;
            const bool is_run = top == 3u && !is_rgb && !is_rgba;
            const uint32_t vg = (b1 & 63u) - 32u;
            const uint32_t v_diff = ((((b1 >> 4) & 3u) - 2u) & 255u) | ((((b1 >> 2) & 3u) - 2u) & 255u) << 8 | (((b1 & 3u) - 2u) & 255u) << 16;
            const uint32_t v_luma = ((vg - 8u + (b2 >> 4)) & 255u) | (vg & 255u) << 8 | ((vg - 8u + (b2 & 15u)) & 255u) << 16;
            const uint32_t v_abs = __builtin_amdgcn_alignbit(hi, lo, 8);
            const bool is_index = active && top == 0u;
            QoiFn f;
            f.M = !active ? 0u : is_rgb ? 0x00FFFFFFu : (is_rgba || top == 0u) ? 0xFFFFFFFFu : 0u;
            f.U = is_index ? 0xFFFFFFFFu : 0u;
            f.V = !active ? 0u : is_rgb ? (v_abs & 0x00FFFFFFu) : is_rgba ? v_abs : top == 1u ? v_diff : top == 2u ? v_luma : 0u;
            const uint32_t npx = !active ? 0u : is_run ? 1u + (b1 & 63u) : 1u;
            uint32_t run_incl = npx;
#if QOI_GROUP_SCAN
            {
                const uint32_t dv = !active ? 0u : top == 1u ? v_diff : top == 2u ? v_luma : 0u;
                f = qoi_group_scan(lane, dv & 0x00FF00FFu, (dv >> 8) & 255u, (active && (is_rgb || is_rgba)) ? 0xFFFFFFFFu : 0u, is_index, active && is_rgba, v_abs, run_incl);
            }
#else
            qoi_scan_step<0x111, 0xF>(f, run_incl); qoi_scan_step<0x112, 0xF>(f, run_incl); qoi_scan_step<0x114, 0xF>(f, run_incl);
            qoi_scan_step<0x118, 0xF>(f, run_incl); qoi_scan_step<0x142, 0xA>(f, run_incl); qoi_scan_step<0x143, 0xC>(f, run_incl);
#endif
            const uint32_t mc = f.M == 0u ? 0u : f.M == 0x00FFFFFFu ? 1u : 2u, uc = f.U == 0u ? 0u : f.U == 0xFF000000u ? 1u : 2u;
            B.prep[g + lane] = make_uint2(f.V, run_incl | npx << 12 | b1 << 18 | (is_index ? 1u << 26 : 0u) | mc << 27 | uc << 29);
        }
        if (tp == 0) B.nops = nops;
        qoi_producers_meet(&f_meet, meet, lane);               // every producer's share of the window is in place
        if (tp == 0) qoi_flag_set(&f_ready[w & 1u], w + 1u, 0);
        if (w > 0) write_pixels(w - 1u);
        qoi_producers_meet(&f_meet, meet, lane);               // ... and every share of window w - 1's pixels is written: its buffers may be reused
    }
    if (w > 0) write_pixels(w - 1u);
}

// The pipelined kernel keeps a compute unit busy by itself (its four producer waves compute nearly all the time: two workgroups on
// one CU take twice as long), so it only pays while every stream has a CU of its own: 64 / 256 streams 17.3 -> 11.6 ms, but 341
// streams 20.2 -> 23.3 ms, 700 streams 22.8 -> 34.8 ms.  GAMUT_HIP_QOI_PIPE=0 / 1 forces the choice (measurements, tests).
inline bool qoi_pipeline(int n)
{
    const char* e = getenv("GAMUT_HIP_QOI_PIPE");
    if (e && *e) return *e != '0';
    static const int cus = [] { int dev = 0, c = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) c = 0; return c; }();
    return n <= cus;
}

inline uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

// header checks of qoi_decode :458-480; 0 = ok
int read_header(const uint8_t* data, int size, gamut_hip_qoi_desc* d, int channels)
{
    memset(d, 0, sizeof(*d));
    if ((channels != 0 && channels != 3 && channels != 4) || !data || size < kQoiHeader + kQoiPadding)
        return set_error(GAMUT_HIP_ERR_DECODE, "qoi: invalid arguments or truncated file");
    const uint32_t magic = be32(data);
    d->width = be32(data + 4); d->height = be32(data + 8); d->channels = data[12]; d->colorspace = data[13];
    if (d->width == 0 || d->height == 0 || d->channels < 3 || d->channels > 4 || d->colorspace > 1 || magic != kQoiMagic ||
        d->height >= kQoiPixelsMax / d->width)
        return set_error(GAMUT_HIP_ERR_DECODE, "qoi: bad header");
    return GAMUT_HIP_OK;
}


int decode_batch(const uint8_t* const* data, const int* size, int count, int channels, const int64_t* out_offset, uint8_t* d_out,
                 gamut_hip_qoi_desc* descs, int* status_host, hipStream_t stream)
{
    std::vector<QoiItem> items; std::vector<int> src; size_t blob_size = 0;
    int first = GAMUT_HIP_OK, first_idx = -1;
    for (int i = 0; i < count; ++i) {
        const int rc = read_header(data[i], size[i], &descs[i], channels);
        if (status_host) status_host[i] = rc;
        if (rc != GAMUT_HIP_OK) { if (first == GAMUT_HIP_OK) { first = rc; first_idx = i; } continue; }
        QoiItem it{}; it.begin = blob_size; it.out_off = out_offset[i]; it.size = (uint32_t)size[i];
        it.npx = descs[i].width * descs[i].height; it.channels = channels ? channels : descs[i].channels;
        items.push_back(it); src.push_back(i);
        blob_size += ((size_t)size[i] + kQoiSlack + 15) & ~(size_t)15;   // the lane's reader fetches whole 64-byte blocks: slack after every stream
    }
    if (!items.empty()) {
        const size_t o_blob = (items.size() * sizeof(QoiItem) + 255) & ~(size_t)255, total = o_blob + blob_size;
        static thread_local PerDevice<DeviceScratch> staging_pd;
        static thread_local PerDevice<PinnedScratch> pinned_pd;
        DeviceScratch& staging = staging_pd.cur(); PinnedScratch& pinned = pinned_pd.cur();
        uint8_t* d = (uint8_t*)staging.get(total, stream);
        uint8_t* h = pinned.get(total, stream);
        if (!d || !h) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "qoi: staging allocation of %zu bytes failed", total);
        // the files are gathered into one pinned image on a few host threads and go up in one DMA
        int workers = host_threads();
        workers = workers < 1 ? 1 : workers > 16 ? 16 : workers;
        if ((size_t)workers > blob_size / (4u << 20) + 1) workers = (int)(blob_size / (4u << 20) + 1);
        // ... file by file: a file's bytes are on their way (on `stream`, ahead of the kernel) while the next ones are still copied
        int dev = 0;
        GAMUT_HIP_CHECK(hipGetDevice(&dev));
        std::atomic<int> upload_failed{ 0 };
        memcpy(h, items.data(), items.size() * sizeof(QoiItem));
        GAMUT_HIP_CHECK(hipMemcpyAsync(d, h, items.size() * sizeof(QoiItem), hipMemcpyHostToDevice, stream));
        const int n = (int)items.size();
        auto launch = [&](int g0, int cnt, hipStream_t s) {
            const QoiItem* its = reinterpret_cast<const QoiItem*>(d) + g0;
            if (cnt < kQoiWideBelow && qoi_pipeline(cnt)) hipLaunchKernelGGL(k_qoi_pipe, dim3(cnt), dim3(320), 0, s, its, cnt, (const uint8_t*)(d + o_blob), d_out);
            else if (cnt < kQoiWideBelow) hipLaunchKernelGGL(k_qoi_decode<4>, dim3(cnt), dim3(256), 0, s, its, cnt, (const uint8_t*)(d + o_blob), d_out);
            else                     hipLaunchKernelGGL(k_qoi_decode<1>, dim3(cnt), dim3(64), 0, s, its, cnt, (const uint8_t*)(d + o_blob), d_out);
        };
        // A file in page-locked memory (gamut_hip_host_malloc_pinned, hipHostRegister ...) goes up from where it is: no staging copy -- a
        // 1080p file is 4-8 MB, and 16 host threads copy about as fast as PCIe moves.  The slack behind such a file is whatever the device
        // buffer held: the lanes only need it readable (GAMUT_HIP_QOI_SLACK), the decode stops at the stream's end.
        // Pageable files are gathered into the pinned image by the host threads and go up in runs of >= 32 MB (or >= one file per thread): a
        // copy per file (2-5 MB) moves 40-45 GB/s over a link that does 57 with large ones (tools/microbench/h2d_streams.hip); the next run is
        // gathered while this one is on its way.
        auto upload = [&](int k0, int k1, hipStream_t s) {
            for (int c0 = k0; c0 < k1; ) {
                int c1 = c0; size_t bytes = 0; bool any_pinned = false;
                while (c1 < k1 && (c1 - c0 < workers || bytes < ((size_t)32 << 20))) {
                    bytes += items[(size_t)c1].size;
                    any_pinned = any_pinned || host_range_is_pinned(data[src[(size_t)c1]], items[(size_t)c1].size);
                    ++c1;
                }
                parallel_for(c1 - c0, workers, [&](int, int j) {
                    (void)hipSetDevice(dev);
                    const int k = c0 + j;
                    const size_t at = o_blob + items[(size_t)k].begin, nb = (size_t)items[(size_t)k].size + kQoiSlack;
                    if (any_pinned && host_range_is_pinned(data[src[(size_t)k]], items[(size_t)k].size)) {
                        if (hipMemcpyAsync(d + at, data[src[(size_t)k]], items[(size_t)k].size, hipMemcpyHostToDevice, s) != hipSuccess) { (void)hipGetLastError(); upload_failed = 1; }
                        return;
                    }
                    memcpy(h + at, data[src[(size_t)k]], items[(size_t)k].size);
                    memset(h + at + items[(size_t)k].size, 0, kQoiSlack);
                    if (any_pinned && hipMemcpyAsync(d + at, h + at, nb, hipMemcpyHostToDevice, s) != hipSuccess) { (void)hipGetLastError(); upload_failed = 1; }
                });
                if (!any_pinned) {                               // the run is one contiguous piece of the image
                    const size_t lo = o_blob + items[(size_t)c0].begin, hi = o_blob + items[(size_t)c1 - 1].begin + items[(size_t)c1 - 1].size + kQoiSlack;
                    if (hipMemcpyAsync(d + lo, h + lo, hi - lo, hipMemcpyHostToDevice, s) != hipSuccess) { (void)hipGetLastError(); upload_failed = 1; }
                }
                c0 = c1;
            }
        };
        constexpr int kGroup = 256;                             // one workgroup per compute unit: the shape k_qoi_pipe is quickest at (a stream ~ 12 ms)
        if (n <= kGroup) {
            upload(0, n, stream);
            if (upload_failed) return set_error(GAMUT_HIP_ERR_HIP, "qoi: upload failed");
            launch(0, n, stream);
            if (int rc = launch_status("qoi_decode")) return rc;
        } else {
            // A large batch is PCIe time (a 1080p file is 4-8 MB): the files go up group by group on a copy stream and every group is
            // decoded behind its own upload, beside the upload of the next one -- the kernels hide behind the copies except the last.
            static thread_local PerDevice<hipStream_t> copy_pd;
            hipStream_t& copy_stream = copy_pd.cur();
            if (!copy_stream) GAMUT_HIP_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
            std::vector<hipEvent_t> up;
            int rc = GAMUT_HIP_OK;
            for (int g0 = 0; g0 < n && rc == GAMUT_HIP_OK; g0 += kGroup) {
                const int cnt = n - g0 < kGroup ? n - g0 : kGroup;
                upload(g0, g0 + cnt, copy_stream);
                hipEvent_t e = nullptr;
                if (upload_failed || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); rc = set_error(GAMUT_HIP_ERR_HIP, "qoi: upload failed"); break; }
                up.push_back(e);
                if (hipEventRecord(e, copy_stream) != hipSuccess || hipStreamWaitEvent(stream, e, 0) != hipSuccess) { (void)hipGetLastError(); rc = set_error(GAMUT_HIP_ERR_HIP, "qoi: event failed"); break; }
                launch(g0, cnt, stream);
                rc = launch_status("qoi_decode");
            }
            (void)hipStreamSynchronize(copy_stream);
            (void)hipStreamSynchronize(stream);
            for (hipEvent_t e : up) (void)hipEventDestroy(e);
            if (rc != GAMUT_HIP_OK) return rc;
        }
        GAMUT_HIP_CHECK(hipStreamSynchronize(stream));         // the per-thread staging buffers are reused by the next call
    }
    if (first != GAMUT_HIP_OK) return set_error(first, "image %d: qoi: bad header or arguments", first_idx);
    return GAMUT_HIP_OK;
}

} // namespace
} // namespace gamut

using namespace gamut;

extern "C" {

#if QOI_PROFILE
int gamut_hip_qoi_profile(unsigned long long* out8)          // measurement builds only: A + chain, barriers, compaction, B, C + D
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_qoi_prof), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    unsigned long long z[8] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_qoi_prof), z, sizeof(z)) != hipSuccess;
}
#endif

int gamut_hip_qoi_read_header(const void* data, int size, gamut_hip_qoi_desc* desc)
{
    clear_error();
    if (!desc) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "qoi_read_header: null desc");
    return read_header((const uint8_t*)data, size, desc, 0);
}

int gamut_hip_qoi_decode_batch_device(const uint8_t* const* data, const int* size, int count, int channels,
                                      const int64_t* out_offset, uint8_t* out, gamut_hip_qoi_desc* descs, int* status_host, void* stream)
{
    clear_error();
    if (count < 0 || (count > 0 && (!data || !size || !out_offset || !out || !descs)))
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "qoi_decode_batch_device: bad arguments");
    if (count == 0) return GAMUT_HIP_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_error(GAMUT_HIP_ERR_NO_DEVICE, "no HIP device available (libgamut_hip has no CPU fallback)");
    try {                                                      // std::vector / bad_alloc must not escape a C entry point
        return decode_batch(data, size, count, channels, out_offset, out, descs, status_host, pick_stream(stream));
    } catch (...) {
        return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "qoi_decode_batch_device: out of host memory");
    }
}

int gamut_hip_qoi_decode_resident_device(const uint8_t* blob, int64_t blob_len, const int64_t* begin, const int* size,
                                         const gamut_hip_qoi_desc* descs, int count, int channels, const int64_t* out_offset,
                                         uint8_t* out, void* stream)
{
    clear_error();
    if (count < 0 || (channels != 0 && channels != 3 && channels != 4) || (count > 0 && (!blob || !begin || !size || !descs || !out_offset || !out)))
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "qoi_decode_resident_device: bad arguments");
    if (count == 0) return GAMUT_HIP_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_error(GAMUT_HIP_ERR_NO_DEVICE, "no HIP device available (libgamut_hip has no CPU fallback)");
    try {
        std::vector<QoiItem> items((size_t)count);
        for (int i = 0; i < count; ++i) {
            const gamut_hip_qoi_desc& d = descs[i];
            if (size[i] < kQoiHeader + kQoiPadding || begin[i] < 0 || begin[i] + (int64_t)size[i] + GAMUT_HIP_QOI_SLACK > blob_len)
                return set_error(GAMUT_HIP_ERR_INVALID_ARG, "qoi_decode_resident_device: stream %d (with its %d slack bytes) is outside the blob", i, GAMUT_HIP_QOI_SLACK);
            if (d.width == 0 || d.height == 0 || d.channels < 3 || d.channels > 4 || d.height >= kQoiPixelsMax / d.width)
                return set_error(GAMUT_HIP_ERR_DECODE, "qoi_decode_resident_device: stream %d: bad header", i);
            QoiItem it{}; it.begin = (uint64_t)begin[i]; it.out_off = out_offset[i]; it.size = (uint32_t)size[i];
            it.npx = d.width * d.height; it.channels = channels ? channels : d.channels;
            items[(size_t)i] = it;
        }
        // Asynchronous on `stream` (the files are resident: nothing here needs the host to wait).  The item table goes up through
        // one of four per-thread (pinned, device) slot pairs; a slot is taken again four calls later, after its event -- recorded
        // behind the kernel that read it -- has passed.
        struct TableSlot { uint8_t* h = nullptr; uint8_t* d = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool busy = false; };
        struct Slots { TableSlot s[4]; unsigned next = 0; };
        static thread_local PerDevice<Slots> slots_pd;
        Slots& sls = slots_pd.cur();
        TableSlot& sl = sls.s[sls.next++ & 3u];
        if (!sl.done) GAMUT_HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        if (sl.busy) { GAMUT_HIP_CHECK(hipEventSynchronize(sl.done)); sl.busy = false; }
        const size_t bytes = items.size() * sizeof(QoiItem);
        if (bytes > sl.cap) {
            if (sl.h) (void)hipHostFree(sl.h);
            if (sl.d) (void)hipFree(sl.d);
            sl.h = sl.d = nullptr; sl.cap = 0;
            const size_t want = bytes + bytes / 4 + 4096;
            void* hp = nullptr; void* dp = nullptr;
            if (hipHostMalloc(&hp, want, hipHostMallocDefault) != hipSuccess || hipMalloc(&dp, want) != hipSuccess) {
                if (hp) (void)hipHostFree(hp);
                (void)hipGetLastError();
                return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "qoi: allocation of the item table failed");
            }
            sl.h = (uint8_t*)hp; sl.d = (uint8_t*)dp; sl.cap = want;
        }
        memcpy(sl.h, items.data(), bytes);
        hipStream_t st = pick_stream(stream);
        GAMUT_HIP_CHECK(hipMemcpyAsync(sl.d, sl.h, bytes, hipMemcpyHostToDevice, st));
        const int n = (int)items.size();
        if (n < kQoiWideBelow && qoi_pipeline(n)) hipLaunchKernelGGL(k_qoi_pipe, dim3(n), dim3(320), 0, st, (const QoiItem*)sl.d, n, blob, out);
        else if (n < kQoiWideBelow) hipLaunchKernelGGL(k_qoi_decode<4>, dim3(n), dim3(256), 0, st, (const QoiItem*)sl.d, n, blob, out);
        else                   hipLaunchKernelGGL(k_qoi_decode<1>, dim3(n), dim3(64), 0, st, (const QoiItem*)sl.d, n, blob, out);
        if (int rc = launch_status("qoi_decode")) return rc;
        GAMUT_HIP_CHECK(hipEventRecord(sl.done, st));
        sl.busy = true;
        return GAMUT_HIP_OK;
    } catch (...) {
        return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "qoi_decode_resident_device: out of host memory");
    }
}

// drop-in for qoi_decode (qoi.d:448): malloc'd pixels or NULL
void* gamut_hip_qoi_decode(const void* data, int size, gamut_hip_qoi_desc* desc, int channels)
{
    clear_error();
    gamut_hip_qoi_desc local;
    if (!desc) desc = &local;
    if (read_header((const uint8_t*)data, size, desc, channels) != GAMUT_HIP_OK) return nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error(GAMUT_HIP_ERR_NO_DEVICE, "no HIP device available (libgamut_hip has no CPU fallback)"); return nullptr; }
    const int ch = channels ? channels : desc->channels;
    const size_t bytes = (size_t)desc->width * desc->height * ch;
    uint8_t* result = (uint8_t*)malloc(bytes ? bytes : 1);
    void* dout = nullptr;
    if (!result) { set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "qoi: out of memory"); return nullptr; }
    static thread_local PerDevice<DeviceScratch> pixels_dev_pd;           // per-thread, per-device staging that grows and stays
    DeviceScratch& pixels_dev = pixels_dev_pd.cur();
    dout = pixels_dev.get(bytes + 16, thread_stream());
    bool ok = dout != nullptr;
    if (!ok) { (void)hipGetLastError(); set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "qoi: device staging of %zu bytes failed", bytes); }
    const uint8_t* ptr = (const uint8_t*)data; const int64_t off = 0;
    hipStream_t st = thread_stream();
    try { ok = ok && decode_batch(&ptr, &size, 1, channels, &off, (uint8_t*)dout, desc, nullptr, st) == GAMUT_HIP_OK; }
    catch (...) { set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "qoi: out of host memory"); ok = false; }
    if (ok && (hipMemcpyAsync(result, dout, bytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) {
        set_error(GAMUT_HIP_ERR_HIP, "qoi: copy back failed"); ok = false;
    }
    if (!ok) { free(result); return nullptr; }
    return result;
}

} // extern "C"
