// png.hip -- K5-K7: PNG row de-filter (None/Sub/Up/Avg/Paeth) + bit-depth / alpha / endian expand.
//
// Replaces stbi__create_png_image_raw (stbdec.d:1406-1635) and the post passes
// stbi__compute_transparency(16) :1682-1730, stbi__expand_png_palette :1732-1765,
// stbi__convert_format(16) :916-1199, stbi__convert_16_to_8 / _8_to_16 :635-666.
//
// De-filter is the only non-trivially-parallel loop of the whole path: byte (x,y) depends on
// its left, upper and upper-left neighbours (Paeth), so the dependency graph is a 2-D
// wavefront.  Mapping used here:
//   * one workgroup of W waves per image; wave w owns the 64-row bands w, w+W, w+2W, ...
//   * inside a band lane = row; lane j runs one "iteration" (a 16-byte piece of its row) behind
//     lane j-1, so the upper neighbours arrive from lane j-1 with one DPP row shift per dword of
//     the piece it produced in the previous trip; the left neighbours are the lane's own
//     previous outputs (registers)
//   * lane 0 of band b needs the last row of band b-1, produced by another wave of the same
//     workgroup: it is read back from the output rows in HBM/L2 (full-row capacity, so no
//     back-pressure is needed), gated by a monotonic per-wave progress counter in LDS with a
//     workgroup-scope release (producer) / acquire (consumer) pair.  Consecutive bands run
//     ~64 iterations apart, so all W waves are busy at once.
//   * the per-row filter type is data: None/Sub/Up/Avg are one masked form evaluated four bytes
//     at a time (SWAR); the Paeth predictor (packed FP16, two channels per instruction) is
//     evaluated only while some row of the wave's band uses it (wave-uniform branch).
// Two kernels share this: k_png_defilter_ring (rows of at least one piece: all I/O staged through
// per-row LDS rings so that loads are coalesced and stores are whole aligned lines -- the fast
// path for every filter unit) and k_png_defilter<FB> (narrower rows: per-lane loads / stores).
// With zero neighbours outside the image the PNG formulas reproduce stb's first-row /
// first-pixel special cases exactly (stbdec.d:1381-1388, :1453-1465).
#include "common.hpp"

namespace gamut {
namespace {

typedef uint32_t u32;

#ifndef PNG_RING_ABL             // ablations of the ring form (wrong pixels; measurements only): 1 = the fast tiles fetch nothing, 2 = store nothing
#define PNG_RING_ABL 0
#endif
#ifndef PNG_PRED_OPAQUE
#define PNG_PRED_OPAQUE 1
#endif
#ifndef PNG_PAETH_ONE_ASM         // tuning knob (tools/variant.sh): the Paeth predictor's packed operations as one asm statement
#define PNG_PAETH_ONE_ASM 0           // (measured slower: 7.74 vs 7.58 ms, profiles/r05_png_ab5.txt -- the rigid block keeps the scheduler from interleaving the pixels)
#endif
#ifndef PNG_WAVES_N
#define PNG_WAVES_N 8
#endif
constexpr int PNG_WAVES = PNG_WAVES_N;           // waves per workgroup (= bands in flight per image); the ring kernel's LDS allows one workgroup per CU
constexpr int PUB = 8;                 // publish / check progress every PUB iterations (32 filter units)
constexpr int PF = 4;                  // loop trips of row data kept in flight per lane
constexpr int PUBLAG = 4;              // progress is published PUBLAG trips after the store it covers (see defilter_band)

struct DefilterArgs {
    const uint8_t* raw; int64_t raw_stride;      // inflated stream(s): per row 1 filter byte + wb bytes
    uint8_t* D; int64_t d_stride; int64_t d_pitch;   // de-filtered rows (pitch % 4 == 0)
    u32* status;                                 // per image, |= 1 on an invalid filter byte (stbdec.d:1438)
    u32 rows, wb;                                // rows, bytes per row
    u32 store_tail_masked;                       // 1: D rows are tight (fused output) -> never write past wb
    const int64_t* raw_offs; const int64_t* d_offs;  // optional (device): byte offset of image i's stream / rows instead of i * stride
    u32 nseg;                                    // > 1: every image is cut into up to nseg row segments (take_segment), one workgroup each
    // work-queue launch (k_png_defilter_queue): qstate[0] = next unit, qstate[QSTATE_HDR + img * nbands + band] = pieces of
    // that band's last row that are visible to every CU; zeroed by the launcher before every launch
    u32* qstate; u32 count, nbands, group;
};
constexpr u32 QSTATE_HDR = 16;
constexpr u32 STATUS_HANDOFF_TIMEOUT = 0x80000000u;      // a band never saw the band above it progress (a bug or a dead producer): reported, never hung
__device__ __forceinline__ const uint8_t* image_raw(const DefilterArgs& a, int img) { return a.raw + (a.raw_offs ? a.raw_offs[img] : (int64_t)img * a.raw_stride); }
__device__ __forceinline__ uint8_t* image_rows(const DefilterArgs& a, int img) { return a.D + (a.d_offs ? a.d_offs[img] : (int64_t)img * a.d_stride); }
// A workgroup's share: the whole image, or -- small batches -- the rows from one cut row to the next.  A cut row has filter None
// or Sub: it does not look at the row above, so the rows from there on de-filter like an image of their own.  Boundary k of an
// image's nseg segments is the cut row nearest to rows * k / nseg within rows / (2 nseg) - 1 rows of it (windows of different
// boundaries are disjoint), or missing; a segment whose own boundary is missing is empty, its rows stay with the segment
// before.  Every wave works this out for itself from the filter bytes (a few strided loads): nothing is handed from one
// kernel to the next (a table written by a kernel just before this one was seen stale by workgroups on other XCDs).
__device__ __forceinline__ u32 cut_row(const uint8_t* raw, u32 rows, u32 wb, u32 nseg, u32 k)      // wave-uniform; 0xFFFFFFFF = none
{
    const u32 lane = threadIdx.x & 63u;
    const u32 target = (u32)((uint64_t)rows * k / nseg), half = rows / (2u * nseg), reach = half ? half - 1u : 0u;
    u32 best = 0xFFFFFFFFu;                                      // distance << 1 | side: ties are settled the same way by every wave
    for (u32 d = lane; d <= reach && best == 0xFFFFFFFFu; d += 64u) {
        if (target + d < rows && raw[(int64_t)(target + d) * (wb + 1)] <= 1) best = d << 1;
        else if (d < target && raw[(int64_t)(target - d) * (wb + 1)] <= 1) best = d << 1 | 1u;
    }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const u32 other = (u32)__shfl_xor((int)best, o); best = other < best ? other : best; }
    return best == 0xFFFFFFFFu ? best : (best & 1u) ? target - (best >> 1) : target + (best >> 1);
}
__device__ __forceinline__ bool take_segment(DefilterArgs& a, int& img, const uint8_t*& raw, uint8_t*& D)
{
    u32 row0 = 0;
    if (a.nseg > 1) {
        img = (int)blockIdx.y;                   // grid = (segments, images)
        const u32 s = blockIdx.x;
        const uint8_t* whole = image_raw(a, img);
        if (s > 0) { row0 = cut_row(whole, a.rows, a.wb, a.nseg, s); if (row0 == 0xFFFFFFFFu) return false; }
        u32 row1 = a.rows;
        for (u32 k = s + 1; k < a.nseg; ++k) { const u32 c = cut_row(whole, a.rows, a.wb, a.nseg, k); if (c != 0xFFFFFFFFu) { row1 = c; break; } }
        a.rows = row1 - row0;
    } else img = (int)blockIdx.x;
    raw = image_raw(a, img) + (int64_t)row0 * (a.wb + 1);
    D = image_rows(a, img) + (int64_t)row0 * a.d_pitch;
    return true;
}

struct __attribute__((packed)) PackedU32 { u32 v; };      // a dword at any byte alignment

__device__ __forceinline__ u32 byte_of(const u32* g, int idx) { return (g[idx >> 2] >> ((idx & 3) * 8)) & 0xFFu; }

// stbi__paeth, stbdec.d:1390-1401
__device__ __forceinline__ u32 paeth(u32 a, u32 b, u32 c)
{
    const int pa = abs((int)b - (int)c), pb = abs((int)a - (int)c), pc = abs((int)a + (int)b - 2 * (int)c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// ---- four filter bytes at a time (one RGBA8 pixel per dword) ---------------------------------------------------
// bytewise (a + b) mod 256
__device__ __forceinline__ u32 add_bytes(u32 x, u32 y)
{
    return ((x & 0x7f7f7f7fu) + (y & 0x7f7f7f7fu)) ^ ((x ^ y) & 0x80808080u);
}
// bytewise floor((a + b) / 2), the Avg predictor (9-bit sum, stbdec.d:1497)
__device__ __forceinline__ u32 avg_bytes(u32 a, u32 b)
{
    return __builtin_amdgcn_lerp(a, b, 0u);         // v_lerp_u8: per byte (a + b + (c & 1)) >> 1 with a 9-bit sum -- c = 0: the floor
}
// ---- stbi__paeth (stbdec.d:1390-1401) on two channels at once, in packed FP16 ---------------------------------
// A byte n is carried as the half-precision number 1024 + n, whose bit pattern is simply 0x6400 | n (ulp = 1 in
// [1024, 2048)), so bytes <-> halves are single v_perm_b32 byte shuffles.  All differences (|.| <= 510), the 0/1
// selectors and the selected value are small integers, exact in FP16; the bias cancels in every difference.
// FP16 buys free negation (VOP3P neg modifiers: |t| = max(t, -t) is one instruction): 9 packed instructions per channel
// pair up to the two differences whose signs decide.  Written as inline asm: the optimizer otherwise rewrites the
// arithmetic into per-channel compares and selects (3x the count).
#define PKF2(name, text) __device__ __forceinline__ u32 name(u32 x, u32 y) { u32 r; asm(text : "=v"(r) : "v"(x), "v"(y)); return r; }
PKF2(pkf_add,       "v_pk_add_f16 %0, %1, %2")
PKF2(pkf_sub,       "v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")
PKF2(pkf_min,       "v_pk_min_f16 %0, %1, %2")
#undef PKF2
__device__ __forceinline__ u32 pkf_abs(u32 t) { u32 r; asm("v_pk_max_f16 %0, %1, %1 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(t)); return r; }
// The predictor of ANY row of a band that has Paeth rows, four channels at once.  The two comparisons of stbi__paeth (in the
// order libpng evaluates them: "pb < pa ? b : a", then "pc < min(pa, pb) ? c : that" -- the same choice, ties included)
// are differences whose SIGN is the answer, and v_perm_b32 can fill a byte with the sign bit of a half (selectors 8..11):
// one permute per comparison turns the four signs into a byte mask, and the selection runs on the packed pixels themselves
// (two v_bfi for four channels).  The permute's selector is a per-lane register, so the other filters ride along for free:
// their rows use constant selectors (0x0c = 0x00, 0x0d = 0xff) that force the masks -- Sub: (0, 0) -> a; Up: (ff, 0) -> b;
// Avg / None: (.., ff) -> the third candidate, which is c for Paeth rows, avg(a, b) for Avg rows and 0 for None rows.
// The two channel pairs (L: bytes 0,1  H: bytes 2,3) are written interleaved so that no packed instruction consumes the
// result of the one just before it (gfx950 needs a wait state there).
struct RowFilter { u32 mA, mB, mAvg, mP; u32 sel1, sel2; };                  // None/Sub/Up/Avg as one masked form; the Paeth-band form
__device__ __forceinline__ RowFilter row_filter(u32 f)
{
    const u32 SIGNS = 0x0b0a0908u, ZERO = 0x0c0c0c0cu, ONES = 0x0d0d0d0du, FF = 0xFFFFFFFFu;
    RowFilter r;
    r.mA = f == 1 ? FF : 0u; r.mB = f == 2 ? FF : 0u; r.mAvg = f == 3 ? FF : 0u; r.mP = f == 4 ? FF : 0u;
    r.sel1 = f == 4 ? SIGNS : f == 2 ? ONES : ZERO;
    r.sel2 = f == 4 ? SIGNS : (f == 0 || f == 3) ? ONES : ZERO;
    return r;
}
__device__ __forceinline__ u32 bfi(u32 m, u32 x, u32 y) { return (m & x) | (~m & y); }              // m ? x : y, bit by bit
__device__ __forceinline__ u32 paeth_band_pred(const RowFilter& f, u32 a, u32 b, u32 c)
{
    const u32 BIAS = 0x64646464u, LO = 0x04010400u, HI = 0x04030402u;                   // bytes -> halves 0x64nn
    const u32 aL = __builtin_amdgcn_perm(BIAS, a, LO), aH = __builtin_amdgcn_perm(BIAS, a, HI);
    const u32 bL = __builtin_amdgcn_perm(BIAS, b, LO), bH = __builtin_amdgcn_perm(BIAS, b, HI);
    const u32 cL = __builtin_amdgcn_perm(BIAS, c, LO), cH = __builtin_amdgcn_perm(BIAS, c, HI);   // = the b of the pixel before: no new instruction
#if !PNG_PAETH_ONE_ASM
    const u32 t1L = pkf_sub(bL, cL),        t1H = pkf_sub(bH, cH);
    const u32 t2L = pkf_sub(aL, cL),        t2H = pkf_sub(aH, cH);
    const u32 paL = pkf_abs(t1L),           paH = pkf_abs(t1H);                         // |p - a|   (p = a + b - c)
    const u32 t3L = pkf_add(t1L, t2L),      t3H = pkf_add(t1H, t2H);
    const u32 pbL = pkf_abs(t2L),           pbH = pkf_abs(t2H);                         // |p - b|
    const u32 pcL = pkf_abs(t3L),           pcH = pkf_abs(t3H);                         // |p - c|
    const u32 d1L = pkf_sub(pbL, paL),      d1H = pkf_sub(pbH, paH);                    // < 0: b over a
    const u32 mL  = pkf_min(paL, pbL),      mH  = pkf_min(paH, pbH);
    const u32 d2L = pkf_sub(pcL, mL),       d2H = pkf_sub(pcH, mH);                     // < 0: c over either  (x - x = +0: ties keep)
    const u32 m1 = __builtin_amdgcn_perm(d1H, d1L, f.sel1), m2 = __builtin_amdgcn_perm(d2H, d2L, f.sel2);
#else
    // One asm statement for the eighteen packed operations and the two permutes: between asm statements the compiler counts no wait
    // states (it does not look inside them), so with one statement per operation every consumer of a packed result got an s_nop in
    // front of it -- nine per 16-byte piece -- although another operation always stood between the two.  In here no operation reads
    // the result of the one just before it (the one wait state gfx950 asks for behind a packed operation).
    u32 t1L, t1H, t2L, t2H, paL, paH, d1L, d1H, m1, m2;
    asm("v_pk_add_f16 %[t1L], %[bL], %[cL] neg_lo:[0,1] neg_hi:[0,1]\n\t"          // t1 = b - c
        "v_pk_add_f16 %[t1H], %[bH], %[cH] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f16 %[t2L], %[aL], %[cL] neg_lo:[0,1] neg_hi:[0,1]\n\t"          // t2 = a - c
        "v_pk_add_f16 %[t2H], %[aH], %[cH] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_max_f16 %[paL], %[t1L], %[t1L] neg_lo:[0,1] neg_hi:[0,1]\n\t"        // pa = |p - a| = |t1|   (p = a + b - c)
        "v_pk_max_f16 %[paH], %[t1H], %[t1H] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f16 %[t1L], %[t1L], %[t2L]\n\t"                                  // t3 = t1 + t2            (in t1)
        "v_pk_add_f16 %[t1H], %[t1H], %[t2H]\n\t"
        "v_pk_max_f16 %[t2L], %[t2L], %[t2L] neg_lo:[0,1] neg_hi:[0,1]\n\t"        // pb = |p - b| = |t2|     (in t2)
        "v_pk_max_f16 %[t2H], %[t2H], %[t2H] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_max_f16 %[t1L], %[t1L], %[t1L] neg_lo:[0,1] neg_hi:[0,1]\n\t"        // pc = |p - c| = |t3|     (in t1)
        "v_pk_max_f16 %[t1H], %[t1H], %[t1H] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f16 %[d1L], %[t2L], %[paL] neg_lo:[0,1] neg_hi:[0,1]\n\t"        // d1 = pb - pa: < 0: b over a
        "v_pk_add_f16 %[d1H], %[t2H], %[paH] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_min_f16 %[paL], %[paL], %[t2L]\n\t"                                  // m = min(pa, pb)         (in pa)
        "v_pk_min_f16 %[paH], %[paH], %[t2H]\n\t"
        "v_pk_add_f16 %[t1L], %[t1L], %[paL] neg_lo:[0,1] neg_hi:[0,1]\n\t"        // d2 = pc - m: < 0: c over either  (x - x = +0: ties keep)
        "v_pk_add_f16 %[t1H], %[t1H], %[paH] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_perm_b32 %[m1], %[d1H], %[d1L], %[sel1]\n\t"
        "v_perm_b32 %[m2], %[t1H], %[t1L], %[sel2]"
        : [t1L] "=&v"(t1L), [t1H] "=&v"(t1H), [t2L] "=&v"(t2L), [t2H] "=&v"(t2H), [paL] "=&v"(paL), [paH] "=&v"(paH),
          [d1L] "=&v"(d1L), [d1H] "=&v"(d1H), [m1] "=&v"(m1), [m2] "=&v"(m2)
        : [aL] "v"(aL), [aH] "v"(aH), [bL] "v"(bL), [bH] "v"(bH), [cL] "v"(cL), [cH] "v"(cH), [sel1] "v"(f.sel1), [sel2] "v"(f.sel2));
#endif
    u32 third;                                                                           // (c & mP) | (avg & mAvg): two masks of the lane -- v_and + v_and_or,
    asm("v_and_b32 %0, %1, %2\n\tv_and_or_b32 %0, %3, %4, %0" : "=&v"(third) : "v"(c), "v"(f.mP), "v"(avg_bytes(a, b)), "v"(f.mAvg));   // which the compiler turns into three selects

    u32 pred = bfi(m2, third, bfi(m1, b, a));
#if PNG_PRED_OPAQUE
    asm("" : "+v"(pred));               // keeps the selection one instruction: left to see through it, the compiler fuses it with the byte-wise add's
#endif                                  // `& 0x7f7f7f7f` as two three-input operations AND computes it again for the `^` (11 -> 10 instructions per pixel)
    return pred;
}

// 4 bytes of the 16-byte piece v[0..3] starting at byte offset O (bytes past the piece read as zero)
template <int O> __device__ __forceinline__ u32 bytes_at(const u32 (&v)[4])
{
    constexpr int k = O >> 2, sh = O & 3;
    if constexpr (sh == 0) return v[k];
    else if constexpr (k == 3) return v[3] >> (8 * sh);
    else return __builtin_amdgcn_alignbyte(v[k + 1], v[k], sh);
}
// 4 bytes of the byte stream  prev[0..3] ++ cur[0..3]  starting at byte offset 16 + O, O in [-16, 12]
template <int O> __device__ __forceinline__ u32 stream_at(const u32 (&prev)[4], const u32 (&cur)[4])
{
    constexpr int P = 16 + O, k = P >> 2, sh = P & 3;
    const u32 lo = k < 4 ? prev[k] : cur[k - 4];
    if constexpr (sh == 0) return lo;
    else { const u32 hi = (k + 1) < 4 ? prev[k + 1] : cur[k + 1 - 4]; return __builtin_amdgcn_alignbyte(hi, lo, sh); }
}

template <bool PAETH>
__device__ __forceinline__ u32 defilter4(const RowFilter& f, u32 x, u32 a, u32 b, u32 c)      // four bytes at once
{
    if constexpr (PAETH) return add_bytes(x, paeth_band_pred(f, a, b, c));
    else                 return add_bytes(x, (a & f.mA) | (b & f.mB) | (avg_bytes(a, b) & f.mAvg));
}

// De-filter one 16-byte piece of a row (stbdec.d:1484-1503, any filter unit FB = bytes per pixel): rg = raw bytes,
// bg = the piece above, po / pb = the previous piece of this row (already de-filtered) / of the row above.
// Byte i depends on bytes i-FB, so any FB (<= 4) consecutive bytes can be computed together as one dword, whatever the
// pixel alignment of the piece: FB >= 4 walks the piece dword by dword and fetches "left" / "upper-left" at byte
// distance FB from the stream; FB < 4 walks it in steps of FB bytes (FB = 3: offsets 0,3,6,9,12 and 13 -- the last step
// recomputes two bytes, identically).  Lanes of a dword beyond the step's FB bytes compute garbage that is dropped.
template <int FB, bool PAETH>
__device__ __forceinline__ void filter_piece(const RowFilter& f, const u32 (&rg)[4], const u32 (&bg)[4],
                                             const u32 (&po)[4], const u32 (&pb)[4], u32 (&og)[4])
{
    if constexpr (FB == 4 || FB == 8) {
        constexpr int D = FB / 4;                      // dwords back
        #pragma unroll
        for (int p = 0; p < 4; ++p)
            og[p] = defilter4<PAETH>(f, rg[p], p >= D ? og[p - D] : po[4 + p - D], bg[p], p >= D ? bg[p - D] : pb[4 + p - D]);
    } else if constexpr (FB == 6) {
        u32 a, c;
        a = stream_at<0 - 6>(po, og);  c = stream_at<0 - 6>(pb, bg);  og[0] = defilter4<PAETH>(f, rg[0], a, bg[0], c);
        a = stream_at<4 - 6>(po, og);  c = stream_at<4 - 6>(pb, bg);  og[1] = defilter4<PAETH>(f, rg[1], a, bg[1], c);
        a = stream_at<8 - 6>(po, og);  c = stream_at<8 - 6>(pb, bg);  og[2] = defilter4<PAETH>(f, rg[2], a, bg[2], c);
        a = stream_at<12 - 6>(po, og); c = stream_at<12 - 6>(pb, bg); og[3] = defilter4<PAETH>(f, rg[3], a, bg[3], c);
    } else if constexpr (FB == 3) {
        const u32 b0 = bytes_at<0>(bg), b1 = bytes_at<3>(bg), b2 = bytes_at<6>(bg), b3 = bytes_at<9>(bg), b4 = bytes_at<12>(bg), b5 = bytes_at<13>(bg);
        const u32 s0 = defilter4<PAETH>(f, bytes_at<0>(rg),  po[3] >> 8, b0, pb[3] >> 8);
        const u32 s1 = defilter4<PAETH>(f, bytes_at<3>(rg),  s0, b1, b0);
        const u32 s2 = defilter4<PAETH>(f, bytes_at<6>(rg),  s1, b2, b1);
        const u32 s3 = defilter4<PAETH>(f, bytes_at<9>(rg),  s2, b3, b2);
        const u32 s4 = defilter4<PAETH>(f, bytes_at<12>(rg), s3, b4, b3);
        const u32 a5 = __builtin_amdgcn_perm(s4, s3, 0x0c040201u);                 // stream bytes 10, 11 (s3) and 12 (s4)
        const u32 s5 = defilter4<PAETH>(f, bytes_at<13>(rg), a5, b5, bytes_at<10>(bg));
        og[0] = __builtin_amdgcn_perm(s1, s0, 0x04020100u);                        // bytes 0,1,2 | 3
        og[1] = __builtin_amdgcn_perm(s2, s1, 0x05040201u);                        // 4,5 | 6,7
        og[2] = __builtin_amdgcn_perm(s3, s2, 0x06050402u);                        // 8 | 9,10,11
        og[3] = __builtin_amdgcn_perm(s5, s4, 0x06050400u);                        // 12 | 13,14,15
    } else if constexpr (FB == 2) {
        u32 s[8], bprev = pb[3] >> 16, sprev = po[3] >> 16;
        #pragma unroll
        for (int u = 0; u < 8; ++u) {
            const u32 x = (u & 1) ? rg[u >> 1] >> 16 : rg[u >> 1], b = (u & 1) ? bg[u >> 1] >> 16 : bg[u >> 1];
            s[u] = defilter4<PAETH>(f, x, sprev, b, bprev);
            sprev = s[u]; bprev = b;
        }
        #pragma unroll
        for (int k = 0; k < 4; ++k) og[k] = __builtin_amdgcn_perm(s[2 * k + 1], s[2 * k], 0x05040100u);
    } else {                                           // FB == 1: every byte hangs on the one before it
        static_assert(FB == 1, "filter unit");
        u32 s[16], bprev = pb[3] >> 24, sprev = po[3] >> 24;
        #pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int sh = 8 * (u & 3);
            const u32 x = sh ? rg[u >> 2] >> sh : rg[u >> 2], b = sh ? bg[u >> 2] >> sh : bg[u >> 2];
            s[u] = defilter4<PAETH>(f, x, sprev, b, bprev);
            sprev = s[u]; bprev = b;
        }
        #pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32 lo = __builtin_amdgcn_perm(s[4 * k + 1], s[4 * k], 0x0c0c0400u), hi = __builtin_amdgcn_perm(s[4 * k + 3], s[4 * k + 2], 0x0c0c0400u);
            og[k] = lo | (hi << 16);
        }
    }
}

// The same for a 12-byte piece of an 8-bit RGB row = four whole pixels (the RGB -> RGBA ring kernel walks rows in 4-pixel
// pieces so that a piece of the stream is exactly one 16-byte chunk of the output): every piece starts on a pixel
// boundary, four serial steps.  px[k] = de-filtered pixel k in the low three bytes (top byte: garbage).
template <bool PAETH>
__device__ __forceinline__ void filter_piece12(const RowFilter& f, const u32 (&rg)[4], const u32 (&bg)[4],
                                               const u32 (&po)[4], const u32 (&pb)[4], u32 (&og)[4], u32 (&px)[4])
{
    const u32 b0 = bytes_at<0>(bg), b1 = bytes_at<3>(bg), b2 = bytes_at<6>(bg), b3 = bytes_at<9>(bg);
    px[0] = defilter4<PAETH>(f, bytes_at<0>(rg), po[2] >> 8, b0, pb[2] >> 8);
    px[1] = defilter4<PAETH>(f, bytes_at<3>(rg), px[0], b1, b0);
    px[2] = defilter4<PAETH>(f, bytes_at<6>(rg), px[1], b2, b1);
    px[3] = defilter4<PAETH>(f, bytes_at<9>(rg), px[2], b3, b2);
    og[0] = __builtin_amdgcn_perm(px[1], px[0], 0x04020100u);                      // bytes 0,1,2 | 3
    og[1] = __builtin_amdgcn_perm(px[2], px[1], 0x05040201u);                      // 4,5 | 6,7
    og[2] = __builtin_amdgcn_perm(px[3], px[2], 0x06050402u);                      // 8 | 9,10,11
    og[3] = 0;
}

// shift a dword one lane up the wave (lane j receives lane j-1's value; lane 0 keeps `fill`):
// one DPP move, wave_shr:1 (gfx9 DPP control 0x138), bound_ctrl off so lane 0 retains `old`
__device__ __forceinline__ u32 from_lane_below(u32 v, u32 fill)
{
    return (u32)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xF, 0xF, false);
}

// One 64-row band.  PAETH = some row of the band uses the Paeth filter (wave-uniform, hoisted out of the byte loop).
template <int FB, int W, bool PAETH>
__device__ __forceinline__ void defilter_band(const DefilterArgs& a, const uint8_t* raw, uint8_t* D, u32* prog, u32 band,
                                              int wave, int lane, u32 niter, u32 f, bool row_live)
{
    const u32 seq = band / W;
    const u32 row = band * 64 + lane;
    const u32 ma = (f == 1 || f == 3 || f == 4) ? 0xFFu : 0u;
    const u32 mb = (f == 2 || f == 3 || f == 4) ? 0xFFu : 0u;
    const u32 sh = (f == 3) ? 1u : 0u;
    const bool is_paeth = f == 4;

    const uint8_t* rbytes = raw + (int64_t)(row_live ? row : 0) * (a.wb + 1) + 1;
    uint8_t* drow = D + (int64_t)(row_live ? row : 0) * a.d_pitch;
    const uint8_t* dprev = band > 0 ? D + (int64_t)(band * 64 - 1) * a.d_pitch : D;     // row above lane 0; band 0: anything valid, masked to 0
    const int prod_wave = (wave + W - 1) % W;
    const u32 prod_base = (band > 0 ? (band - 1) / W : 0) * niter;
    const u32 full_iters = a.wb / (4 * FB);                              // iterations whose 4*FB bytes all lie inside the row

    u32 outp[FB], bp[FB];                     // previous iteration: own outputs, upper-row values
    #pragma unroll
    for (int i = 0; i < FB; ++i) { outp[i] = 0; bp[i] = 0; }

    // Software prefetch: the loads of loop trip T+PF are issued at trip T into register set T % PF (static after
    // unrolling by PF), so a lane always has PF iterations of its row in flight.  The loads are unconditional
    // (addresses clamped into the row, results of dead lanes unused): a branch around a load makes the compiler drain
    // the memory queue (s_waitcnt vmcnt(0)) on every trip.  gfx950 handles the byte-unaligned dword / dwordx4
    // accesses natively (tools/unaligned_probe.hip).
    u32 rset[PF][FB], dset[PF][FB];
    auto issue_loads = [&](u32 Tn, u32 (&rs)[FB], u32 (&ds)[FB]) {
        int itn = (int)Tn - lane;
        itn = itn < 0 ? 0 : itn;
        const u32 itc = min((u32)itn, full_iters > 0 ? full_iters - 1 : 0u);       // always a fully readable group
        const uint8_t* pn = rbytes + (int64_t)itc * (4 * FB);
        const uint8_t* dn = dprev + (int64_t)min(Tn, niter - 1) * (4 * FB);        // lane 0's iteration is Tn: wave-uniform address
        #pragma unroll
        for (int i = 0; i < FB; ++i) {
            rs[i] = full_iters ? reinterpret_cast<const PackedU32*>(pn)[i].v : 0u;     // (uniform) rows shorter than one group use the tail path only
            ds[i] = band > 0 ? reinterpret_cast<const u32*>(dn)[i] : 0u;               // (uniform) band 0 has no row above it
        }
    };
    auto wait_for_band_above = [&](u32 upto) {      // wave-uniform: rows of the band above are visible up to iteration `upto`
        const u32 need = prod_base + min(niter, upto);
        while (__hip_atomic_load(&prog[prod_wave], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
            __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    if (band > 0) wait_for_band_above(PUB + PF);
    #pragma unroll
    for (int u = 0; u < PF; ++u) issue_loads((u32)u, rset[u], dset[u]);

    const u32 T_end = niter + 63 + PUBLAG;
    for (u32 T0 = 0; T0 < T_end; T0 += PF) {
        #pragma unroll
        for (int u = 0; u < PF; ++u) {
            const u32 T = T0 + u;
            const int it = (int)T - lane;
            const bool live = row_live && it >= 0 && it < (int)niter;

            u32 rg[FB], bg[FB];
            #pragma unroll
            for (int i = 0; i < FB; ++i) rg[i] = rset[u][i];
            if (live && it >= (int)full_iters) {          // ragged last group of a row: bytewise, zero-padded (never reads past the row)
                const uint8_t* pt = rbytes + (int64_t)it * (4 * FB);
                const u32 nb = a.wb - (u32)it * (4 * FB);
                #pragma unroll
                for (int i = 0; i < FB; ++i) rg[i] = 0;
                #pragma unroll
                for (int i = 0; i < 4 * FB; ++i) if ((u32)i < nb) rg[i >> 2] |= (u32)pt[i] << ((i & 3) * 8);
            }
            // upper row: lane j-1's outputs of its previous loop trip; lane 0 takes the band above (prefetched)
            #pragma unroll
            for (int i = 0; i < FB; ++i) bg[i] = from_lane_below(outp[i], dset[u][i]);

            // consumer side of the band hand-off, then refill this register set for trip T+PF
            if (band > 0 && T + PF < niter && ((T + PF) % PUB) == 0) wait_for_band_above(T + PF + PUB);
            issue_loads(T + PF, rset[u], dset[u]);

            u32 og[FB];
            #pragma unroll
            for (int i = 0; i < FB; ++i) og[i] = 0;
            #pragma unroll
            for (int k = 0; k < 4 * FB; ++k) {        // byte k of the group; its left / upper-left neighbours are FB bytes back
                const u32 x  = byte_of(rg, k);
                const u32 bb = byte_of(bg, k);
                const u32 aa = k >= FB ? byte_of(og, k - FB) : byte_of(outp, 3 * FB + k);
                u32 pred = ((aa & ma) + (bb & mb)) >> sh;
                if constexpr (PAETH) {
                    const u32 cc = k >= FB ? byte_of(bg, k - FB) : byte_of(bp, 3 * FB + k);
                    const u32 pp = paeth(aa, bb, cc);
                    pred = is_paeth ? pp : pred;
                }
                og[k >> 2] |= ((x + pred) & 0xFFu) << ((k & 3) * 8);
            }

            if (live) {
                #pragma unroll
                for (int i = 0; i < FB; ++i) { outp[i] = og[i]; bp[i] = bg[i]; }
                uint8_t* dst = drow + (int64_t)it * (4 * FB);
                const u32 valid = min(4u * FB, a.wb - (u32)it * (4 * FB));     // bytes of this group inside the row
                if (valid == 4 * FB || !a.store_tail_masked) {
                    if constexpr (FB == 4) *reinterpret_cast<uint4*>(dst) = make_uint4(og[0], og[1], og[2], og[3]);
                    else if constexpr (FB == 2 || FB == 6) {
                        #pragma unroll
                        for (int i = 0; i < FB; i += 2) *reinterpret_cast<uint2*>(dst + 4 * i) = make_uint2(og[i], og[i + 1]);
                    } else if constexpr (FB == 8) {
                        *reinterpret_cast<uint4*>(dst) = make_uint4(og[0], og[1], og[2], og[3]);
                        *reinterpret_cast<uint4*>(dst + 16) = make_uint4(og[4], og[5], og[6], og[7]);
                    } else {
                        #pragma unroll
                        for (int i = 0; i < FB; ++i) reinterpret_cast<u32*>(dst)[i] = og[i];
                    }
                } else {
                    for (u32 i = 0; i < valid; ++i) dst[i] = (uint8_t)(og[i >> 2] >> ((i & 3) * 8));
                }
            }

            // producer side.  Lane 63 stored iteration T-63 in this trip; what is published is the iteration it stored
            // PUBLAG trips ago: every trip issues at least two loads after its store, and the vector-memory counter
            // retires in order, so once at most 2*PUBLAG operations are outstanding that older store has been
            // acknowledged by the L2 -- without draining the prefetches that are in flight behind it.
            const int itp = (int)T - 63 - PUBLAG;
            if (itp >= 0 && itp < (int)niter && (((itp + 1) % PUB) == 0 || itp == (int)niter - 1)) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PUBLAG) : "memory");
                if (lane == 63)
                    __hip_atomic_store(&prog[wave], seq * niter + (u32)itp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
}

template <int FB, int W>
__global__ __launch_bounds__(W * 64) void k_png_defilter(DefilterArgs a)
{
    __shared__ u32 prog[W];                       // cumulative iterations finished (and visible) by each wave's lane 63
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int img; const uint8_t* raw; uint8_t* D;
    if (!take_segment(a, img, raw, D)) return;
    const u32 npix = a.wb / FB;                   // filter units per row
    const u32 niter = (npix + 3) / 4;
    const u32 nbands = (a.rows + 63) / 64;
    if (threadIdx.x < W) prog[threadIdx.x] = 0;
    __syncthreads();

    for (u32 band = wave; band < nbands; band += W) {
        const u32 row = band * 64 + lane;
        const bool row_live = row < a.rows;
        u32 f = row_live ? raw[(int64_t)row * (a.wb + 1)] : 0;
        if (f > 4) { if (a.status) atomicOr(a.status + img, 1u); f = 0; }
        if (__any(f == 4)) defilter_band<FB, W, true >(a, raw, D, prog, band, wave, lane, niter, f, row_live);
        else               defilter_band<FB, W, false>(a, raw, D, prog, band, wave, lane, niter, f, row_live);
    }
}

// =====================================================================================================
// FB == 4 (8-bit RGBA, 16-bit grey+alpha): LDS-staged, coalesced I/O.
//
// With lane = row, a plain per-lane load/store touches 64 different cache lines per wave instruction and the
// vector-memory address path serialises them (measured: 1.2 TB/s regardless of the filter mix, VALU idle).
// Here each row of the band owns a ring of RING 16-byte pieces in LDS, piece `it` of a row living in slot it % RING.
// A wave works in tiles of TT loop trips (lane j handles iteration T - j in trip T):
//   * the raw pieces of the next tile -- a parallelogram in (row, byte) space -- are fetched into registers by
//     cooperative loads in which 8 consecutive lanes cover 128 contiguous bytes of one row, and dropped into the
//     rings at the start of the tile;
//   * in a trip a lane reads its raw piece from its ring and overwrites the slot with the de-filtered piece;
//   * at the end of the tile every row has finished exactly one more 128-BYTE-ALIGNED group of 8 pieces (iterations
//     8g .. 8g+7, computed during this tile and the one before); those groups are written back, 8 lanes per row.
// Writing whole aligned 128-byte lines matters: the same bytes stored as tile-shaped (16-byte-granular, unaligned)
// 128-byte runs ran at 2.0 TB/s store-only against 3.7 TB/s aligned (tools/variant.sh experiments).
constexpr int TT = 8;
constexpr int RING = 16;                          // pieces per row: a finished-but-unwritten group (<= 7 pieces) + the tile in flight (8)
constexpr int ROW_PITCH = RING * 16;              // 256 B: lane j's slot (T - j) % 16 => ds_read_b128 conflict-free across 8 consecutive lanes
#ifndef PNG_NT_LOADS             // the filtered stream is read exactly once (tuning knob, tools/variant.sh)
#define PNG_NT_LOADS 0
#endif
#ifndef PNG_NT_STORES
#define PNG_NT_STORES 1
#endif
#ifndef PNG_AL_PAETH             // line-aligned loads in bands with Paeth rows too?  No: those bands are bound by vector instructions (188 per 16-byte trip),
#define PNG_AL_PAETH 0           // and the byte-unaligned ds_write_b128 of the aligned drop runs at a lane per LDS cycle (SQ_LDS_IDX_ACTIVE x 3,
#endif                           // profiles/r04_png_sq.txt): 512 x 4K with random filters 8.18 ms with, 8.02 without; bands without Paeth rows are bound by HBM
#ifndef PNG_Q_NO_CHUNK           // ablation only (wrong pixels): the queue kernel without its loads of the row above the band
#define PNG_Q_NO_CHUNK 0
#endif

// RGBA = 8-bit RGB stream in, RGBA8 rows out (alpha = 255 inserted, stbdec.d:1504-1546, out_n == img_n + 1): the row is walked
// in pieces of IB = 12 stream bytes = 4 pixels = one 16-byte chunk of the output, the ring slots hold the expanded pixels,
// and everything about write-back, alignment and the hand-off between bands is the RGBA8 case; the row above the band comes
// back from the output as RGBA and is squeezed to RGB again (three byte permutes per trip).
// Q = work-queue launch: the band above may run on ANY compute unit of the device, so the hand-off is the placement-
// independent one: the band's last row (the only bytes another wave reads) is stored write-through (16-byte sc1 stores), its
// progress word is an agent-scope relaxed store behind the counted vmcnt wait, the consumer polls that word relaxed and reads
// the row with sc1 loads (no fence on either side, nothing depends on which XCD runs what).  prog = the image's progress
// words (one per band) instead of the workgroup's per-wave ones.
// AL = line-aligned loads of the filtered stream.  A row of the stream starts at an odd byte address (every row is wb + 1 bytes), so
// the 128 contiguous bytes a row needs per tile straddle two 128-byte lines of memory and every line is fetched by two consecutive
// tiles; with every wave slot of the chip streaming (512 x 4K images) the second touch misses the L2 more often than not --
// rocprofv3: 51.4 MB read per 33.2 MB image, 1.55 x (profiles/r04_png_reads.txt; 2.02 x with nontemporal loads, 1.09 x at batch 64).
// Here the eight lanes of a row fetch ONE whole aligned line per tile, as eight aligned 16-byte chunks; a chunk is dropped into the
// row's ring at the byte offset it has in the row (an unaligned ds_write_b128; the ring is a byte line of 256 bytes with 16-byte guards
// at both ends, a chunk that crosses the end is written at both places), and the part of the line that belongs to the NEXT tile (the
// chunks behind the row's phase P) waits in the lanes' registers for the next drop: every line is read exactly once, nothing else
// changes -- pieces, DPP hand-over, write-back are those of the row-aligned grid.  The ring must start out zero (a row's first
// pieces are no longer all written by drops).
template <int FB, int W, bool PAETH, bool RGBA, bool Q = false, bool AL = false, bool LN = true>
__device__ __forceinline__ void defilter_band_ring(const DefilterArgs& a, const uint8_t* raw, uint8_t* D, u32* prog, uint8_t* ring, u32 band,
                                               int wave, int lane, u32 niter, u32 f, bool row_live, u32* status = nullptr)
{
    static_assert(!RGBA || FB == 3, "alpha insertion is the 8-bit RGB case");
    static_assert(!(AL && RGBA), "the alpha-inserting walk has 12-byte pieces: its ring is not a byte line of the row");
    constexpr int IB = RGBA ? 12 : 16;            // stream bytes per piece
    constexpr int PW = 4;                         // dwords per piece
    constexpr int PITCH = (AL || !LN) ? ROW_PITCH + 32 : ROW_PITCH;      // AL: guards; 288 keeps the per-lane ds_read_b128 pattern conflict-free ((T + lane) mod 16); !LN: room behind slot 15 (dwp below)
    constexpr int GUARD = AL ? 16 : 0;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    struct __attribute__((packed, aligned(1))) AnyVec { u32x4 v; };
    const u32 seq = Q ? 0u : band / W;
    const u32 row = band * 64 + lane;
    const RowFilter rf = row_filter(f);

    uint8_t* drow = D + (int64_t)(row_live ? row : 0) * a.d_pitch;
    const uint8_t* dprev = band > 0 ? D + (int64_t)(band * 64 - 1) * a.d_pitch : D;
    const int prod_wave = (wave + W - 1) % W;
    const u32 prod_base = Q ? 0u : (band > 0 ? (band - 1) / W : 0) * niter;
    u32* const my_flag = Q ? prog + band : prog + wave;
    u32* const prod_flag = Q ? prog + (band > 0 ? band - 1 : 0) : prog + prod_wave;
    const u32 full_iters = a.wb / IB;             // whole pieces of the stream = whole 16-byte chunks of the output row
    const u32 out_wb = RGBA ? (a.wb / 3) * 4 : a.wb;
    // pieces written back cooperatively: a partial last piece goes along when the destination rows are padded (scratch)
    const u32 wb_iters = a.store_tail_masked ? full_iters : niter;
    // Rows that do not start on a 128-byte line (tight rows of a width that is no multiple of 32 pixels: the reference's own layout): a group
    // of 8 pieces counted from the row's first byte straddles two lines of memory, every line is then written in two partial pieces a tile
    // apart -- 512 x 3848 x 2160 took 12 ms where 3840 takes 6 (profiles/r06_png_width_probe.txt).  The groups of such rows are the LINES of
    // memory instead: a row whose first byte is ph pieces into its line writes back pieces 8 g - ph .. 8 g - ph + 7 (wave-uniform switch;
    // the generic forms only -- the fast forms keep their per-lane constants for line-aligned rows).
    // LN (the launcher's verdict: every row of every image of the batch starts on a line) makes it a constant: the line-aligned kernels are
    // compiled exactly as before (with the test at run time in the one kernel the Paeth bands spilled 25 registers instead of 17: 7.51 -> 7.77 ms).
    const bool wb_lines = LN || ((reinterpret_cast<uintptr_t>(D) | (uintptr_t)(uint64_t)a.d_pitch) & 127u) == 0;

    // cooperative mapping: in transfer k (0..7) this lane handles row 8k + crow, piece cslot of that row's 8
    const int crow = lane >> 3, cslot = lane & 7;
    const u32 rows_left = a.rows - band * 64;                           // live rows in this band (may exceed 64)
    // row crow of the band if the image has it, else the band's first row; + 8k rows per transfer
    const uint8_t* craw = raw + (int64_t)(band * 64 + ((u32)crow < rows_left ? crow : 0)) * (a.wb + 1) + 1;
    uint8_t* cdst = D + (int64_t)(band * 64 + crow) * a.d_pitch;
    uint8_t* dband;                                                     // the band's first output row (wave-uniform: the fast write-back adds lane offsets)
    {
        const int64_t bo = (int64_t)(band * 64) * a.d_pitch;
        dband = D + (int64_t)(((uint64_t)(u32)__builtin_amdgcn_readfirstlane((int)((uint64_t)bo >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)(uint64_t)bo));
    }
    auto wb_phase = [&](int k) -> int {                                   // pieces between the line's first byte and the first byte of row 8 k + crow
        return wb_lines ? 0 : (int)(((reinterpret_cast<uintptr_t>(cdst) + (uintptr_t)((int64_t)(8 * k) * a.d_pitch)) >> 4) & 7u);
    };
    const int ph63 = wb_lines ? 0 : (int)(((reinterpret_cast<uintptr_t>(D) + (uintptr_t)((int64_t)(band * 64 + 63) * a.d_pitch)) >> 4) & 7u);
    // Rows that are not even 16-byte aligned (a width that is no multiple of 4 pixels: 1366 x 4 = 5464 bytes): every 16-byte piece of the row straddles two
    // 16-byte units of memory, and a store of it is two partial ones -- 2048 x 1366x768 took 6.5 ms where 1024x1024 takes 3.5 (profiles/r06_geometry_sweep_after.txt).
    // The write-back's units are then the 16-byte CHUNKS OF MEMORY: chunk m of a row whose first byte lies d dwords into its unit holds the row's dwords
    // 4 m - d .. 4 m - d + 3 -- the last d dwords of piece m - 1 and the first 4 - d of piece m, one LDS read at a dword offset (slot 15's chunk ends in a copy of
    // slot 0 behind it) -- and leaves as one aligned store; a row's first and last chunk go out dword by dword.  dwp: wave-uniform, only in the !LN kernels.
    const bool dwp = !LN && !wb_lines && ((reinterpret_cast<uintptr_t>(D) | (uintptr_t)(uint64_t)a.d_pitch) & 15u) != 0;
    auto dw_of = [&](int k) -> int { return dwp ? (int)(((reinterpret_cast<uintptr_t>(cdst) + (uintptr_t)((int64_t)(8 * k) * a.d_pitch)) >> 2) & 3u) : 0; };
    const int d63 = dwp ? (int)(((reinterpret_cast<uintptr_t>(D) + (uintptr_t)((int64_t)(band * 64 + 63) * a.d_pitch)) >> 2) & 3u) : 0;
    // pieces of the band's last row that the write-backs up to tile T0s have stored (what a hand-off may publish once they have arrived)
    auto stored_through = [&](int T0s) -> int { return (((T0s - 63 + ph63) & ~7) - ph63) + 8 - (d63 ? 1 : 0); };
    uint8_t* my_ring = ring + lane * PITCH + GUARD;
    uint8_t* co_ring = ring + crow * PITCH + GUARD;                     // + k * 8 * PITCH + slot * 16

    u32 outp[PW], bprev[PW];                      // previous piece of this lane's row (de-filtered) and of the row above it
    #pragma unroll
    for (int i = 0; i < PW; ++i) { outp[i] = 0; bprev[i] = 0; }

    // descriptors of the two rows that cross waves -- the row above the band (read) and the band's last row (written, Q).
    // Band 0 has no row above it: its descriptor is empty, every load through it returns zeros (no mask, no branch).
    const auto rs_prev = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(dprev), 0, band > 0 ? (int)(niter * 16) : 0, 0x00020000);
    const auto rs_last = __builtin_amdgcn_make_buffer_rsrc(D + (int64_t)(band * 64 + 63) * a.d_pitch, 0, (int)(niter * 16), 0x00020000);
    u32 seen = 0;                                   // Q: the producer's progress as last read (it only grows: most checks cost nothing)
    auto wait_for_band_above = [&](u32 upto) {
        const u32 need = prod_base + min(niter, upto);
        if constexpr (Q) {
            if (seen >= need) return;
            const uint64_t t0 = wall_clock64();
            for (;;) {
                seen = (u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(prod_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (seen >= need) break;
                __builtin_amdgcn_s_sleep(8);
                if (wall_clock64() - t0 > 400000000ull) {                 // 4 s of the 100 MHz clock: give up, say so, never hang
                    if (status && lane == 0) atomicOr(status, STATUS_HANDOFF_TIMEOUT);
                    seen = 0xFFFFFFFFu;
                    break;
                }
            }
        } else {
            while (__hip_atomic_load(prod_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
                __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    };
    auto publish = [&](u32 value) {
        if (lane == 63) {
            if constexpr (Q) __hip_atomic_store(my_flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else             __hip_atomic_store(my_flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    uint4 pre[TT];                                  // next tile's raw pieces (cooperative layout)
    // AL: per transfer k (row 8k + crow of the band): the address of "chunk 0" of the row = row start - S' (16-byte aligned, S' = 1..16
    // bytes in front of the row's first byte), S', and the row's phase P = position in its memory line of the last chunk a tile needs.
    // The general forms (al_row, prefetch_line, drop_al) recompute them where they are needed: the first and the last tiles of a band.
    // The tiles in between (nothing in front of a row, nothing behind it, 64 live rows) take the *_fast forms, which run on a few
    // per-row constants kept in registers: the tile overhead (drop + prefetch + write-back) was 580 of a tile's 2 080 instructions with
    // Paeth rows and of 1 340 without.
    uint4 pre_old[AL ? TT : 1];
    auto al_row = [&](int k, const uint8_t*& c0, int& Sp, int& P) {
        const u32 rowk = (u32)(8 * k + crow);
        const u32 rlive = rowk < rows_left ? rowk : 0u;                 // rows past the image read a live row of the band (unused)
        const uint8_t* row0 = raw + (int64_t)(band * 64 + rlive) * (a.wb + 1) + 1;
        const u32 S = (u32)(reinterpret_cast<uintptr_t>(row0) & 15u);
        Sp = S ? (int)S : 16;
        const u32 q = (u32)((reinterpret_cast<uintptr_t>(row0) - (u32)Sp) >> 4) & 7u;
        c0 = row0 - Sp;
        P = (int)((q - rowk) & 7u);
    };
    // fast forms: byte offset of the chunk prefetch_line(0) would fetch, from a wave-uniform base (+ 16 T0 per tile); ring offset of the
    // chunk drop_al(0) would write (its low 8 bits: + 16 T0 mod 256 flips bit 7 every other tile); which lanes drop the fresh line
    const uint8_t* ubase = nullptr;
    u32 pfo[AL ? TT : 1], freshbits = 0;          // per row: bits 0-23 the prefetch offset, bits 24-31 the ring offset (one register, not two: the
                                                  // fast tiles hold three sets of eight pieces as it is)
    if constexpr (AL) {
        // (a wave-uniform byte offset from `raw`, said so to the compiler half by half: the loads then take a scalar base and a
        // 32-bit lane offset; - 2048 keeps the lane offsets positive whatever a row's constants)
        const int64_t uo = (int64_t)band * 64 * (a.wb + 1) - 2048;
        ubase = raw + (int64_t)(((uint64_t)(u32)__builtin_amdgcn_readfirstlane((int)((uint64_t)uo >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)(uint64_t)uo));
        #pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint8_t* c0; int Sp, P;
            al_row(k, c0, Sp, P);
            const int cconst = -(8 * k + crow) + 8 - P + cslot;                            // prefetch_line(T0): chunk T0 + cconst
            const u32 fetch_off = (u32)((int64_t)(c0 - ubase) + (int64_t)cconst * 16);          // < 2^24: al_full
            const bool fresh = cslot <= P;
            const int dconst = cconst - (fresh ? 0 : 8);                                   // drop_al(T0): chunk T0 + dconst, row byte 16 (T0 + dconst) - S'
            pfo[k] = (fetch_off & 0xFFFFFFu) | ((u32)(dconst * 16 - Sp) << 24);
            freshbits |= fresh ? 1u << k : 0u;
            pre_old[k] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    auto prefetch_line = [&](int T0) {              // chunk (T0 - row) + 8 - P + cslot of every row: phase cslot of the line that ends tile T0's needs
        if constexpr (AL) {
            #pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint8_t* c0; int Sp, P;
                al_row(k, c0, Sp, P);
                int c = T0 - (8 * k + crow) + 8 - P + cslot;
                const int c_first = Sp == 16 ? 1 : 0, c_last = (int)((a.wb - 1 + (u32)Sp) >> 4);
                c = c < c_first ? c_first : c > c_last ? c_last : c;    // never an address outside the row (a chunk that holds a byte of it is inside its page)
                const u32x4 v = *reinterpret_cast<const u32x4*>(c0 + (int64_t)c * 16);
                pre[k] = make_uint4(v.x, v.y, v.z, v.w);
            }
        }
    };
    auto prefetch_fast = [&](u32 T0) {              // the same where no chunk can leave its row: one add and one load per row
        if constexpr (AL) {
            #pragma unroll
            for (int k = 0; k < 8; ++k) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(ubase + (size_t)(__umul24(pfo[k], 1u) + 16u * T0));
                pre[k] = make_uint4(v.x, v.y, v.z, v.w);
            }
        }
    };
    // the tiles that may take the fast forms (wave-uniform)
    const bool full64 = rows_left >= 64, al_full = AL && full64 && a.wb < 200000u;       // (64 rows of the stream + 4 KB inside 2^24 bytes)
    auto fast_prefetch_ok = [&](u32 T0) { return al_full && T0 >= 64 && T0 + 16 <= full_iters; };      // chunks T0 - 62 .. T0 + 15 of every row exist
    auto fast_drop_ok     = [&](u32 T0) { return al_full && T0 >= 72; };                                  // no chunk in front of its row
    auto fast_wb_ok       = [&](u32 T0) { return wb_lines && full64 && T0 >= 64 && T0 + 8 <= wb_iters; }; // every row writes a whole group, a whole line of memory
    // the row-aligned grid's own fast form: 64 live rows, every piece of the tile a whole piece inside its row -- the address is a
    // constant of the lane behind a wave-uniform pointer that moves by 8 rows less 8 pieces per transfer and a piece per trip
    u32 ploff = 0;
    const uint8_t* pbase = nullptr;
    if constexpr (!AL) {
        const int64_t uo = (int64_t)band * 64 * (a.wb + 1) + 1 - 64 * IB;
        pbase = raw + (int64_t)(((uint64_t)(u32)__builtin_amdgcn_readfirstlane((int)((uint64_t)uo >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)(uint64_t)uo));
        ploff = (u32)crow * (a.wb + 1) + (u32)(64 + cslot - crow) * IB;
    }
    auto prefetch_tile = [&](u32 T0) {
        if constexpr (AL) { if (fast_prefetch_ok(T0)) { if (!(PNG_RING_ABL & 1)) prefetch_fast(T0); } else prefetch_line((int)T0); return; }
        if (!PNG_NT_LOADS && full64 && T0 >= 64 && T0 + 8 <= full_iters) {
            if (PNG_RING_ABL & 1) return;
            const uint8_t* pt = pbase + (size_t)T0 * IB;
            #pragma unroll
            for (int k = 0; k < 8; ++k) {
                const PackedU32* q = reinterpret_cast<const PackedU32*>(pt + (int64_t)k * 8 * ((int64_t)a.wb + 1 - IB) + ploff);
                pre[k] = make_uint4(q[0].v, q[1].v, q[2].v, RGBA ? 0u : q[3].v);
            }
            return;
        }
        #pragma unroll
        for (int k = 0; k < 8; ++k) {
            int it = (int)T0 + cslot - (8 * k + crow);                                     // piece of row 8k+crow used in trip T0 + cslot
            const u32 r = (u32)(8 * k + crow) < rows_left ? (u32)(8 * k) : 0u;             // rows past the image re-read a live row of the band (unused):
                                                                                           // never an address outside the stream
            it = it < 0 ? 0 : it;
            // full pieces by index; the ragged last piece (and anything past it, unused) = the last 16 bytes of the row
            const int64_t off = (u32)it < full_iters ? (int64_t)it * IB : (int64_t)a.wb - IB;
#if PNG_NT_LOADS
            typedef u32x4 u32x4_unaligned __attribute__((aligned(1)));
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_unaligned*>(craw + (int64_t)r * (a.wb + 1) + off));
            pre[k] = make_uint4(v.x, v.y, v.z, v.w);
#else
            const PackedU32* q = reinterpret_cast<const PackedU32*>(craw + (int64_t)r * (a.wb + 1) + off);
            pre[k] = make_uint4(q[0].v, q[1].v, q[2].v, RGBA ? 0u : q[3].v);
#endif
        }
    };
    u32x4 dset[PF];
    auto issue_dprev = [&](u32 Tn, u32x4& ds) {
        // lane 0's iteration is Tn: wave-uniform offset (band 0: zeros, see rs_prev)
        ds = __builtin_amdgcn_raw_buffer_load_b128(rs_prev, min(Tn, niter - 1) * 16u, 0, 0);
    };
    // Q: the row above is in memory (sc1 stores leave no copy in any L2), a round trip of a few microseconds: it is fetched a
    // tile (8 pieces = 128 bytes, one piece per lane 0..7, sc1 loads) ahead, parked in 128 bytes of LDS behind the wave's ring at
    // the start of the tile that uses it, and every trip reads its piece from there (one broadcast ds_read).
    u32x4 chunk = { 0u, 0u, 0u, 0u };
    uint8_t* const dch = ring + 64 * PITCH;
    auto issue_chunk = [&](u32 Tbase) {
        const u32 piece = min(Tbase + (u32)(lane & 7), niter - 1);
#if !PNG_Q_NO_CHUNK
        if (lane < 8) chunk = __builtin_amdgcn_raw_buffer_load_b128(rs_prev, piece * 16u, 0, 16);        // sc1: past this CU's L1
#else
        (void)piece;
#endif
    };

    if constexpr (Q) {
        if (band > 0) wait_for_band_above(TT);
        issue_chunk(0);
    } else if (band > 0) wait_for_band_above(PUB + PF);
    // AL: one 16-byte chunk of every row per lane, at the byte offset it has in the row: from the line fetched for this tile
    // (phases up to P) or from the one fetched for the tile before (the phases behind P, kept in registers since)
    auto drop_al = [&](int T0) {
        if constexpr (AL) {
            #pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint8_t* c0; int Sp, P;
                al_row(k, c0, Sp, P);
                const bool fresh = cslot <= P;
                const int cd = T0 - (8 * k + crow) - P + cslot + (fresh ? 8 : 0);
                const int start = cd * 16 - Sp;                                              // row byte of the chunk's first byte
                uint4 v = fresh ? pre[k] : pre_old[k];
                if (start < 0) {                                                              // in front of the row: zeros (what a lane that has not
                    const int n = -start;                                                     // reached its row must find); the row's first chunk: its
                    u32 w[4] = { v.x, v.y, v.z, v.w };                                        // leading bytes (filter byte, the row before)
                    #pragma unroll
                    for (int d = 0; d < 4; ++d) { const int lo = n - 4 * d; w[d] = lo <= 0 ? w[d] : lo >= 4 ? 0u : w[d] & (0xFFFFFFFFu << (8 * lo)); }
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
                const u32 o = (u32)start & 255u;
                uint8_t* dst = co_ring + k * 8 * PITCH + o;
                reinterpret_cast<AnyVec*>(dst)->v = u32x4{ v.x, v.y, v.z, v.w };
                if (o > 240u) reinterpret_cast<AnyVec*>(dst - 256)->v = u32x4{ v.x, v.y, v.z, v.w };
                pre_old[k] = pre[k];
            }
        }
    };
    auto drop_fast = [&](u32 T0) {                  // nothing in front of any row: the ring offset is a constant of the row, bit 7 flipping every other tile
        if constexpr (AL) {
            const u32 flip = (T0 & 8u) << 4;
            #pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool fresh = (freshbits >> k) & 1u;
                const uint4 v = fresh ? pre[k] : pre_old[k];
                const u32 o = (pfo[k] >> 24) ^ flip;
                uint8_t* dst = co_ring + k * 8 * PITCH + o;
                reinterpret_cast<AnyVec*>(dst)->v = u32x4{ v.x, v.y, v.z, v.w };
                if (o > 240u) reinterpret_cast<AnyVec*>(dst - 256)->v = u32x4{ v.x, v.y, v.z, v.w };
                pre_old[k] = pre[k];
            }
        }
    };
    if constexpr (AL) {                             // the ring starts out zero; the line in front of tile 0's goes into the registers first
        #pragma unroll
        for (int i = 0; i < PITCH / 16; ++i) *reinterpret_cast<uint4*>(ring + lane * PITCH + i * 16) = make_uint4(0u, 0u, 0u, 0u);
        prefetch_line(-TT);
        drop_al(-TT);                               // row 0's first chunk belongs to the tile in front of tile 0 (for the other rows: zeros onto zeros)
    }
    prefetch_tile(0);
    if constexpr (!Q) {
        #pragma unroll
        for (int u = 0; u < PF; ++u) issue_dprev((u32)u, dset[u]);
    }

    // Write-back of the group of 8 pieces each row completed with tile T0, in two halves: all eight LDS reads first (one wait instead of
    // eight round trips in a row), then one aligned 128-byte run per row.
    uint4 wbv[8];
    u32 wb_T0 = 0;                                  // Q: the last tile, whose groups go out behind the loop
    // fast form: piece T0 - 8 [crow > 0] + cslot of row crow, as a byte offset from the band's first row: one constant of the lane, + 16 T0
    const u32 wb_lane = (u32)crow * (u32)a.d_pitch + ((u32)cslot - (crow ? 8u : 0u)) * 16u;
    const u32 row63_off = 7u * (u32)a.d_pitch + 56u * 16u;     // (lanes of crow == 7, k == 7: the same piece as an offset into row 63)
    auto wb_read = [&](u32 T0) {
        if (fast_wb_ok(T0)) {
            // every row of the band writes a whole group: piece it = T0 - 8 k - 8 [crow > 0] + cslot of row 8 k + crow.  Its ring slot is
            // (cslot | 8 [crow > 0] ^ 8 [k odd] ^ (T0 & 8)): two addresses per tile
            const u32 flip = (T0 & 8u) << 4;
            const u32 base_w = ((crow ? 128u : 0u) | ((u32)cslot << 4)) ^ flip;
            const uint8_t* rd[2] = { co_ring + base_w, co_ring + (base_w ^ 128u) };
            #pragma unroll
            for (int k = 0; k < 8; ++k) wbv[k] = *reinterpret_cast<const uint4*>(rd[k & 1] + k * 8 * PITCH);
        } else {
            if constexpr (!LN) {
                if (dwp) {                                            // slot 0 once more behind slot 15 (the write-back's own copy: the trips know nothing of it)
                    #pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (cslot == 0) *reinterpret_cast<uint4*>(co_ring + k * 8 * PITCH + RING * 16) = *reinterpret_cast<const uint4*>(co_ring + k * 8 * PITCH);
                }
            }
            #pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ph = wb_phase(k);
                const int it = ((((int)T0 - (8 * k + crow) + ph) & ~7) - ph) + cslot;
                if constexpr (!LN) {
                    const int d = dw_of(k);                            // chunk `it`: from dword 4 - d of piece it - 1 (d > 0), else piece it
                    const u32 off = ((u32)(it - (d ? 1 : 0)) & (RING - 1)) * 16u + (d ? 16u - 4u * (u32)d : 0u);
                    const AnyVec q = *reinterpret_cast<const AnyVec*>(co_ring + k * 8 * PITCH + off);
                    wbv[k] = make_uint4(q.v.x, q.v.y, q.v.z, q.v.w);
                } else
                wbv[k] = *reinterpret_cast<const uint4*>(co_ring + k * 8 * PITCH + ((u32)it & (RING - 1)) * 16);
            }
        }
    };
    auto wb_store = [&](u32 T0) {
        if (fast_wb_ok(T0)) {
            if (PNG_RING_ABL & 2) { asm volatile("" :: "v"(wbv[0].x ^ wbv[1].x ^ wbv[2].x ^ wbv[3].x ^ wbv[4].x ^ wbv[5].x ^ wbv[6].x ^ wbv[7].x)); return; }
            // a piece's place in the output is a lane constant minus 128 k behind a wave-uniform row pointer -- no predicates, no per-row
            // address arithmetic
            const u32 goff0 = wb_lane + T0 * 16u;                                         // (d_pitch * 64 < 2^31: checked by the launcher)
            #pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint4 v = wbv[k];
                u32x4* dst = reinterpret_cast<u32x4*>(dband + (int64_t)(8 * k) * a.d_pitch + (size_t)(goff0 - 128u * (u32)k));
                if (Q && k == 7) {
                    if (crow == 7) __builtin_amdgcn_raw_buffer_store_b128(u32x4{ v.x, v.y, v.z, v.w }, rs_last, goff0 - row63_off, 0, 16);      // row 63: sc1
                    else if (PNG_NT_STORES) __builtin_nontemporal_store(u32x4{ v.x, v.y, v.z, v.w }, dst);
                    else *dst = u32x4{ v.x, v.y, v.z, v.w };
                }
                else if (PNG_NT_STORES) __builtin_nontemporal_store(u32x4{ v.x, v.y, v.z, v.w }, dst);
                else               *dst = u32x4{ v.x, v.y, v.z, v.w };
            }
        } else {
            #pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ph = wb_phase(k);
                const int it = ((((int)T0 - (8 * k + crow) + ph) & ~7) - ph) + cslot;          // piece 8g - ph + cslot, g = floor((T0 - row + ph) / 8)
                const uint4 v = wbv[k];
                if constexpr (!LN) {
                    if (dwp) {
                        const int d = dw_of(k), q0 = 4 * it - d;                             // the chunk's first dword of the row (in front of the row for it = 0)
                        const int lo = q0 < 0 ? -q0 : 0, hi = min(4, 4 * (int)wb_iters - q0);  // its dwords [lo, hi) belong to the row's written-back part
                        if ((u32)(8 * k + crow) < rows_left && hi > lo) {
                            uint8_t* const rowp = cdst + (int64_t)(8 * k) * a.d_pitch;
                            const bool last_row = Q && k == 7 && crow == 7;
                            if (lo == 0 && hi == 4) {
                                u32x4* dst = reinterpret_cast<u32x4*>(rowp + (int64_t)q0 * 4);
                                if (last_row) __builtin_amdgcn_raw_buffer_store_b128(u32x4{ v.x, v.y, v.z, v.w }, rs_last, (u32)q0 * 4u, 0, 16);   // row 63: sc1
                                else if (PNG_NT_STORES) __builtin_nontemporal_store(u32x4{ v.x, v.y, v.z, v.w }, dst);
                                else *dst = u32x4{ v.x, v.y, v.z, v.w };
                            } else {
                                const u32 w4[4] = { v.x, v.y, v.z, v.w };
                                #pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    if (i >= lo && i < hi) {
                                        if (last_row) __builtin_amdgcn_raw_buffer_store_b32(w4[i], rs_last, (u32)(q0 + i) * 4u, 0, 16);
                                        else *reinterpret_cast<u32*>(rowp + (int64_t)(q0 + i) * 4) = w4[i];
                                    }
                            }
                        }
                        continue;
                    }
                }
                if ((u32)(8 * k + crow) < rows_left && it >= 0 && it < (int)wb_iters) {
                    u32x4* dst = reinterpret_cast<u32x4*>(cdst + (int64_t)(8 * k) * a.d_pitch + (int64_t)it * 16);
                    if (Q && k == 7 && crow == 7) __builtin_amdgcn_raw_buffer_store_b128(u32x4{ v.x, v.y, v.z, v.w }, rs_last, (u32)it * 16u, 0, 16);   // row 63: sc1
                    else if (PNG_NT_STORES) __builtin_nontemporal_store(u32x4{ v.x, v.y, v.z, v.w }, dst);
                    else               *dst = u32x4{ v.x, v.y, v.z, v.w };
                }
            }
        }
    };

    // trips must reach iteration niter-1 of lane 63; write-back must reach row 63's last group (tile T0 = 8 g_last + 64)
    const u32 T_end = max(niter + 63, 8 * ((wb_iters - 1) >> 3) + 65) + (wb_lines ? 0u : dwp ? 16u : 8u);       // (line groups end up to 7 pieces later than row groups; chunks of memory one more)
    u32 my_slot = (u32)(-lane) & (RING - 1);                                               // slot of iteration T - lane, kept incrementally
    u32 polled = 0;                                 // Q: the progress word as loaded one tile ago (a round trip to L2 / the fabric that nobody waits for)
    for (u32 T0 = 0; T0 < T_end; T0 += TT) {
        // Q: the order of a tile's vector-memory operations is  stores (the groups the tile before completed), progress word, row above,
        // prefetch -- and nothing else until the next tile's drop.  The counter retires in order: what a drop waits for (its pieces) must be
        // the YOUNGEST operations in flight, or it waits for whatever was issued behind them -- with the stores at the end of the tile
        // (where the groups are complete) every drop waited for memory to acknowledge them, a round trip of a microsecond with nothing
        // between issue and wait.  So the finished groups are read from the ring before the drop overwrites part of them, and stored
        // behind it; by the next drop they have had a whole tile to complete.
        if constexpr (Q) {
            if (T0 > 0) wb_read(T0 - TT);
            if (lane < 8) *reinterpret_cast<u32x4*>(dch + lane * 16) = chunk;      // this tile's pieces of the row above
        }
        // drop this tile's raw pieces (prefetched) into the rings, then start fetching the next tile.  A lane that has not
        // reached its row yet (iteration < 0: the first 64 trips) finds zeros in its slot: with a zero piece, a zero row above (the
        // lane below is not there yet either) and a zero pixel to the left every filter yields zeros, which is what the lane
        // must present to the lane above it and to its own first pixel -- no per-trip masking.
        if constexpr (AL) { if (fast_drop_ok(T0)) drop_fast(T0); else drop_al((int)T0); }
        else
        if (T0 < 64) {
            #pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int it = (int)T0 + cslot - (8 * k + crow);
                const u32 slot = (u32)it & (RING - 1);
                *reinterpret_cast<uint4*>(co_ring + k * 8 * PITCH + slot * 16) = it < 0 ? make_uint4(0u, 0u, 0u, 0u) : pre[k];
            }
        } else {
            const u32 s0 = ((T0 + (u32)cslot - (u32)crow) & (RING - 1)) * 16;              // (x - 8 k) mod 16: bit 3 flips with k
            uint8_t* const wr[2] = { co_ring + s0, co_ring + (s0 ^ 128u) };
            #pragma unroll
            for (int k = 0; k < 8; ++k) *reinterpret_cast<uint4*>(wr[k & 1] + k * 8 * PITCH) = pre[k];
        }
        if constexpr (Q) {
            if (T0 > 0) {
                // the drop has taken its pieces, the youngest loads: everything this wave ever issued has completed (the wait is free),
                // so what the tile before this one stored (the groups up to tile T0 - 16) can be published
                const int done = stored_through((int)T0 - 2 * TT);                   // (line-aligned rows: ((T0 - TT - 63) & ~7))
                if (done > 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    publish(min((u32)done, wb_iters));
                }
                wb_store(T0 - TT);
            }
            if (band > 0) {
                seen = max(seen, (u32)__builtin_amdgcn_readfirstlane((int)polled));
                polled = __hip_atomic_load(prod_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                wait_for_band_above(T0 + 2 * TT);                      // the pieces of the next tile, fetched below
            }
            issue_chunk(T0 + TT);
        }
        prefetch_tile(T0 + TT);
        // some lane meets the partial last piece of its row (iteration full_iters, trip full_iters + lane) in this tile
        const bool rag_tile = (a.wb % IB) != 0 && T0 + TT > full_iters && T0 <= full_iters + 63;

        // The tile's pieces (and, Q, its pieces of the row above) are all in LDS by now: every trip fetches the NEXT trip's
        // operands before it computes, so that no trip begins by waiting for an LDS round trip (with two waves on a SIMD
        // nothing else hides it).
        uint4 rv_next = *reinterpret_cast<const uint4*>(my_ring + my_slot * 16);
        u32x4 dch_next = { 0u, 0u, 0u, 0u };
        if constexpr (Q) dch_next = *reinterpret_cast<const u32x4*>(dch);
        #pragma unroll
        for (int u = 0; u < TT; ++u) {
            const u32 T = T0 + u;
            const int it = (int)T - lane;
            const bool ragged = rag_tile && row_live && it == (int)full_iters;

            uint4* piece = reinterpret_cast<uint4*>(my_ring + my_slot * 16);
            my_slot = (my_slot + 1) & (RING - 1);
            const uint4 rv = rv_next;
            const u32x4 dch_cur = dch_next;
            if (u + 1 < TT) {
                rv_next = *reinterpret_cast<const uint4*>(my_ring + my_slot * 16);
                if constexpr (Q) dch_next = *reinterpret_cast<const u32x4*>(dch + (u + 1) * 16);
            }
            u32 rg[PW] = { rv.x, rv.y, rv.z, rv.w }, bg[PW];
            if (!AL && rag_tile) {      // last, partial piece of a row: the staged piece is the row's LAST 16 bytes (see prefetch_tile);
                if (ragged) {           // keep its top nb bytes, moved down by 16 - nb bytes (zeros come in behind).  No memory op here.
                    const u32 sh = IB - (a.wb - (u32)it * IB), ds = sh >> 2, bs = sh & 3;
                    u32 w[5];
                    #pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        const u32 v0 = i < 4 ? rg[i] : 0u, v1 = i + 1 < 4 ? rg[i + 1] : 0u, v2 = i + 2 < 4 ? rg[i + 2] : 0u, v3 = i + 3 < 4 ? rg[i + 3] : 0u;
                        w[i] = ds == 0 ? v0 : ds == 1 ? v1 : ds == 2 ? v2 : v3;
                    }
                    #pragma unroll
                    for (int i = 0; i < 4; ++i) rg[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], bs);
                }
            }
            u32x4 dcur;
            if constexpr (Q) dcur = dch_cur;
            else             dcur = dset[u % PF];
            if constexpr (RGBA) {       // the chunk of the output row above the band: R,G,B,255 x 4 -> 12 stream bytes
                const u32x4 d = dcur;
                bg[0] = from_lane_below(outp[0], __builtin_amdgcn_perm(d[1], d[0], 0x04020100u));
                bg[1] = from_lane_below(outp[1], __builtin_amdgcn_perm(d[2], d[1], 0x05040201u));
                bg[2] = from_lane_below(outp[2], __builtin_amdgcn_perm(d[3], d[2], 0x06050402u));
                bg[3] = 0;
            } else {
                #pragma unroll
                for (int i = 0; i < PW; ++i) bg[i] = from_lane_below(outp[i], dcur[i]);
            }
            if constexpr (!Q) {
                if (band > 0 && T + PF < niter && ((T + PF) % PUB) == 0) wait_for_band_above(T + PF + PUB);
                issue_dprev(T + PF, dset[u % PF]);
            }

            u32 og[PW], ow[PW];         // de-filtered stream bytes of the piece; what goes into the ring slot (= output bytes)
            if constexpr (RGBA) {
                filter_piece12<PAETH>(rf, rg, bg, outp, bprev, og, ow);
                #pragma unroll
                for (int i = 0; i < PW; ++i) ow[i] |= 0xFF000000u;
            } else {
                filter_piece<FB, PAETH>(rf, rg, bg, outp, bprev, og);
                #pragma unroll
                for (int i = 0; i < PW; ++i) ow[i] = og[i];
            }
            // a lane that has not reached its row yet (it < 0) computes zeros from zeros (see the drop above); past the end of
            // the row (it >= niter) whatever it computes is only seen by lanes that are past the end of theirs as well, and is
            // never written back
            #pragma unroll
            for (int i = 0; i < PW; ++i) { outp[i] = og[i]; bprev[i] = bg[i]; }
            *piece = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            if (rag_tile && a.store_tail_masked) {   // partial piece of an exact-size destination row (wb % 4 == 0 there): up to three
                u32* dst = reinterpret_cast<u32*>(drow + (int64_t)(ragged ? it : 0) * 16);      // dword stores straight to the row
                const u32 nb = ragged ? out_wb - (u32)it * 16 : 0u;
                if constexpr (Q) {      // the band's last row is read by another compute unit: write-through
                    if (nb >= 4) __hip_atomic_store(dst + 0, ow[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (nb >= 8) __hip_atomic_store(dst + 1, ow[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (nb >= 12) __hip_atomic_store(dst + 2, ow[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    if (nb >= 4) dst[0] = ow[0];
                    if (nb >= 8) dst[1] = ow[1];
                    if (nb >= 12) dst[2] = ow[2];
                }
            }
        }

        // write back the group of 8 pieces each row completed with this tile: one aligned 128-byte run per row.  The LDS reads come
        // here (the next drop overwrites part of the group); Q: the stores wait until that drop has happened (see there)
        if constexpr (!Q) {
            wb_read(T0);
            wb_store(T0);
            // publish what lane 63 had written back BEFORE this tile (groups below floor((T0 - 63) / 8)): since then this
            // tile issued 8 prefetch loads and 8 row-above loads, so "at most 16 vector-memory operations outstanding" implies
            // those older stores have been acknowledged (the counter retires in order) -- the prefetches in flight are not drained.
            const int done = stored_through((int)T0 - TT);                           // (line-aligned rows: ((T0 - 63) & ~7))
            if (done > 0) {
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                publish(seq * niter + min((u32)done, wb_iters));
            }
        } else wb_T0 = T0;
    }
    if constexpr (Q) { wb_read(wb_T0); wb_store(wb_T0); }      // the last tile's groups
    // band finished: everything is on its way; drain and publish the whole band
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    publish(seq * niter + niter);
}

template <int FB, int W, int MINW, bool RGBA = false, bool AL = false, bool LN = true>
__global__ __launch_bounds__(W * 64, MINW) void k_png_defilter_ring(DefilterArgs a)
{
    __shared__ u32 prog[W];
    __shared__ __attribute__((aligned(16))) uint8_t tiles[W][64 * ((AL || !LN) ? ROW_PITCH + 32 : ROW_PITCH)];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int img; const uint8_t* raw; uint8_t* D;
    if (!take_segment(a, img, raw, D)) return;
    constexpr u32 IB = RGBA ? 12 : 16;
    const u32 niter = (a.wb + IB - 1) / IB;
    const u32 nbands = (a.rows + 63) / 64;
    if (threadIdx.x < W) prog[threadIdx.x] = 0;
    __syncthreads();
    for (u32 band = wave; band < nbands; band += W) {
        const u32 row = band * 64 + lane;
        const bool row_live = row < a.rows;
        u32 f = row_live ? raw[(int64_t)row * (a.wb + 1)] : 0;
        if (f > 4) { if (a.status) atomicOr(a.status + img, 1u); f = 0; }
        if (__any(f == 4)) defilter_band_ring<FB, W, true,  RGBA, false, AL && PNG_AL_PAETH, LN>(a, raw, D, prog, tiles[wave], band, wave, lane, niter, f, row_live);
        else               defilter_band_ring<FB, W, false, RGBA, false, AL, LN>(a, raw, D, prog, tiles[wave], band, wave, lane, niter, f, row_live);
    }
}

// The same bands, handed out by a queue: a persistent grid (one workgroup of W waves per compute unit) in which every WAVE
// draws the next (image, band) unit from one device-wide counter.  An image no longer belongs to a workgroup, so a batch
// occupies every wave slot of the chip until the last units (341 x 1080p images = 5 797 bands on 2 048 slots, instead of two
// rounds of workgroups that each run 17 bands on 8 waves in three rounds), and a single image's bands spread over as many
// compute units as it has bands, whatever its filters.
// Order of the units: the batch is cut into groups of `group` images, a group's units run band-major (band 0 of every image
// of the group, then band 1, ...).  Unit (i, b) therefore follows (i, b - 1) -- whoever holds it is already running, and the
// oldest unit in flight never waits, so the queue cannot deadlock whatever the number of resident workgroups -- and by the time
// a wave draws (i, b) the band above it has usually had the ~80 trips of head start it needs (lane 63 of a band runs 63 pieces
// behind its lane 0), so waves rarely sit waiting.
template <int FB, int W, int MINW, bool RGBA = false, bool AL = false, bool LN = true>
__global__ __launch_bounds__(W * 64, MINW) void k_png_defilter_queue(DefilterArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t tiles[W][64 * ((AL || !LN) ? ROW_PITCH + 32 : ROW_PITCH) + 128];      // a wave's ring + 128 bytes of the row above its band
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    constexpr u32 IB = RGBA ? 12 : 16;
    const u32 niter = (a.wb + IB - 1) / IB;
    const u32 total = a.count * a.nbands, per_group = a.group * a.nbands;
    for (;;) {
        u32 u = 0;
        if (lane == 0) u = __hip_atomic_fetch_add(a.qstate, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u = (u32)__builtin_amdgcn_readfirstlane((int)u);
        if (u >= total) break;
        const u32 g = u / per_group, r = u - g * per_group;
        const u32 gl = min(a.group, a.count - g * a.group);        // images in this group (the last one may be short)
        const u32 band = r / gl;
        const int img = (int)(g * a.group + (r - band * gl));
        const uint8_t* raw = image_raw(a, img);
        uint8_t* D = image_rows(a, img);
        u32* prog = a.qstate + QSTATE_HDR + (size_t)img * a.nbands;
        const u32 row = band * 64 + lane;
        const bool row_live = row < a.rows;
        u32 f = row_live ? raw[(int64_t)row * (a.wb + 1)] : 0;
        if (f > 4) { if (a.status) atomicOr(a.status + img, 1u); f = 0; }
        u32* st = a.status ? a.status + img : nullptr;
        if (__any(f == 4)) defilter_band_ring<FB, W, true,  RGBA, true, AL && PNG_AL_PAETH, LN>(a, raw, D, prog, tiles[wave], band, wave, lane, niter, f, row_live, st);
        else               defilter_band_ring<FB, W, false, RGBA, true, AL, LN>(a, raw, D, prog, tiles[wave], band, wave, lane, niter, f, row_live, st);
    }
}

// =====================================================================================================
// The rolling form of the queue launch: the same bands, the same lane = row trips, the same DPP hand-over -- but a row's ring is
// EIGHT pieces (128 bytes), not sixteen, and the cooperative transfers are spread over the trips instead of bunched at tile ends.
// Why: a wave issues a dependent vector instruction every 9 clocks and an independent one every 7 (tools/microbench/valu_latency.hip),
// a SIMD can take one every 4 -- with the 18 KB rings of defilter_band_ring a compute unit holds 8 waves, two per SIMD, and the Paeth
// bands (200 instructions per 16-byte piece) ran with the vector ALUs 70 % busy.  9 KB per wave is 16 waves per compute unit.
//   * piece p of row r lives in slot (p + r) mod 8: in trip T EVERY lane works on slot T mod 8 of its own row (an immediate offset);
//     rows are 144 bytes apart (nine slots), so eight consecutive lanes reading the same slot number hit eight different bank groups.
//   * rows r = T + 1 (mod 8) finish a group of eight pieces with trip T and start the next with trip T + 1.  Those are eight rows,
//     8 q + rho: lane (q, c) = (lane / 8, lane mod 8) takes slot c of row 8 q + rho -- reads the finished piece out of it (to be
//     stored: 128 aligned bytes per row), puts the raw piece that belongs there next into it (fetched eight trips ago), and fetches
//     the one after that.  One ds_read, one ds_write, one store and one load per trip, always 8 lanes to a row's 128 bytes.
//   * the hand-off between bands (progress words, the band's last row written through, the row above in 128 bytes of LDS) is
//     that of the queue form of defilter_band_ring.
#ifndef PNG_ROLL_WPS             // waves per SIMD the rolling form is compiled for (tuning knob): 4 -> 128 registers, 3 -> 168
#define PNG_ROLL_WPS 3
#endif
#ifndef PNG_ROLL_ABL             // ablations (wrong pixels; measurements only): 1 = the fast tiles fetch nothing, 2 = store nothing
#define PNG_ROLL_ABL 0
#endif
constexpr int RPITCH = 144;
template <int FB, bool PAETH>
__device__ __forceinline__ void defilter_band_roll(const DefilterArgs& a, const uint8_t* raw, uint8_t* D, u32* prog, uint8_t* ring, u32 band,
                                                   int lane, u32 niter, u32 f, bool row_live, u32* status)
{
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 u32x4_unaligned __attribute__((aligned(1)));
    constexpr int PW = 4;
    const RowFilter rf = row_filter(f);
    const u32 row = band * 64 + lane;
    const u32 rows_left = a.rows - band * 64;
    const bool full64 = rows_left >= 64;
    const u32 full_iters = a.wb / 16;
    const u32 wb_iters = a.store_tail_masked ? full_iters : niter;
    const int q = lane >> 3, cslot = lane & 7;
    uint8_t* drow = D + (int64_t)(row_live ? row : 0) * a.d_pitch;
    const uint8_t* dprev = band > 0 ? D + (int64_t)(band * 64 - 1) * a.d_pitch : D;
    u32* const my_flag = prog + band;
    u32* const prod_flag = prog + (band > 0 ? band - 1 : 0);
    uint8_t* const my_ring = ring + lane * RPITCH;                       // + 16 (T mod 8)
    uint8_t* const co_ring = ring + q * 8 * RPITCH + cslot * 16;         // + rho RPITCH
    uint8_t* const dch = ring + 64 * RPITCH;

    // wave-uniform bases (said so to the compiler half by half, so that loads and stores take a scalar base and a 32-bit lane offset)
    auto uniform_ptr = [](const uint8_t* base, int64_t off) {
        return base + (int64_t)(((uint64_t)(u32)__builtin_amdgcn_readfirstlane((int)((uint64_t)off >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)(uint64_t)off));
    };
    const uint8_t* const rband = uniform_ptr(raw, (int64_t)band * 64 * ((int64_t)a.wb + 1) + 1);      // first byte of the band's first row
    uint8_t* const dband = const_cast<uint8_t*>(uniform_ptr(D, (int64_t)band * 64 * a.d_pitch));
    const u32 lq_raw = (u32)q * (8u * (a.wb + 1u) - 128u);               // fast forms: row 8 q + rho, piece ... - 8 q: the lane's share of the offset
    const u32 lq_out = (u32)q * (8u * (u32)a.d_pitch - 128u);
    const u32 cslot16 = (u32)cslot * 16u;

    const auto rs_prev = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(dprev), 0, band > 0 ? (int)(niter * 16) : 0, 0x00020000);
    const auto rs_last = __builtin_amdgcn_make_buffer_rsrc(D + (int64_t)(band * 64 + 63) * a.d_pitch, 0, (int)(niter * 16), 0x00020000);
    // generic forms store through a descriptor of the band's live rows with the offsets of lanes that have nothing to store out of its
    // range (the hardware drops them): one store instruction per trip whatever the data, like the fast forms -- see the vmcnt note below
    const auto rs_band = __builtin_amdgcn_make_buffer_rsrc(dband, 0, (int)(min(rows_left, 64u) * (u32)a.d_pitch), 0x00020000);
    u32 seen = 0;
    auto wait_for_band_above = [&](u32 upto) {
        const u32 need = min(niter, upto);
        if (seen >= need) return;
        const uint64_t t0 = wall_clock64();
        for (;;) {
            seen = (u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(prod_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (seen >= need) break;
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > 400000000ull) {                     // 4 s of the 100 MHz clock: give up, say so, never hang
                if (status && lane == 0) atomicOr(status, STATUS_HANDOFF_TIMEOUT);
                seen = 0xFFFFFFFFu;
                break;
            }
        }
    };
    auto publish = [&](u32 value) { if (lane == 63) __hip_atomic_store(my_flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    u32x4 chunk = { 0u, 0u, 0u, 0u };
    auto issue_chunk = [&](u32 Tbase) {
        const u32 piece = min(Tbase + (u32)(lane & 7), niter - 1);
        if (lane < 8) chunk = __builtin_amdgcn_raw_buffer_load_b128(rs_prev, piece * 16u, 0, 16);        // sc1: past this CU's L1
    };

    // the raw piece that the boundary of trip Tb (class rho = (Tb + 1) mod 8) will drop: piece Tb + 1 - (8 q + rho) + e of row 8 q + rho
    // (generic forms: the first and last tiles of a band.  Their per-lane addresses are recomputed where they are used -- `opaque` keeps the
    // compiler from hoisting sixteen of them per class out of the loop into registers the fast tiles need)
    auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
    auto fetch_generic = [&](int Tb, int rho) -> u32x4 {
        const u32 rc = (u32)(8 * opaque(q) + rho);
        const int e = (cslot - rho) & 7;
        int p = Tb + 1 - (int)rc + e;
        p = p < 0 ? 0 : p;
        const int64_t off = (u32)p < full_iters ? (int64_t)p * 16 : (int64_t)a.wb - 16;      // the ragged last piece (and anything past it, unused) = the row's last 16 bytes
        const u32 rl = rc < rows_left ? rc : 0u;                                                // rows past the image re-read a live row of the band (unused)
        return *reinterpret_cast<const u32x4_unaligned*>(rband + (int64_t)rl * ((int64_t)a.wb + 1) + off);
    };

    #pragma unroll
    for (int i = 0; i < RPITCH / 16; ++i) *reinterpret_cast<uint4*>(my_ring + i * 16) = make_uint4(0u, 0u, 0u, 0u);   // a lane that has not reached its row finds zeros
    u32x4 pre[8];
    #pragma unroll
    for (int rho = 0; rho < 8; ++rho) pre[rho] = fetch_generic(rho - 1, rho);       // what the first boundary of every class drops
    if (band > 0) wait_for_band_above(8);
    issue_chunk(0);
    // The first piece of a group never comes out of the ring: slot c = rho of row 8 q + rho is dropped by lane 8 q + rho -- the lane
    // that computes that row.  It keeps the piece (`keep`) and takes it from there in the next trip, so every lane can ask the ring for
    // its next piece at the START of a trip (a whole trip before it is needed) although the rows of one class are dropped at its end.
    u32x4 keep;
    {   // the boundary "of trip -1": rows 0 (mod 8) start with trip 0 -- only row 0 itself has pieces to drop, the others keep their zeros
        const int e = cslot;
        const int p = -(8 * q) + e;
        keep = p < 0 ? u32x4{ 0u, 0u, 0u, 0u } : pre[0];
        *reinterpret_cast<u32x4*>(co_ring) = keep;
        pre[0] = fetch_generic(7, 0);
    }
    // (Eight stores that change nothing -- the band's progress word is 0 and stays 0.  The compiler counts, per load, the operations
    // issued behind it and waits with vmcnt(that many); where paths meet it takes the smaller count.  Entering the loop from here the
    // fetches above would have 0-8 operations behind them, coming round the loop 1-15: with these the loop's own count stands, and a
    // drop waits for the piece fetched eight trips ago, not for the loads and stores of the last four trips.)
    #pragma unroll
    for (int i = 0; i < 8; ++i) publish(0u);

    u32 outp[PW] = { 0u, 0u, 0u, 0u }, bprev[PW] = { 0u, 0u, 0u, 0u };
    const u32 g_last = (wb_iters ? wb_iters - 1 : 0u) >> 3;
    const u32 T_end = max(niter + 63, 63 + 8 * g_last + 8);
    u32 polled = 0;
    bool settled = false;                           // the two tiles before this one took the fast forms (one store and one load per trip, whatever the data)
    bool was_fast = false;
    uint4 rv_next = *reinterpret_cast<const uint4*>(my_ring);
    for (u32 T0 = 0; T0 < T_end; T0 += 8) {
        // head of a tile: what row 63 stored two tiles ago has arrived (at most the 16 operations of the last tile can be outstanding
        // behind it when both tiles ran the fast forms; otherwise wait for everything) -- publish it; then this tile's pieces of the row above
        {
            const int done = (int)T0 - 72;
            if (done > 0) {
                if (settled) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else         asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                publish(min((u32)done, wb_iters));
            }
        }
        if (band > 0) {
            seen = max(seen, (u32)__builtin_amdgcn_readfirstlane((int)polled));
            polled = __hip_atomic_load(prod_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wait_for_band_above(T0 + 16);                                  // the pieces of the next tile, fetched below
        }
        if (lane < 8) *reinterpret_cast<u32x4*>(dch + lane * 16) = chunk;  // this tile's pieces of the row above
        issue_chunk(T0 + 8);
        const bool fast = full64 && T0 >= 72 && T0 + 24 <= full_iters;
        settled = was_fast && fast;
        was_fast = fast;
        const bool rag_tile = (a.wb % 16) != 0 && T0 + 8 > full_iters && T0 <= full_iters + 63;
        u32x4 dch_next = *reinterpret_cast<const u32x4*>(dch);

        auto trip = [&](auto fast_c, auto u_c) {
            constexpr bool FAST = decltype(fast_c)::value;
            constexpr int u = decltype(u_c)::value, rho = (u + 1) & 7;
            const u32 T = T0 + u;
            const int it = (int)T - lane;
            const bool ragged = rag_tile && row_live && it == (int)full_iters;
            const bool first_of_group = (lane & 7) == u;                        // this lane's row starts a group with this trip: its piece is in `keep`
            const uint4 rv = make_uint4(first_of_group ? keep[0] : rv_next.x, first_of_group ? keep[1] : rv_next.y,
                                        first_of_group ? keep[2] : rv_next.z, first_of_group ? keep[3] : rv_next.w);
            const u32x4 dcur = dch_next;
            rv_next = *reinterpret_cast<const uint4*>(my_ring + ((u + 1) & 7) * 16);       // (for rows dropped at the end of this trip: stale, replaced by `keep`)
            if (u + 1 < 8) dch_next = *reinterpret_cast<const u32x4*>(dch + (u + 1) * 16);
            u32 rg[PW] = { rv.x, rv.y, rv.z, rv.w }, bg[PW];
            if (!FAST && rag_tile) {    // last, partial piece of a row: the staged piece is the row's LAST 16 bytes (see fetch_generic);
                if (ragged) {           // keep its top nb bytes, moved down by 16 - nb bytes (zeros come in behind).  No memory op here.
                    const u32 sh = 16 - (a.wb - (u32)it * 16), ds = sh >> 2, bs = sh & 3;
                    u32 w[5];
                    #pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        const u32 v0 = i < 4 ? rg[i] : 0u, v1 = i + 1 < 4 ? rg[i + 1] : 0u, v2 = i + 2 < 4 ? rg[i + 2] : 0u, v3 = i + 3 < 4 ? rg[i + 3] : 0u;
                        w[i] = ds == 0 ? v0 : ds == 1 ? v1 : ds == 2 ? v2 : v3;
                    }
                    #pragma unroll
                    for (int i = 0; i < 4; ++i) rg[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], bs);
                }
            }
            #pragma unroll
            for (int i = 0; i < PW; ++i) bg[i] = from_lane_below(outp[i], dcur[i]);
            u32 og[PW];
            filter_piece<FB, PAETH>(rf, rg, bg, outp, bprev, og);
            #pragma unroll
            for (int i = 0; i < PW; ++i) { outp[i] = og[i]; bprev[i] = bg[i]; }
            *reinterpret_cast<uint4*>(my_ring + u * 16) = make_uint4(og[0], og[1], og[2], og[3]);
            if (!FAST && rag_tile && a.store_tail_masked) {   // partial piece of an exact-size destination row (wb % 4 == 0 there): up to three
                u32* dst = reinterpret_cast<u32*>(drow + (int64_t)(ragged ? it : 0) * 16);      // dword stores straight to the row, written through
                const u32 nb = ragged ? a.wb - (u32)it * 16 : 0u;                               // (the band's last row is read by another compute unit)
                if (nb >= 4) __hip_atomic_store(dst + 0, og[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (nb >= 8) __hip_atomic_store(dst + 1, og[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (nb >= 12) __hip_atomic_store(dst + 2, og[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // the boundary: rows 8 q + rho have just finished the group that ends with piece pn - 1, pn = T + 1 - (8 q + rho)
            uint8_t* const slot = co_ring + rho * RPITCH;
            const uint4 wbv = *reinterpret_cast<const uint4*>(slot);
            const u32 e16 = (cslot16 - 16u * (u32)rho) & 0x70u;               // 16 e, e = (c - rho) mod 8: slot c holds pieces pn - 8 + e, then pn + e
            if constexpr (FAST) keep = pre[rho];
            else {
                const int p_dr = (int)T + 1 - (8 * opaque(q) + rho) + (int)(e16 >> 4);
                keep = p_dr < 0 ? u32x4{ 0u, 0u, 0u, 0u } : pre[rho];
            }
            *reinterpret_cast<u32x4*>(slot) = keep;
            const u32x4 wv = { wbv.x, wbv.y, wbv.z, wbv.w };
            if constexpr (FAST) {
                // piece pn - 8 + e of row 8 q + rho: 16 (T - 7 - rho) + rho d_pitch from the band's first row (uniform), + the lane's q and e
                uint8_t* const sb = dband + ((int64_t)rho * a.d_pitch + 16 * ((int64_t)T - 7 - rho));
                const u32 vo = lq_out + e16;
#if !(PNG_ROLL_ABL & 2)
                if (rho == 7) {
                    if (q == 7) __builtin_amdgcn_raw_buffer_store_b128(wv, rs_last, 16u * (T - 7u - 7u - 56u) + e16, 0, 16);      // row 63: sc1
                    else __builtin_nontemporal_store(wv, reinterpret_cast<u32x4*>(sb + (size_t)vo));
                } else __builtin_nontemporal_store(wv, reinterpret_cast<u32x4*>(sb + (size_t)vo));
#else
                asm volatile("" :: "v"(wv), "v"(vo), "s"(sb));
#endif
                const uint8_t* const lb = rband + ((int64_t)rho * ((int64_t)a.wb + 1) + 16 * ((int64_t)T + 9 - rho));
#if !(PNG_ROLL_ABL & 1)
                pre[rho] = *reinterpret_cast<const u32x4_unaligned*>(lb + (size_t)(lq_raw + e16));
#else
                asm volatile("" :: "v"(lq_raw + e16), "s"(lb));
#endif
            } else {
                const u32 rc = (u32)(8 * opaque(q) + rho);
                const int p_wb = (int)T + 1 - (int)rc - 8 + (int)(e16 >> 4);
                const bool in_row = p_wb >= 0 && p_wb < (int)wb_iters;
                const u32 nowhere = 0x80000000u;                                     // (rows past the image are past the descriptor's range by themselves)
                __builtin_amdgcn_raw_buffer_store_b128(wv, rs_band, in_row && rc != 63 ? rc * (u32)a.d_pitch + (u32)p_wb * 16u : nowhere, 0, 0);
                if (rho == 7) __builtin_amdgcn_raw_buffer_store_b128(wv, rs_last, in_row && rc == 63 && rows_left >= 64 ? (u32)p_wb * 16u : nowhere, 0, 16);   // row 63: sc1
                pre[rho] = fetch_generic((int)T + 8, rho);
            }
        };
        auto tile = [&](auto fast_c) {
            trip(fast_c, std::integral_constant<int, 0>{}); trip(fast_c, std::integral_constant<int, 1>{});
            trip(fast_c, std::integral_constant<int, 2>{}); trip(fast_c, std::integral_constant<int, 3>{});
            trip(fast_c, std::integral_constant<int, 4>{}); trip(fast_c, std::integral_constant<int, 5>{});
            trip(fast_c, std::integral_constant<int, 6>{}); trip(fast_c, std::integral_constant<int, 7>{});
        };
        if (fast) tile(std::true_type{}); else tile(std::false_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    publish(niter);
}

template <int FB, int W, int MINW>
__global__ __launch_bounds__(W * 64, MINW) void k_png_defilter_rollq(DefilterArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t tiles[W][64 * RPITCH + 128];      // a wave's rings + 128 bytes of the row above its band
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const u32 niter = (a.wb + 15) / 16;
    const u32 total = a.count * a.nbands, per_group = a.group * a.nbands;
    for (;;) {
        u32 u = 0;
        if (lane == 0) u = __hip_atomic_fetch_add(a.qstate, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u = (u32)__builtin_amdgcn_readfirstlane((int)u);
        if (u >= total) break;
        const u32 g = u / per_group, r = u - g * per_group;
        const u32 gl = min(a.group, a.count - g * a.group);
        const u32 band = r / gl;
        const int img = (int)(g * a.group + (r - band * gl));
        const uint8_t* raw = image_raw(a, img);
        uint8_t* D = image_rows(a, img);
        u32* prog = a.qstate + QSTATE_HDR + (size_t)img * a.nbands;
        const u32 row = band * 64 + lane;
        const bool row_live = row < a.rows;
        u32 f = row_live ? raw[(int64_t)row * (a.wb + 1)] : 0;
        if (f > 4) { if (a.status) atomicOr(a.status + img, 1u); f = 0; }
        u32* st = a.status ? a.status + img : nullptr;
        if (__any(f == 4)) defilter_band_roll<FB, true >(a, raw, D, prog, tiles[wave], band, lane, niter, f, row_live, st);
        else               defilter_band_roll<FB, false>(a, raw, D, prog, tiles[wave], band, lane, niter, f, row_live, st);
    }
}

// ---- stage B: expand de-filtered rows into the output image (stbdec.d:1467-1480, 1504-1546, 1552-1632) ----
struct ExpandArgs {
    const uint8_t* D; int64_t d_stride; int64_t d_pitch;
    uint8_t* out; int64_t out_stride;
    u32 x, y; int img_n, out_n, depth, color;
    const int64_t* out_offs;                     // optional (device): byte offset of image i's pixels instead of i * out_stride
};
__global__ __launch_bounds__(256) void k_png_expand(ExpandArgs a)
{
    const int img = blockIdx.y;
    const uint8_t* D = a.D + (int64_t)img * a.d_stride;
    uint8_t* out = a.out + (a.out_offs ? a.out_offs[img] : (int64_t)img * a.out_stride);
    const int64_t npx = (int64_t)a.x * a.y;
    const int bytes = a.depth == 16 ? 2 : 1;
    const u32 scale = (a.color == 0) ? (a.depth == 1 ? 0xFFu : a.depth == 2 ? 0x55u : a.depth == 4 ? 0x11u : 1u) : 1u;   // stbi__depth_scale_table :1403
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npx; p += (int64_t)gridDim.x * 256) {
        const u32 row = (u32)(p / a.x), col = (u32)(p - (int64_t)row * a.x);
        const uint8_t* d = D + (int64_t)row * a.d_pitch;
        uint8_t* o = out + p * (a.out_n * bytes);
        for (int ch = 0; ch < a.out_n; ++ch) {
            if (ch < a.img_n) {
                const u32 s = col * a.img_n + ch;                 // sample index in the row
                if (a.depth == 8) o[ch] = d[s];
                else if (a.depth == 16) { o[2 * ch] = d[2 * s + 1]; o[2 * ch + 1] = d[2 * s]; }      // (hi << 8) | lo, little-endian store
                else {
                    const u32 per = 8 / a.depth, byte = d[s / per];
                    const u32 v = (byte >> (8 - a.depth - (s % per) * a.depth)) & ((1u << a.depth) - 1);
                    o[ch] = (uint8_t)(scale * v);
                }
            } else {                                              // inserted alpha = 255 / 65535
                if (bytes == 2) { o[2 * ch] = 255; o[2 * ch + 1] = 255; } else o[ch] = 255;
            }
        }
    }
}


// ---- post passes (one thread per pixel) --------------------------------------------------------
// stbi__compute_transparency / 16 (stbdec.d:1682-1730): colour-key -> alpha
template <typename T>
__global__ __launch_bounds__(256) void k_png_transparency(T* p, int64_t npx, int out_n, T t0, T t1, T t2, T maxv)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (int64_t)gridDim.x * 256) {
        T* q = p + i * out_n;
        if (out_n == 2) q[1] = (q[0] == t0) ? 0 : maxv;
        else if (q[0] == t0 && q[1] == t1 && q[2] == t2) q[3] = 0;
    }
}
// stbi__expand_png_palette (:1732-1765); palette = 256 x RGBA in HBM
__global__ __launch_bounds__(256) void k_png_palette(const uint8_t* idx, uint8_t* out, int64_t npx, int pal_n, const uint8_t* palette)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (int64_t)gridDim.x * 256) {
        const uint8_t* e = palette + (int)idx[i] * 4;
        uint8_t* o = out + i * pal_n;
        o[0] = e[0]; o[1] = e[1]; o[2] = e[2];
        if (pal_n == 4) o[3] = e[3];
    }
}
// stbi__convert_format / 16 (:916-1199); luma = (77 r + 150 g + 29 b) >> 8 (:911-914)
template <typename T>
__global__ __launch_bounds__(256) void k_png_convert_format(const T* src, T* dst, int64_t npx, int img_n, int req, T maxv)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (int64_t)gridDim.x * 256) {
        const T* s = src + i * img_n; T* d = dst + i * req;
        const T y = img_n >= 3 ? (T)(((int)s[0] * 77 + (int)s[1] * 150 + 29 * (int)s[2]) >> 8) : s[0];
        switch (img_n * 8 + req) {
        case 1*8+2: d[0] = s[0]; d[1] = maxv; break;
        case 1*8+3: d[0] = d[1] = d[2] = s[0]; break;
        case 1*8+4: d[0] = d[1] = d[2] = s[0]; d[3] = maxv; break;
        case 2*8+1: d[0] = s[0]; break;
        case 2*8+3: d[0] = d[1] = d[2] = s[0]; break;
        case 2*8+4: d[0] = d[1] = d[2] = s[0]; d[3] = s[1]; break;
        case 3*8+4: d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = maxv; break;
        case 3*8+1: d[0] = y; break;
        case 3*8+2: d[0] = y; d[1] = maxv; break;
        case 4*8+1: d[0] = y; break;
        case 4*8+2: d[0] = y; d[1] = s[3]; break;
        case 4*8+3: d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; break;
        default: break;
        }
    }
}
// stbi__convert_16_to_8 (:635-649, >> 8) and stbi__convert_8_to_16 (:651-666, * 257)
__global__ __launch_bounds__(256) void k_png_16_to_8(const uint16_t* src, uint8_t* dst, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = (uint8_t)((src[i] >> 8) & 0xFF);
}
__global__ __launch_bounds__(256) void k_png_8_to_16(const uint8_t* src, uint16_t* dst, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = (uint16_t)((src[i] << 8) + src[i]);
}
// Adam7 scatter of one pass into the final image (stbi__create_png_image :1664-1671)
__global__ __launch_bounds__(256) void k_png_adam7_scatter(const uint8_t* pass, uint8_t* final_, u32 px, u32 py, u32 img_x, int out_bytes,
                                                           int xorig, int yorig, int xspc, int yspc)
{
    const int64_t n = (int64_t)px * py;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const u32 j = (u32)(i / px), ii = (u32)(i - (int64_t)j * px);
        const uint8_t* s = pass + i * out_bytes;
        uint8_t* d = final_ + ((int64_t)(j * yspc + yorig) * img_x + (ii * xspc + xorig)) * out_bytes;
        for (int b = 0; b < out_bytes; ++b) d[b] = s[b];
    }
}

// the grid covers the items once (a persistent grid-stride grid streams ~15 % slower on this part: tools/copy_probe.hip)
inline int blocks_for(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b > 0x7FFFFFFFLL ? 0x7FFFFFFFLL : (b < 1 ? 1 : b)); }


} // namespace

// Vector form of stage B for whole-byte samples: a thread expands 4 pixels (whole dwords on both sides for every channel
// count), i.e. inserts alpha = 255 / 65535 (stbdec.d:1467-1480, :1504-1546) and swaps 16-bit samples to host order
// (:1617-1632).  De-filtered rows are 16-byte aligned (scratch pitch); the tight output rows may start anywhere, which
// gfx950's unaligned dword stores absorb.
template <int IN_N, int OUT_N, int BYTES>
__global__ __launch_bounds__(256) void k_png_expand_vec(ExpandArgs a)
{
    constexpr int IB = 4 * IN_N * BYTES, OB = 4 * OUT_N * BYTES;
    const u32 g = blockIdx.x * 256 + threadIdx.x;
    if (g >= (a.x + 3) / 4) return;
    const uint8_t* D = a.D + (int64_t)blockIdx.z * a.d_stride + (int64_t)g * IB;
    uint8_t* out = a.out + (a.out_offs ? a.out_offs[blockIdx.z] : (int64_t)blockIdx.z * a.out_stride) + (int64_t)g * OB;
    const u32 npx = min(4u, a.x - 4 * g);
    for (u32 row = blockIdx.y; row < a.y; row += gridDim.y) {
        const u32* src = reinterpret_cast<const u32*>(D + (int64_t)row * a.d_pitch);
        u32 in[IB / 4], ow[OB / 4];
        #pragma unroll
        for (int i = 0; i < IB / 4; ++i) in[i] = src[i];
        #pragma unroll
        for (int i = 0; i < OB / 4; ++i) ow[i] = 0;
        #pragma unroll
        for (int px = 0; px < 4; ++px)
            #pragma unroll
            for (int ch = 0; ch < OUT_N; ++ch)
                #pragma unroll
                for (int b = 0; b < BYTES; ++b) {
                    const int ob = (px * OUT_N + ch) * BYTES + b;
                    u32 v = 0xFFu;                                          // inserted alpha
                    if (ch < IN_N) { const int ib = (px * IN_N + ch) * BYTES + (BYTES == 2 ? 1 - b : b); v = (in[ib >> 2] >> ((ib & 3) * 8)) & 0xFFu; }
                    ow[ob >> 2] |= v << ((ob & 3) * 8);
                }
        uint8_t* dst = out + (int64_t)row * a.x * (OUT_N * BYTES);
        if (npx == 4) {
            #pragma unroll
            for (int i = 0; i < OB / 4; ++i) reinterpret_cast<PackedU32*>(dst)[i].v = ow[i];
        } else {
            for (u32 i = 0; i < npx * OUT_N * BYTES; ++i) dst[i] = (uint8_t)(ow[i >> 2] >> ((i & 3) * 8));
        }
    }
}

int png_defilter_launch(const uint8_t* raw, int64_t raw_stride, uint32_t raw_len,
                         uint8_t* out, int64_t out_stride,
                         uint32_t x, uint32_t y, int img_n, int out_n, int depth, int color,
                         int count, uint32_t* status, hipStream_t stream,
                         const int64_t* raw_offs, const int64_t* out_offs, bool offs_dword_aligned, bool offs_line_aligned)
{
    // validation as in stbi__create_png_image_raw (stbdec.d:1419-1430, 1441-1442) and parse_png_file (:1890-1906)
    if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16)
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "png_defilter: depth must be 1/2/4/8/16");
    if (img_n < 1 || img_n > 4 || !(out_n == img_n || out_n == img_n + 1) || out_n > 4)
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "png_defilter: bad channel counts %d -> %d", img_n, out_n);
    if (x == 0 || y == 0 || x > (1u << 24) || y > (1u << 24) || (1u << 30) / x / (uint32_t)img_n < y)
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "png_defilter: bad image size");
    if (count < 0) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "png_defilter: negative count");
    if (count == 0) return GAMUT_HIP_OK;
    if (!raw || !out) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "png_defilter: null pointer");
    const uint32_t wb = ((uint32_t)img_n * x * (uint32_t)depth + 7) >> 3;
    if ((uint64_t)raw_len < ((uint64_t)wb + 1) * y)
        return set_error(GAMUT_HIP_ERR_DECODE, "png_defilter: not enough pixels (raw_len %u < %llu)", raw_len, (unsigned long long)((uint64_t)wb + 1) * y);
    if (depth < 8 && wb > x) return set_error(GAMUT_HIP_ERR_DECODE, "png_defilter: invalid width");

    const int bytes = depth == 16 ? 2 : 1;
    const int FB = depth < 8 ? 1 : img_n * bytes;
    // raw_offs / out_offs (device arrays, both or neither): images at arbitrary offsets instead of a constant stride
    const bool out_dwords = ((uintptr_t)out % 4) == 0 && (out_offs ? offs_dword_aligned : (count == 1 || out_stride % 4 == 0));
    // 8-bit RGB -> RGBA8 in one pass: the ring kernel walks the row in 4-pixel pieces and writes the expanded pixels itself
    const bool rgba_fused = depth == 8 && img_n == 3 && out_n == 4 && wb >= 16 && out_dwords;
    const bool fused = rgba_fused || (depth == 8 && out_n == img_n && (wb % 4) == 0 && out_dwords);

    DefilterArgs a{};
    a.raw = raw; a.raw_stride = raw_stride; a.rows = y; a.wb = wb; a.status = status; a.raw_offs = raw_offs; a.d_offs = nullptr;
    // The non-fused formats de-filter into a scratch that lives until k_png_expand has read it.  These entry points are
    // asynchronous and may be called from one thread on several streams, so every (thread, stream) pair owns its scratch:
    // launches on one stream are ordered, launches on different streams never share a buffer.  Growing a scratch waits for
    // ITS stream only (the buffer may still be in use there) -- never for the device.
    struct StreamScratch { int device; hipStream_t stream; int kind; void* p; size_t cap; };      // kind 0: de-filtered rows, 1: queue state
    static thread_local std::vector<StreamScratch> scratches;
    auto scratch_get = [&](size_t n, int kind = 0) -> void* {
        StreamScratch* e = nullptr;
        const int device = current_device();                // (the null stream is one handle for every device)
        for (StreamScratch& c : scratches) if (c.stream == stream && c.kind == kind && c.device == device) { e = &c; break; }
        if (!e) { scratches.push_back(StreamScratch{ device, stream, kind, nullptr, 0 }); e = &scratches.back(); }
        if (n > e->cap) {
            if (e->p) { (void)hipStreamSynchronize(stream); (void)hipFree(e->p); e->p = nullptr; e->cap = 0; }
            const size_t want = n + n / 4 + 4096;
            if (hipMalloc(&e->p, want) != hipSuccess) { (void)hipGetLastError(); e->p = nullptr; return nullptr; }
            e->cap = want;
        }
        return e->p;
    };
    if (fused) { a.D = out; a.d_stride = out_stride; a.d_offs = out_offs; a.d_pitch = rgba_fused ? (int64_t)x * 4 : wb; a.store_tail_masked = 1; }
    else {
        const int64_t group = 4 * FB;
        a.d_pitch = ((int64_t)wb + group - 1) / group * group;
        a.d_pitch = (a.d_pitch + 127) / 128 * 128;             // the scratch is ours: every row on a line of its own (the ring kernels' fast write-back)
        a.d_stride = a.d_pitch * y;
        a.D = (uint8_t*)scratch_get((size_t)a.d_stride * count + 512);
        if (!a.D) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "png_defilter: scratch allocation failed");
        a.store_tail_masked = 0;
    }
    // Few images: cut each into row segments at None / Sub rows so that more compute units than images take part (a 4K image
    // alone keeps one workgroup busy for 34 bands in turn).  Which rows qualify is data: the workgroups find out (take_segment).
    u32 nseg = 1;
    if (count < 512 && y >= 256) { nseg = 1024u / (u32)count; nseg = nseg > 8 ? 8 : nseg; while (nseg > 1 && y / nseg < 128) --nseg; }
    // From a few hundred bands on, the bands of the whole batch go through one work queue instead (k_png_defilter_queue):
    // every wave slot of the chip stays busy until the batch ends, and it does not depend on the filters.
    // GAMUT_HIP_PNG_QUEUE=0 / 1 forces the choice (measurements, tests).
    const u32 nbands = (y + 63) / 64;
    int dev = 0, cus = 0;
    GAMUT_HIP_CHECK(hipGetDevice(&dev));
    GAMUT_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    // One workgroup per image keeps a compute unit for ceil(bands / 8) rounds of bands (the last one on idle SIMDs), the images
    // take ceil(count / CUs) rounds of workgroups, and with fewer than two workgroups per compute unit nothing hides the hand-off
    // latency between the waves of one image's pipeline; the queue needs count * bands / (8 CUs) rounds whatever the filters.
    // Measured (RGBA8, Mpx/s, workgroups / queue): 341 x 1080p 374 k / 478 k, 64 x 4K 106 k / 240 k, 256 x 4K 202 k / 533 k,
    // 384 x 4K 365 k / 595 k, 512 x 4K 616 k / 643 k (random filters 512 k / 529 k): the queue from 1024 units on.
    // line-aligned loads of the stream (defilter_band_ring<..., AL>): rows of at least two lines; GAMUT_HIP_PNG_ALIGNED=0 / 1 forces either
    const char* al_env = getenv("GAMUT_HIP_PNG_ALIGNED");
    const bool aligned_asked = !rgba_fused && wb >= 16 && (al_env && *al_env ? atoi(al_env) != 0 : wb >= 256);
    // every row of every image on a 128-byte line of its own?  (the kernels' LN: write-back groups = the rows' own pieces; otherwise the
    // kernels that look at every image's rows and write back by the lines of memory -- defilter_band_ring, wb_lines)
    const bool lines = ((uintptr_t)a.D % 128) == 0 && (a.d_pitch % 128) == 0 && (a.d_offs ? offs_line_aligned : (count == 1 || a.d_stride % 128 == 0));
    // rows on 16-byte units at least?  If not, the write-back reads its chunks across two pieces of the ring (dwp) -- the oldest of them 16 pieces behind the
    // newest the ring holds, and the line-aligned drop has by then put the head of the next raw piece into the slot that piece ends in (a chunk of the stream
    // straddles the tile's last piece).  Such rows take the row-aligned loads, whose drops write whole slots only.
    const bool rows16 = ((uintptr_t)a.D % 16) == 0 && (a.d_pitch % 16) == 0 && (a.d_offs ? offs_line_aligned : (count == 1 || a.d_stride % 16 == 0));
    const bool aligned = aligned_asked && (lines || rows16);
    const char* queue_env = getenv("GAMUT_HIP_PNG_QUEUE");       // read per call: tests flip it
    const uint64_t units = (uint64_t)count * nbands;
    bool queue = wb >= 16 && (int64_t)a.d_pitch * 64 < (1ll << 31) && units < (1ull << 31) &&
                 (queue_env && *queue_env ? atoi(queue_env) != 0 : units >= 1024);
    if (queue) {
        const size_t words = QSTATE_HDR + (size_t)count * nbands;
        a.qstate = (u32*)scratch_get(words * 4, 1);
        if (!a.qstate) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "png_defilter: queue state allocation failed");
        GAMUT_HIP_CHECK(hipMemsetAsync(a.qstate, 0, words * 4, stream));
        a.count = (u32)count; a.nbands = nbands; a.nseg = 1;
        // group: images whose bands are dealt out side by side (band-major inside a group).  The whole batch: band b + 1 of an image is
        // then drawn count draws after band b, by when band b has the ~90 trips of head start band b + 1 needs (groups of 128 of
        // 512 4K images: waves queue up behind their producers, -25 %).  GAMUT_HIP_PNG_GROUP overrides (measurements).
        static const int group_env = [] { const char* e = getenv("GAMUT_HIP_PNG_GROUP"); return e && *e ? atoi(e) : 0; }();      // (read once: not a getenv per launch)
        const u32 group = group_env > 0 ? (u32)group_env : (u32)count;
        const u32 ngroups = ((u32)count + group - 1) / group;
        a.group = ((u32)count + ngroups - 1) / ngroups;              // equal groups
        const unsigned wgs = (unsigned)std::min<uint64_t>((uint64_t)cus, (units + PNG_WAVES - 1) / PNG_WAVES);
        const dim3 qgrid(wgs), qblock(PNG_WAVES * 64);
        // the rolling form (defilter_band_roll: 9 KB of LDS per wave, ROLL_WPS waves per SIMD) -- only with GAMUT_HIP_PNG_ROLL=1 (every
        // row of at least one piece).  Bit-exact in every test, and NOT the default: 512 x 4K with random filters 7.95 ms against 7.53,
        // encoder filters 7.54 against 6.04 on one box (profiles/r05_png_roll2.txt).  Its vector work alone takes 6.17 / 2.97 ms
        // (profiles/r05_png_roll_abl.txt: the fast tiles without their loads and stores) -- but with 3 072 waves reading rows at odd
        // addresses a memory line is fetched by two groups of a row eight trips apart and the second fetch no longer finds it in
        // the L2: loads alone 5.1 ms for 17 GB.  It needs the line-aligned loads of defilter_band_ring<AL> in a form its 8-slot ring
        // can take (chunks into the ring as they lie in memory, the row's alignment undone per lane in registers) -- DESIGN 4.3.
        // (No co-residency is assumed by either form: a wave only ever waits for the unit drawn just before its own band-major, and whoever
        // drew that one is running or done -- units are drawn by waves that are resident, in queue order -- so a grid larger than what fits
        // runs its surplus workgroups when earlier ones retire, at a loss of time, never of progress.)
        const char* roll_env = getenv("GAMUT_HIP_PNG_ROLL");                // read per call, like GAMUT_HIP_PNG_QUEUE: the tests flip it inside one process
        const bool roll = !rgba_fused && wb >= 16 && roll_env && atoi(roll_env) != 0;
        if (roll) {
            constexpr int RW = 4;                                       // waves per workgroup: ROLL_WPS workgroups per compute unit
            const unsigned rwgs = (unsigned)std::min<uint64_t>((uint64_t)cus * PNG_ROLL_WPS, (units + RW - 1) / RW);
            switch (FB) {
#define GAMUT_PNG_CASE(N) case N: hipLaunchKernelGGL((k_png_defilter_rollq<N, RW, PNG_ROLL_WPS>), dim3(rwgs), dim3(RW * 64), 0, stream, a); break;
            GAMUT_PNG_CASE(1) GAMUT_PNG_CASE(2) GAMUT_PNG_CASE(3) GAMUT_PNG_CASE(4) GAMUT_PNG_CASE(6) GAMUT_PNG_CASE(8)
#undef GAMUT_PNG_CASE
            default: return set_error(GAMUT_HIP_ERR_INVALID_ARG, "png_defilter: unsupported filter unit %d", FB);
            }
        } else
        if (rgba_fused) { if (lines) hipLaunchKernelGGL((k_png_defilter_queue<3, PNG_WAVES, 2, true>), qgrid, qblock, 0, stream, a);
                          else       hipLaunchKernelGGL((k_png_defilter_queue<3, PNG_WAVES, 2, true, false, false>), qgrid, qblock, 0, stream, a); }
        else switch (FB) {
#define GAMUT_PNG_CASE(N) case N: if (aligned && lines) hipLaunchKernelGGL((k_png_defilter_queue<N, PNG_WAVES, 2, false, true>), qgrid, qblock, 0, stream, a); \
                                  else if (aligned)     hipLaunchKernelGGL((k_png_defilter_queue<N, PNG_WAVES, 2, false, true, false>), qgrid, qblock, 0, stream, a); \
                                  else if (lines)       hipLaunchKernelGGL((k_png_defilter_queue<N, PNG_WAVES, 2>), qgrid, qblock, 0, stream, a); \
                                  else                  hipLaunchKernelGGL((k_png_defilter_queue<N, PNG_WAVES, 2, false, false, false>), qgrid, qblock, 0, stream, a); break;
        GAMUT_PNG_CASE(1) GAMUT_PNG_CASE(2) GAMUT_PNG_CASE(3) GAMUT_PNG_CASE(4) GAMUT_PNG_CASE(6) GAMUT_PNG_CASE(8)
#undef GAMUT_PNG_CASE
        default: return set_error(GAMUT_HIP_ERR_INVALID_ARG, "png_defilter: unsupported filter unit %d", FB);
        }
    } else {
    a.nseg = nseg;
    const dim3 grid(nseg > 1 ? nseg : (unsigned)count, nseg > 1 ? (unsigned)count : 1u), block(PNG_WAVES * 64);
    // rows of at least one 16-byte piece: the LDS-ring kernel (coalesced, aligned I/O); 8 waves per workgroup, register
    // budget left unconstrained (no spills: measured faster than 128-VGPR variants that spill).  Narrower rows: the
    // per-lane kernel.
    const bool ring = wb >= 16;
    if (rgba_fused) { if (lines) hipLaunchKernelGGL((k_png_defilter_ring<3, PNG_WAVES, 2, true>), grid, block, 0, stream, a);
                      else       hipLaunchKernelGGL((k_png_defilter_ring<3, PNG_WAVES, 2, true, false, false>), grid, block, 0, stream, a); }
    else switch (FB) {
#define GAMUT_PNG_CASE(N) case N: if (ring && aligned && lines) hipLaunchKernelGGL((k_png_defilter_ring<N, PNG_WAVES, 2, false, true>), grid, block, 0, stream, a); \
                                  else if (ring && aligned) hipLaunchKernelGGL((k_png_defilter_ring<N, PNG_WAVES, 2, false, true, false>), grid, block, 0, stream, a); \
                                  else if (ring && lines) hipLaunchKernelGGL((k_png_defilter_ring<N, PNG_WAVES, 2>), grid, block, 0, stream, a); \
                                  else if (ring) hipLaunchKernelGGL((k_png_defilter_ring<N, PNG_WAVES, 2, false, false, false>), grid, block, 0, stream, a); \
                                  else      hipLaunchKernelGGL((k_png_defilter<N, PNG_WAVES>), grid, block, 0, stream, a); break;
    GAMUT_PNG_CASE(1) GAMUT_PNG_CASE(2) GAMUT_PNG_CASE(3) GAMUT_PNG_CASE(4) GAMUT_PNG_CASE(6) GAMUT_PNG_CASE(8)
#undef GAMUT_PNG_CASE
    default: return set_error(GAMUT_HIP_ERR_INVALID_ARG, "png_defilter: unsupported filter unit %d", FB);
    }
    }
    if (int rc = launch_status("png_defilter")) return rc;
    if (!fused) {
        ExpandArgs e{};
        e.D = a.D; e.d_stride = a.d_stride; e.d_pitch = a.d_pitch; e.out = out; e.out_stride = out_stride;
        e.x = x; e.y = y; e.img_n = img_n; e.out_n = out_n; e.depth = depth; e.color = color; e.out_offs = out_offs;
        const dim3 vgrid(((x + 3) / 4 + 255) / 256, y < 65535u ? y : 65535u, count);
        bool vec = depth >= 8 && count <= 65535;
        if (vec) {
#define GAMUT_PNG_EXPAND(I, O) if (img_n == I && out_n == O) { \
                if (depth == 8) hipLaunchKernelGGL((k_png_expand_vec<I, O, 1>), vgrid, dim3(256), 0, stream, e); \
                else            hipLaunchKernelGGL((k_png_expand_vec<I, O, 2>), vgrid, dim3(256), 0, stream, e); } else
            GAMUT_PNG_EXPAND(1, 1) GAMUT_PNG_EXPAND(1, 2) GAMUT_PNG_EXPAND(2, 2) GAMUT_PNG_EXPAND(3, 3) GAMUT_PNG_EXPAND(3, 4) GAMUT_PNG_EXPAND(4, 4)
            vec = false;
#undef GAMUT_PNG_EXPAND
        }
        if (!vec) hipLaunchKernelGGL(k_png_expand, dim3(blocks_for((int64_t)x * y), count), dim3(256), 0, stream, e);
        if (int rc = launch_status("png_expand")) return rc;
    }
    return GAMUT_HIP_OK;
}


int png_transparency_launch(void* img, int64_t npx, int out_n, int depth16, const uint16_t tc[3], hipStream_t st)
{
    if (depth16) hipLaunchKernelGGL(k_png_transparency<uint16_t>, dim3(blocks_for(npx)), dim3(256), 0, st, (uint16_t*)img, npx, out_n, tc[0], tc[1], tc[2], (uint16_t)65535);
    else hipLaunchKernelGGL(k_png_transparency<uint8_t>, dim3(blocks_for(npx)), dim3(256), 0, st, (uint8_t*)img, npx, out_n, (uint8_t)tc[0], (uint8_t)tc[1], (uint8_t)tc[2], (uint8_t)255);
    return launch_status("png_transparency");
}
int png_palette_launch(const uint8_t* idx, uint8_t* out, int64_t npx, int pal_n, const uint8_t* palette_dev, hipStream_t st)
{
    hipLaunchKernelGGL(k_png_palette, dim3(blocks_for(npx)), dim3(256), 0, st, idx, out, npx, pal_n, palette_dev);
    return launch_status("png_palette");
}
int png_convert_format_launch(const void* src, void* dst, int64_t npx, int img_n, int req, int depth16, hipStream_t st)
{
    if (depth16) hipLaunchKernelGGL(k_png_convert_format<uint16_t>, dim3(blocks_for(npx)), dim3(256), 0, st, (const uint16_t*)src, (uint16_t*)dst, npx, img_n, req, (uint16_t)0xffff);
    else hipLaunchKernelGGL(k_png_convert_format<uint8_t>, dim3(blocks_for(npx)), dim3(256), 0, st, (const uint8_t*)src, (uint8_t*)dst, npx, img_n, req, (uint8_t)255);
    return launch_status("png_convert_format");
}
int png_depth_convert_launch(const void* src, void* dst, int64_t n, int to16, hipStream_t st)
{
    if (to16) hipLaunchKernelGGL(k_png_8_to_16, dim3(blocks_for(n)), dim3(256), 0, st, (const uint8_t*)src, (uint16_t*)dst, n);
    else hipLaunchKernelGGL(k_png_16_to_8, dim3(blocks_for(n)), dim3(256), 0, st, (const uint16_t*)src, (uint8_t*)dst, n);
    return launch_status("png_depth_convert");
}
int png_adam7_scatter_launch(const uint8_t* pass, uint8_t* final_, uint32_t px, uint32_t py, uint32_t img_x, int out_bytes, int p, hipStream_t st)
{
    static const int xorig[7] = { 0,4,0,2,0,1,0 }, yorig[7] = { 0,0,4,0,2,0,1 }, xspc[7] = { 8,8,4,4,2,2,1 }, yspc[7] = { 8,8,8,4,4,2,2 };
    hipLaunchKernelGGL(k_png_adam7_scatter, dim3(blocks_for((int64_t)px * py)), dim3(256), 0, st, pass, final_, px, py, img_x, out_bytes,
                       xorig[p], yorig[p], xspc[p], yspc[p]);
    return launch_status("png_adam7_scatter");
}

} // namespace gamut
