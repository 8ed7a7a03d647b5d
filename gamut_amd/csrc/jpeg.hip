// jpeg.hip -- K1-K4: JPEG block reconstruction on gfx950.
//
// Replaces, for every MCU of every image of a batch, the reference's
//   transform_mcu / transform_mcu_expand   jpegload.d:2120-2255  (IDCT, H2V2 chroma upsample)
//   expanded_convert / H?V?Convert / gray  jpegload.d:2528-2823  (YCbCr -> RGBA)
//   output packing                         jpegload.d:3761-3801  (rgba8 / rgb8 / l8)
// Input: dense de-quantised int16 coefficients (what decode_next_row leaves in
// m_pMCU_coefficients, :2432/:2474), 128 B per block, blocks in MCU order.
//
// Two kernels:
//   k_jpeg_h2v2<OC>    -- the tuned path for 4:2:0 (the headline case -> rgba8; also rgb8, what loadJPEG produces by
//       default, plugins/jpeg.d:48-86, and l8).
//       One 256-thread workgroup reconstructs a strip of 8 MCUs (128x16 px).
//       HBM traffic is exactly the algorithmic one: 768 B of coefficients in
//       (dwordx4 per lane, lane-contiguous) and 1024 B of pixels out (each wave
//       store instruction writes two full 128-B lines).  The 8x8 transposes
//       between the row and the column pass, and the hand-off between the
//       upsample stages, go through LDS (padded to be bank-conflict free); the
//       VALU does only arithmetic.
//   k_jpeg_plain<ST,OC> -- tuned grey / 4:4:4 / 4:2:2 / 4:4:0 (two-pass IDCT as above, four pixels per thread in the colour stage).
//   k_jpeg_generic     -- every sampling mode (grey, H1V1, H2V1, H1V2, H2V2) and
//       every output format (l8 / rgb8 / rgba8): one thread per coefficient block
//       into an LDS sample buffer, then one thread per pixel.  Correct, untuned.
#include "common.hpp"
#include "jpeg_math.hpp"

namespace gamut {
namespace {

using namespace jpg;

constexpr int TILE_MCUS = 8;      // MCUs per workgroup (generic kernel)
#ifndef JPEG_H2V2_MCUS
#define JPEG_H2V2_MCUS 8
#endif
constexpr int H2V2_MCUS = JPEG_H2V2_MCUS;         // MCUs per workgroup of the tuned 4:2:0 kernel (a wave handles two): tuning knob
constexpr int H2V2_THREADS = 32 * H2V2_MCUS;

struct JpegArgs {
    const int16_t* coeffs; int64_t coeff_stride;     // int16 elements between images
    const uint8_t* max_zag; int64_t zag_stride;      // bytes between images (NULL = dense)
    uint8_t* out; int64_t out_pitch; int64_t out_stride;
    int width, height;
    int mcus_per_row, mcus_per_col;
    int scan_type, out_comps;
    int count;                                       // images of this launch
    // the compact hand-off (k_jpeg_h2v2<.., TOK>): instead of `coeffs`, image i's coefficients are the tokens
    // tokens[tok_offs[i] + strip_tab[strip_offs[i] + s] .. strip_tab[strip_offs[i] + s + 1]) of strip s (jpeg_host.hip, SUB_TOKENS)
    const u32* tokens; const u32* strip_tab; const int64_t* tok_offs; const int64_t* strip_offs;
    int nt;                                          // every row of every image starts on a 128-byte line (the launcher's verdict): the packed outputs' 16-byte chunks leave
                                                     // with the nontemporal hint (rgb8 2.81 -> 2.78 ms, l8 2.64 -> 2.62; off the lines the hint costs 15 %: profiles/r06_jpeg_packed_nt_ab.txt)
    int flip;                                        // rows stored bottom-up (the caller's pitch was negative): out = the image's LAST row in memory
};                                                   // order = its first row by address, out_pitch = |pitch|; image row y lives at row height - 1 - y

// =============================================================================
// tuned H2V2 -> rgba8
// =============================================================================
// LDS layout (ints).  Every 8x8 int tile is stored with a 72-int block stride and
// 8-int row stride: a column read (8 lanes x consecutive c, 4 blocks per 32-lane
// group) then touches 32 distinct banks.
constexpr int BLK_STRIDE = 72;
constexpr int T1_INTS = 4 * H2V2_MCUS * BLK_STRIDE;           // 32 Y blocks: pass-1 results
constexpr int H_INTS  = 2 * H2V2_MCUS * BLK_STRIDE;           // 16 chroma blocks: horizontal upsample stage H[k][m]
constexpr int V_INTS  = 2 * H2V2_MCUS * BLK_STRIDE;           // vertical stage V[n][m]
// T2 (32 (mcu,quadrant) tiles: rows 0-3 = Cb pass-1 rows, 4-7 = Cr) reuses T1's storage: T1 is last read in
// phase P2, T2 is first written in P3.  What separates the two is wave_sync(), NOT a workgroup barrier: it is enough only
// because tile n of T1 and tile n of T2 are both touched by the threads with t >> 3 == n alone (one 8-lane group of one
// wave).  Any change of the thread -> tile mappings must keep that, or put a __syncthreads() between P2 and P3.
// The V area holds Xs, the int16 coefficients of the four expanded blocks per chroma block (2 KB); the rgb8 / l8 variants
// reuse H + V as their output staging.
constexpr int LDS_INTS = T1_INTS + H_INTS + V_INTS;

// Both upsample maps have the same shape: two pass-through inputs and two rounded 4-term sums (jpegload.d:929-952,
// 1001-1032).  In the horizontal stage a lane evaluates either the "E" or the "O" map depending on its `half` bit; instead
// of branching (both sides would run for every wave) it carries its multipliers in registers and stores results in the fixed
// physical order (pass0, pass1, sumA, sumB).  Logical index m of the 8 outputs (0-3 = E0..E3, 4-7 = O0..O3) <-> physical slot:
//   E: e0=pass0 e1=sumA e2=pass1 e3=sumB     O: o0=sumA o1=pass0 o2=sumB o3=pass1
__device__ __forceinline__ constexpr int phys_slot(int m) { constexpr int P[8] = { 0, 2, 1, 3, 6, 4, 7, 5 }; return P[m]; }

// the same multipliers as packed int16 pairs (u1|u3, u5|u7) for the horizontal stage, whose inputs are int16 (v_dot2_i32_i16)
__device__ const u32 kMapPacked[2][4] __attribute__((aligned(16))) = {
    { pk16(E1a, E1b), pk16(E1c, E1d), pk16(E3a, E3b), pk16(E3c, E3d) }, { pk16(O0a, O0b), pk16(O0c, O0d), pk16(O2a, O2b), pk16(O2c, O2d) } };
// vertical stage, by output index j: the rounded sum of (E_j, O_j) is O0 / E1 / O2 / E3 (the other one passes row 2j through)
__device__ const i32 kVTable[4][4] __attribute__((aligned(16))) = { { O0a, O0b, O0c, O0d }, { E1a, E1b, E1c, E1d }, { O2a, O2b, O2c, O2d }, { E3a, E3b, E3c, E3d } };
struct __attribute__((packed, aligned(1))) Dwords4u { u32 v[4]; };      // 16 bytes at any alignment (gfx950 stores them natively)

// Every LDS hand-off of the tuned kernel stays inside one 32-lane half of a wave (thread t only ever reads tiles
// written by threads with the same t >> 5: Y tile t>>3, chroma block t>>4, T2 tile t>>3), so no workgroup barrier is
// needed: LDS operations of one wave execute in program order, and this fence only stops the compiler from moving
// them across the phase boundary.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#ifndef JPEG_NT_LOADS             // coefficients are read exactly once: nontemporal loads (A/B on one box: 2.59 -> 2.57 ms)
#define JPEG_NT_LOADS 1
#endif
__device__ __forceinline__ uint4 load_coeffs16(const int16_t* p)
{
    typedef u32 u32x4n __attribute__((ext_vector_type(4)));
    if (JPEG_NT_LOADS) { const u32x4n v = __builtin_nontemporal_load(reinterpret_cast<const u32x4n*>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
    return *reinterpret_cast<const uint4*>(p);
}

// one nontemporal dword store at (scalar row base + 32-bit lane offset).  Written by hand: the compiler otherwise keeps the
// address as a 64-bit vector value and advances it with a 64-bit vector add per row.
__device__ __forceinline__ void store_px_nt(uint8_t* row_base, u32 voff, u32 px)
{
    asm volatile("global_store_dword %0, %1, %2 nt" :: "v"(voff), "v"(px), "s"(row_base));
}
// The same without the hint, for rows that do not start on a 128-byte line (tight rgba8 rows of a width that is no multiple of 32 pixels): a wave's
// 128-byte run then straddles two lines, each line is written by two workgroups at different times, and with the nontemporal hint every piece goes
// to memory on its own -- plain stores let the pieces of neighbouring strips meet in the L2 (the XCD-aware order keeps them on one).  Round 6, A/B on
// one box (profiles/r06_jpeg_nt_ab.txt): 2048 x 1366x768 3.31-3.41 -> 2.70 ms, 1024 x 1080x1920 2.82-2.86 -> 2.71 ms; on lines the hint is 1.5 % better.
// The other sampling modes' kernels keep the hint everywhere: plain stores measured slower there on and off the lines.
__device__ __forceinline__ void store_px_plain(uint8_t* row_base, u32 voff, u32 px)
{
    asm volatile("global_store_dword %0, %1, %2" :: "v"(voff), "v"(px), "s"(row_base));
}

#ifndef JPEG_XCD_REMAP
#define JPEG_XCD_REMAP 1
#endif
#ifndef JPEG_STRIPS               // consecutive strips one workgroup reconstructs: rgba8 / packed (rgb8, l8) outputs (tuning knobs)
#define JPEG_STRIPS 1
#endif
#ifndef JPEG_STRIPS_PACKED
#define JPEG_STRIPS_PACKED 5
#endif
#ifndef JPEG_UNROLL_STRIPS        // 1 = the strip loop fully unrolled (exact s_waitcnt vmcnt counts across the strips' stores)
#define JPEG_UNROLL_STRIPS 0
#endif
#ifndef JPEG_444_MCUS             // 4:4:4: MCUs per strip: 10 MCUs = 30 blocks = 240 of the 256 threads at work (8 MCUs: 192).  Round 4, A/B on one box
#define JPEG_444_MCUS 10          // (profiles/r04_jpeg_444_var.txt, 1024 x 1080p): -> rgba8 8 / 10 / 12 / 16 MCUs (256 / 256 / 320 / 384 threads) 4.50 / 4.48 /
#endif                            // 5.37 / 5.16 ms; -> rgb8 4.44 / 4.93 / 5.51 / 5.08 ms (a strip row of 8 MCUs is 192 bytes: whole 64-byte sectors): 8 for rgb8
#ifndef JPEG_444_MCUS_RGB8
#define JPEG_444_MCUS_RGB8 8
#endif
// strips per workgroup of k_jpeg_plain, by sampling mode (A/B on one box, 1024 x 1080p -> rgba8, 1 / 2 / 4 / 8 strips: grey 1.69 / 1.56 /
// 1.49 / 1.47 ms, 4:4:4 5.77 / 5.44 / 5.44 / 5.29 ms, 4:2:2 3.32 / 3.48 / 3.42 / 3.46 ms, 4:4:0 3.39 / 3.34 / 3.42 / 3.45 ms)
constexpr int plain_mcus(int scan_type, int oc = 4) { return scan_type == GAMUT_JPGD_GRAYSCALE ? 32 : scan_type == GAMUT_JPGD_YH1V1 ? (oc == 3 ? JPEG_444_MCUS_RGB8 : JPEG_444_MCUS) : 8; }     // MCUs per strip
constexpr int plain_strips(int scan_type) { return (scan_type == GAMUT_JPGD_GRAYSCALE || scan_type == GAMUT_JPGD_YH1V1) ? 8 : 1; }
constexpr int plain_threads(int scan_type) { return scan_type == GAMUT_JPGD_YH1V1 ? ((JPEG_444_MCUS > JPEG_444_MCUS_RGB8 ? JPEG_444_MCUS : JPEG_444_MCUS_RGB8) * 3 * 8 + 63) / 64 * 64 : 256; }   // a thread per block row
#ifndef JPEG_ABLATE               // measurement only (tools/variant.sh): 1 = loads + stores, no arithmetic; 2 = no stores; 3 = no loads
#define JPEG_ABLATE 0
#endif

#ifndef JPEG_MIN_WAVES            // __launch_bounds__'s second argument: minimum waves per SIMD (0 = unconstrained register allocation)
#define JPEG_MIN_WAVES 0
#endif
template <int OC, bool TOK = false, bool NT = true>     // output components: 4 = rgba8, 3 = rgb8, 1 = l8 (grey of the RGB result, jpegload.d:3786-3792); TOK: coefficients as tokens;
                                                         // NT (rgba8): every row on a 128-byte line -> nontemporal pixel stores (the launcher's verdict; see store_px_plain)
#if JPEG_MIN_WAVES
__global__ __launch_bounds__(H2V2_THREADS, JPEG_MIN_WAVES) void k_jpeg_h2v2(JpegArgs a)
#else
__global__ __launch_bounds__(H2V2_THREADS) void k_jpeg_h2v2(JpegArgs a)
#endif
{
    __shared__ __attribute__((aligned(16))) i32 lds[LDS_INTS];
    i32* const T1 = lds;
    i32* const Hs = T1 + T1_INTS;
    i32* const Vs = Hs + H_INTS;
    i32* const T2 = T1;
    int16_t* const Xs = reinterpret_cast<int16_t*>(Vs);

    constexpr int STRIPS = OC == 4 ? JPEG_STRIPS : JPEG_STRIPS_PACKED;
    const int t = threadIdx.x;
    // Workgroup -> work.  (1) XCD-aware order: workgroups are dealt to the 8 XCDs round-robin by linear id (observed on gfx950,
    // not a contract -- it only matters for speed); with gridDim.x a multiple of 8 the XCD is blockIdx.x & 7.  Every XCD
    // takes whole images (image = 8 z + xcd) and walks their strips in raster order, so what an XCD's L2 collects at any time
    // are neighbours: adjacent 512-byte pieces of the same pixel rows, consecutive 6 KB runs of coefficients.  Measured on
    // the load + store skeleton of this kernel: 2.74 -> 2.40 ms per 1024 x 1080p (dispatch order spreads every pixel row
    // over all 8 L2s).  (2) A workgroup reconstructs STRIPS consecutive strips of one MCU row, fetching the
    // coefficients of strip s + 1 before it computes strip s.  For rgba8 one strip per workgroup is as fast as any (A/B on
    // one box: 1 / 2 / 3 / 5 / 15 strips = 2.55 / 2.63 / 2.57 / 2.64 / 2.84 ms); the packed outputs, whose workgroups end in
    // a barrier-separated staging pass, gain from 5 (rgb8 2.67 -> 2.63 ms).
#if JPEG_XCD_REMAP
    const int img = blockIdx.z * 8 + (blockIdx.x & 7), mcu_y = blockIdx.y, tile0 = (blockIdx.x >> 3) * STRIPS;
    if (img >= a.count) return;
#else
    const int img = blockIdx.z, mcu_y = blockIdx.y, tile0 = blockIdx.x * STRIPS;
#endif
    const int n_tiles = (a.mcus_per_row + H2V2_MCUS - 1) / H2V2_MCUS;
    // wave-uniform bases (scalar registers); per-thread parts are small 32-bit offsets
    const int16_t* cbase = TOK ? nullptr : a.coeffs + (int64_t)img * a.coeff_stride + ((int64_t)mcu_y * a.mcus_per_row + tile0 * H2V2_MCUS) * (6 * 64);
    // (bottom-up rows: the strip's 16 rows lie in descending order; the base is the strip's LAST row -- it may lie in front of the
    // image for the rows a partial strip does not have, which are never written -- and the row offsets count down)
    uint8_t* obase = a.out + (int64_t)img * a.out_stride + (a.flip ? (int64_t)(a.height - 1 - (mcu_y * 16 + 15)) : (int64_t)(mcu_y * 16)) * a.out_pitch +
                     (int64_t)tile0 * (H2V2_MCUS * 16 * OC);
    const uint8_t* zbase = a.max_zag ? a.max_zag + (int64_t)img * a.zag_stride + ((int64_t)mcu_y * a.mcus_per_row + tile0 * H2V2_MCUS) * 6 : nullptr;

    // ---- mapping A: thread = (Y block b = (mcu m, quadrant q), row/column index r) ----
    const int b = t >> 3, r = t & 7, m = b >> 2, q = b & 3;
    // ---- mapping B: thread = (chroma block cb = (mcu, comp), row k, half) ----
    const int cbk = t >> 4, k = (t >> 1) & 7, half = t & 1;
    // the lane's multipliers as packed int16 pairs: one 16-byte load from a constant table
    const uint4 mp = *reinterpret_cast<const uint4*>(kMapPacked[half]);
    // ---- mapping C (vertical stage): thread = (chroma block cbk, column pair ci / 4 + ci, output index cj) ----
    const int ci = (t >> 2) & 3, cj = t & 3;
    const int pe = ((ci & 1) << 1) | (ci >> 1);                   // phys_slot(ci), phys_slot(4 + ci)
    const int po = 4 + ((((ci & 1) ^ 1) << 1) | (ci >> 1));
    const int4 vc = *reinterpret_cast<const int4*>(kVTable[cj]);
    const i32 sg = (cj & 1) ? -1 : 1;

    const ColourConsts cc = colour_consts();
    const i32 c_round = col_round();
    const int lx = m * 16 + (q & 1) * 8 + r;               // pixel column inside the strip
    const int ly0 = (q >> 1) * 8;                          // first pixel row inside the strip
    const bool all_rows = a.height - mcu_y * 16 >= 16;     // wave-uniform: all 16 rows of the strip exist
    const int rows_here = a.height - mcu_y * 16 - ly0;
    const u32 pitch = (u32)a.out_pitch;                    // the launcher checks 16 * out_pitch < 2^31
    const u32 voff = (u32)(lx * 4) + (u32)(a.flip ? 8 - ly0 : ly0) * pitch;     // rgba8: the lane's part of a pixel address; the row advances on the scalar side

    // P0: loads (16 B per lane, lane-contiguous inside each MCU).  Lanes of MCUs beyond the edge of the image re-read the
    // strip's first MCU (there is always one): their results are never stored, and nothing has to be zeroed.
    const u32 yoff = (u32)(m * 384 + q * 64 + r * 8), yoff0 = (u32)(q * 64 + r * 8);
    const u32 coff = (u32)((cbk >> 1) * 384 + 256 + (cbk & 1) * 64 + k * 8), coff0 = (u32)(256 + (cbk & 1) * 64 + k * 8);
    auto fetch = [&](int tile, uint4& yv, uint4& cv) {
        const int here = a.mcus_per_row - tile * H2V2_MCUS;                              // >= 1
        const int16_t* base = cbase + (int64_t)(tile - tile0) * (H2V2_MCUS * 384);
#if JPEG_ABLATE == 3
        yv = make_uint4(t + tile, t * 3, t * 5, t * 7); cv = make_uint4(t * 11, t, t * 13, t * 17 + tile); (void)here; (void)base;
#else
        yv = load_coeffs16(base + (m < here ? yoff : yoff0));
        cv = load_coeffs16(base + ((cbk >> 1) < here ? coff : coff0));
#endif
    };
    // the Y block's m_mcu_block_max_zag (when the caller has it): fetched with the coefficients, a strip ahead
    auto fetch_zag = [&](int tile) -> u32 {
        if (!zbase) return 64u;
        const int here = a.mcus_per_row - tile * H2V2_MCUS;
        return m < here ? (u32)zbase[((tile - tile0) * H2V2_MCUS + m) * 6 + q] : 1u;
    };
    // TOK: the strip's coefficients arrive as tokens (block of the strip, natural position, value): the tile the loads above would read
    // from memory is assembled in LDS instead -- cleared, the strip's tokens scattered into it by all threads, then every thread takes
    // its two rows from it.  The tile (48 blocks x 128 bytes) borrows the H / V staging area, which the strip's arithmetic writes later.
    auto fetch_tokens = [&](int tile, uint4& yv, uint4& cv) {
        if constexpr (TOK) {
            uint8_t* const ctile = reinterpret_cast<uint8_t*>(Hs);
            __syncthreads();                                          // (another wave may still read the area for the strip before)
            static_assert((H_INTS + V_INTS) * 4 >= 64 * 128, "the coefficient tile (any 6-bit block index) fits the H / V area");
            *reinterpret_cast<uint4*>(ctile + t * 16) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint2*>(ctile + H2V2_THREADS * 16 + t * 8) = make_uint2(0, 0);
            const u32* st = a.strip_tab + a.strip_offs[img] + ((int64_t)mcu_y * n_tiles + tile);
            const u32 t0 = st[0], t1 = st[1];
            const u32* tk = a.tokens + a.tok_offs[img];
            __syncthreads();
            for (u32 idx = t0 + (u32)t; idx < t1; idx += H2V2_THREADS) {
                const u32 w = tk[idx];
                *reinterpret_cast<uint16_t*>(ctile + ((w >> 26) << 7) + ((w >> 19) & 126u)) = (uint16_t)w;
            }
            __syncthreads();
            yv = *reinterpret_cast<const uint4*>(ctile + yoff * 2);
            cv = *reinterpret_cast<const uint4*>(ctile + coff * 2);
            __syncthreads();                                          // the H stage of this strip overwrites the tile
        }
    };
    uint4 yrow, crow;
    u32 yzag = 64;
    if (!TOK && tile0 < n_tiles) { fetch(tile0, yrow, crow); yzag = fetch_zag(tile0); }

#if JPEG_UNROLL_STRIPS
    #pragma unroll
#else
    #pragma unroll 1
#endif
    for (int s = 0; s < STRIPS; ++s) {
        const int tile = tile0 + s;
        if (tile >= n_tiles) break;                                // wave-uniform
        const int mcu_x0 = tile * H2V2_MCUS;
        const int mcus_here = min(H2V2_MCUS, a.mcus_per_row - mcu_x0);
        const bool mcu_live = m < mcus_here;
        uint8_t* const otile = obase + (size_t)s * (H2V2_MCUS * 16 * OC);
        if constexpr (TOK) { fetch_tokens(tile, yrow, crow); yzag = fetch_zag(tile); }
        uint4 ynext = yrow, cnext = crow;
        u32 znext = yzag;
        if (!TOK && s + 1 < STRIPS && tile + 1 < n_tiles) { fetch(tile + 1, ynext, cnext); znext = fetch_zag(tile + 1); }     // in flight during this strip's arithmetic
        // jpgd picks Row!N / Col!N per block from max_zag (jpegload.d:295-376): literal zeros for the coefficients a block does
        // not have, i.e. the dense transform at a fraction of the work.  A wave cannot branch per block, but it can per WAVE:
        // when none of its 8 Y blocks reaches zig-zag position 10 (row 4 / column 4) -- the smooth parts of a photograph -- both
        // luma passes are idct_4x4's (Row!4 on rows 0-3, Col!4).  Same bits: the butterfly is linear and evaluated mod 2^32.
        const bool y_sparse = zbase && __builtin_amdgcn_ballot_w64(yzag > 10u) == 0;                   // wave-uniform

#if JPEG_ABLATE == 1
        if constexpr (OC == 4) {
            const u32 v0 = yrow.x ^ yrow.y ^ yrow.z ^ yrow.w ^ crow.x ^ crow.y ^ crow.z ^ crow.w;
            if (mcu_live && mcu_x0 * 16 + lx < a.width && all_rows) {
                #pragma unroll
                for (int i = 0; i < 8; ++i) store_px_nt(otile + (size_t)i * pitch, voff, v0 + i);
            }
            yrow = ynext; crow = cnext; yzag = znext; (void)y_sparse;
            continue;
        }
#endif

        // P1a: luma pass 1 (row r of block b) -> T1[b][r][0..7]
        if (y_sparse) {                                                // rows 4-7 are not read by the Col!4 pass below
            i32 tv[8];
            row_pass4_pairs(__builtin_amdgcn_perm(yrow.y, yrow.x, 0x05040100u), __builtin_amdgcn_perm(yrow.y, yrow.x, 0x07060302u), tv);
            if (r < 4) {
                i32* dst = T1 + b * BLK_STRIDE + r * 8;
                *reinterpret_cast<int4*>(dst)     = make_int4(tv[0], tv[1], tv[2], tv[3]);
                *reinterpret_cast<int4*>(dst + 4) = make_int4(tv[4], tv[5], tv[6], tv[7]);
            }
        } else {
            i32 tv[8];
            row_pass_packed(yrow, tv);
            i32* dst = T1 + b * BLK_STRIDE + r * 8;
            *reinterpret_cast<int4*>(dst)     = make_int4(tv[0], tv[1], tv[2], tv[3]);
            *reinterpret_cast<int4*>(dst + 4) = make_int4(tv[4], tv[5], tv[6], tv[7]);
        }
        // P1b: chroma horizontal stage: source row k of chroma block cbk -> H[k][physical slots half*4 .. +3]
        {
            i32 hv[4];
            const u32 w0 = half ? crow.y : crow.x, w1 = half ? crow.w : crow.z;        // (u2 | u3) : (u0 | u1),  (u6 | u7) : (u4 | u5)
            const u32 p13 = __builtin_amdgcn_perm(crow.y, crow.x, 0x07060302u), p57 = __builtin_amdgcn_perm(crow.w, crow.z, 0x07060302u);
            hv[0] = (i32)(short)(w0 & 0xFFFF);
            hv[1] = (i32)(short)(w1 & 0xFFFF);
            hv[2] = dot2(p13, mp.x, dot2(p57, mp.y, 512)) >> 10;
            hv[3] = dot2(p13, mp.z, dot2(p57, mp.w, 512)) >> 10;
            *reinterpret_cast<int4*>(Hs + cbk * BLK_STRIDE + k * 8 + half * 4) = make_int4(hv[0], hv[1], hv[2], hv[3]);
        }
        wave_sync();

        // P2a: luma pass 2 (column r of block b) -> 8 samples in registers
        i32 ys[8];
        {
            const i32* src = T1 + b * BLK_STRIDE + r;
            i32 t0;
            if (y_sparse) {
                t0 = src[0];
                col_pass4_direct(t0, src[8], src[16], src[24], ys, c_round);
            } else {
                i32 tv[8];
                #pragma unroll
                for (int i = 0; i < 8; ++i) tv[i] = src[i * 8];
                col_pass<8>(tv, ys);
                t0 = tv[0];
            }
            if (zbase && __builtin_amdgcn_ballot_w64(yzag <= 2u) != 0) {   // wave-uniform: the reference's Col!(1) shortcut (max_zag <= 2), :222-232
                const bool y_col1 = mcu_live && yzag <= 2u;
                const i32 v = col1_sample(t0);
                #pragma unroll
                for (int i = 0; i < 8; ++i) ys[i] = y_col1 ? v : ys[i];
            }
        }
        // P2b: chroma vertical stage + quadrant combine.  Thread = (chroma block cbk, column pair ci / 4+ci, output index cj): for
        //     both columns it evaluates E_j and O_j over the 8 source rows -- one of the two is a pass-through of row 2j, the other
        //     a rounded 4-term sum over rows 1,3,5,7 (:955-989, :1036-1069) -- i.e. P[j][i], Q[j][i], R[j][i], S[j][i], and then
        //     the four coefficients blk_q[j][i] (:2230-2251, transposed store :886-902):
        //         blk0 = (P+Q)+(R+S), blk1 = (P+Q)-(R+S), blk2 = (P-Q)+(R-S), blk3 = (P-Q)-(R-S)
        //     as plain adds: with (pass, sum) for (E_j, O_j) when j is even and (O_j, E_j) when j is odd, P-Q = sg (pass - sum),
        //     sg = +-1 by the parity of j.  cast(jpgd_block_t) = the 16-bit store.  Xs[cbk][q][j] holds (x0, x2, x1, x3): the
        //     pairs idct_4x4's pass 1 multiplies (row_pass4_pairs).
        {
            const i32* he = Hs + cbk * BLK_STRIDE + pe;
            const i32* ho = Hs + cbk * BLK_STRIDE + po;
            const i32 pass_e = he[cj * 16], pass_o = ho[cj * 16];
            const i32 sum_e = mad24_1(vc.w, he[56], mad24_1(vc.z, he[40], mad24_1(vc.y, he[24], mad24_1(vc.x, he[8], 512)))) >> 10;
            const i32 sum_o = mad24_1(vc.w, ho[56], mad24_1(vc.z, ho[40], mad24_1(vc.y, ho[24], mad24_1(vc.x, ho[8], 512)))) >> 10;
            const i32 pa = wadd(pass_e, sum_e), pc = wadd(pass_o, sum_o);
            const i32 pu = wsub(pass_e, sum_e), pv = wsub(pass_o, sum_o);
            int16_t* dst = Xs + cbk * 64 + cj * 4 + pe;                   // [cbk][q][j][slot]: 64, 16, 4, 1 elements
            dst[0]  = (int16_t)wadd(pa, pc);
            dst[16] = (int16_t)wsub(pa, pc);
            dst[32] = (int16_t)mad24_1(wadd(pu, pv), sg, 0);
            dst[48] = (int16_t)mad24_1(wsub(pu, pv), sg, 0);
        }
        wave_sync();

        // P3: idct_4x4's pass 1 (Row!4) on row j of quadrant block qq of (mcu, comp) -> T2
        {
            const int mm = t >> 5, comp = (t >> 4) & 1, qq = (t >> 2) & 3, j = t & 3;
            const uint2 xp = *reinterpret_cast<const uint2*>(Xs + (mm * 2 + comp) * 64 + qq * 16 + j * 4);
            i32 tv[8];
            row_pass4_pairs(xp.x, xp.y, tv);
            i32* dst = T2 + (mm * 4 + qq) * BLK_STRIDE + (comp * 4 + j) * 8;
            *reinterpret_cast<int4*>(dst)     = make_int4(tv[0], tv[1], tv[2], tv[3]);
            *reinterpret_cast<int4*>(dst + 4) = make_int4(tv[4], tv[5], tv[6], tv[7]);
        }
        wave_sync();

        // P4: chroma pass 2 (Col!4 on column r of the quadrant's Cb and Cr), colour, store
        {
            i32 cbs[8], crs[8];
            const i32* src = T2 + b * BLK_STRIDE + r;
            col_pass4_direct(src[0], src[8], src[16], src[24], cbs, c_round);
            col_pass4_direct(src[32], src[40], src[48], src[56], crs, c_round);

            const bool px_live = mcu_live && mcu_x0 * 16 + lx < a.width;
            // Addresses = uniform strip base + 32-bit lane offset.  Pixel rows are written with nontemporal stores.
            if constexpr (OC == 4) {
                // one pixel per lane per row: a wave store instruction writes two full 128-byte lines.  (16-byte stores after an
                // in-quad DPP transpose were measured 2 % slower, and on the bare load + store skeleton no faster.)
                auto store_px = [](uint8_t* row_base, u32 vo, u32 px) { if constexpr (NT) store_px_nt(row_base, vo, px); else store_px_plain(row_base, vo, px); };
                if (px_live) {
                    if (JPEG_ABLATE == 2) {
                        u32 acc = 0;
                        #pragma unroll
                        for (int i = 0; i < 8; ++i) acc += ycc_to_rgba(ys[i], cbs[i], crs[i], cc.kr, cc.kb, cc.kg);
                        if (acc == 0x12345679u) store_px_nt(otile, voff, acc);
                    } else
                    if (all_rows) {
                        #pragma unroll
                        for (int i = 0; i < 8; ++i) store_px(otile + (size_t)(a.flip ? 7 - i : i) * pitch, voff, ycc_to_rgba(ys[i], cbs[i], crs[i], cc.kr, cc.kb, cc.kg));
                    } else {
                        #pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (i < rows_here) store_px(otile + (size_t)(a.flip ? 7 - i : i) * pitch, voff, ycc_to_rgba(ys[i], cbs[i], crs[i], cc.kr, cc.kb, cc.kg));
                    }
                }
            } else {
                // rgb8 / l8: a strip row is 384 / 128 bytes, but a wave's lanes hold 16-pixel runs of it, i.e. 48- / 16-byte
                // pieces: written directly they are partial lines (measured: 13 % more HBM traffic than the algorithmic bytes).
                // The strip is assembled in LDS (the H / V staging area is free by now) and written out in whole 16-byte
                // chunks, consecutive lanes along a row.
                // rgb8: the four pixels of a lane quad are 12 bytes = 3 dwords; lane j < 3 of the quad builds dword j from its own
                // pixel and its right neighbour's (one DPP quad shuffle + one byte permute).  l8: four grey bytes make one
                // dword, gathered with two in-quad OR steps; lane 0 of the quad keeps it.
                static_assert(OC == 3 || OC == 1, "output components");
                // (Round 4, again: the quads' dwords straight to memory without the staging, as k_jpeg_cols does where a wave's row is 192+ bytes --
                // here 3.15 instead of 2.71 ms for rgb8, 2.89 instead of 2.61 ms for l8, 12-19 % more HBM traffic: profiles/r04_h2v2_packed_direct.txt.)
                constexpr int BPR = 16 * H2V2_MCUS * OC;               // bytes per strip row
                uint8_t* stage = reinterpret_cast<uint8_t*>(Hs);       // 16 rows x BPR <= 6144 B of the 9216 B H/V area
                const int j = lx & 3;
                __syncthreads();                                       // other waves may still be reading Xs in P3
                #pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const u32 px = ycc_to_rgba(ys[i], cbs[i], crs[i], cc.kr, cc.kb, cc.kg);
                    if constexpr (OC == 3) {
                        const u32 sel = j == 0 ? 0x04020100u : j == 1 ? 0x05040201u : 0x06050402u;
                        const u32 nx = (u32)__builtin_amdgcn_mov_dpp((int)px, 0xF9, 0xF, 0xF, true);      // quad_perm [1,2,3,3]: right neighbour
                        const u32 dw = __builtin_amdgcn_perm(nx, px, sel);
                        if (j < 3) *reinterpret_cast<u32*>(stage + (ly0 + i) * BPR + lx * 3 + j) = dw;        // (lx - j) * 3 + 4 j
                    } else {
                        u32 v = rgb_to_luma(px) << (8 * j);
                        v |= (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);                   // quad_perm [1,0,3,2]
                        v |= (u32)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);                   // quad_perm [2,3,0,1]
                        if (j == 0) *reinterpret_cast<u32*>(stage + (ly0 + i) * BPR + lx) = v;
                    }
                }
                __syncthreads();
                const int row_bytes = min(mcus_here * 16, a.width - mcu_x0 * 16) * OC;       // live bytes of a strip row (rows are tight)
                const int live_rows = min(16, a.height - mcu_y * 16);
                constexpr int CPR = BPR / 16;                          // 16-byte chunks per row: 24 / 8
                for (int c = t; c < 16 * CPR; c += H2V2_THREADS) {
                    const int row = c / CPR, off = (c - row * CPR) * 16;
                    if (row >= live_rows || off >= row_bytes) continue;
                    const uint4 v = *reinterpret_cast<const uint4*>(stage + row * BPR + off);
                    uint8_t* o = otile + (u32)(a.flip ? 15 - row : row) * pitch + (u32)off;
                    if (off + 16 <= row_bytes) {
                        if (a.nt) {                                    // (workgroup-uniform) whole lines leave the strip: nothing of them is touched again
                            typedef u32 u32x4p __attribute__((ext_vector_type(4), aligned(4)));
                            __builtin_nontemporal_store(u32x4p{ v.x, v.y, v.z, v.w }, reinterpret_cast<u32x4p*>(o));
                        } else {
                            Dwords4u d; d.v[0] = v.x; d.v[1] = v.y; d.v[2] = v.z; d.v[3] = v.w;
                            *reinterpret_cast<Dwords4u*>(o) = d;
                        }
                    } else {
                        const u32 w[4] = { v.x, v.y, v.z, v.w };
                        for (int kk = 0; kk < row_bytes - off; ++kk) o[kk] = (uint8_t)(w[kk >> 2] >> ((kk & 3) * 8));
                    }
                }
                __syncthreads();                                       // the next strip's horizontal stage overwrites the staging area
            }
        }
        wave_sync();                                                   // T2 (read above) aliases the next strip's T1
        yrow = ynext; crow = cnext; yzag = znext;
    }
}

// =============================================================================
// tuned grey / 4:4:4 / 4:2:2 (no frequency-domain upsample: H1V1Convert :2528-2555, H2V1Convert :2558-2600,
// gray_convert :2715-2728 replicate chroma samples)
// =============================================================================
// A workgroup reconstructs a strip of 8 MCUs (32 for grey): the same two-pass IDCT as above (thread = block x row, then
// block x column, 8x8 transposes through LDS), samples dropped into an LDS byte buffer [block][row][col], then one thread
// per group of four horizontally adjacent pixels: whole-dword sample reads, four pixels packed into 16 / 12 / 4 bytes and
// stored with lane-contiguous addresses (16 lanes cover one row of the strip).
struct __attribute__((packed, aligned(4))) Dwords4 { u32 v[4]; };
struct __attribute__((packed, aligned(1))) Dwords3 { u32 v[3]; };
struct __attribute__((packed, aligned(1))) Dword1 { u32 v; };

template <int ST, int OC, bool NT = true>     // NT: rows on 128-byte lines -> nontemporal stores (see emit_px below)
__global__ __launch_bounds__(plain_threads(ST)) void k_jpeg_plain(JpegArgs a)
{
    constexpr int NTHR = plain_threads(ST);
    static_assert(ST == GAMUT_JPGD_GRAYSCALE || ST == GAMUT_JPGD_YH1V1 || ST == GAMUT_JPGD_YH2V1 || ST == GAMUT_JPGD_YH1V2, "sampling mode");
    constexpr int BPM  = ST == GAMUT_JPGD_GRAYSCALE ? 1 : ST == GAMUT_JPGD_YH1V1 ? 3 : 4;
    constexpr int MH   = ST == GAMUT_JPGD_YH1V2 ? 16 : 8;           // MCU height: 4:4:0 stacks two Y blocks (H1V2Convert :2603-2647)
    constexpr int MCUS = plain_mcus(ST, OC);
    constexpr int NBLK = MCUS * BPM;                              // 32 / 30 / 32 blocks = NBLK * 8 working threads
    static_assert(NBLK * 8 <= NTHR, "a thread per block row");
    constexpr int MW   = ST == GAMUT_JPGD_YH2V1 ? 16 : 8;
    constexpr int SW   = MCUS * MW;                               // strip width in pixels: 256 / 64 / 128
    __shared__ __attribute__((aligned(16))) i32 T1[NBLK * BLK_STRIDE];
    __shared__ __attribute__((aligned(16))) uint8_t S[NBLK * 64];

    const int t = threadIdx.x;
    const int img = blockIdx.z * 8 + (blockIdx.x & 7), mcu_y = blockIdx.y, bx = blockIdx.x >> 3;      // XCD-aware order: an image's strips on one XCD (gridDim.x is a multiple of 8)
    if (img >= a.count) return;
    // A strip is 3-4 KB of coefficients in and 2-4 KB of pixels out: a workgroup per strip made the launch millions of workgroups
    // (1024 x 1080p 4:4:4: 4.1 M) that the dispatcher hands out slower than the kernel could run them.  A workgroup takes
    // PLAIN_STRIPS consecutive strips of its MCU row where that pays (plain_strips).
    constexpr int PLAIN_STRIPS = plain_strips(ST);
    for (int strip = 0; strip < PLAIN_STRIPS; ++strip) {
    const int mcu_x0 = (bx * PLAIN_STRIPS + strip) * MCUS;
    if (mcu_x0 >= a.mcus_per_row) break;                          // workgroup-uniform
    const int64_t blk0 = ((int64_t)mcu_y * a.mcus_per_row + mcu_x0) * BPM;
    const int16_t* cbase = a.coeffs + (int64_t)img * a.coeff_stride + blk0 * 64;
    const int mcus_here = min(MCUS, a.mcus_per_row - mcu_x0);
    const int b = t >> 3, r = t & 7;
    const bool blk_live = b < mcus_here * BPM;

    if (t < NBLK * 8) {
        uint4 row = make_uint4(0, 0, 0, 0);
        if (blk_live) row = load_coeffs16(cbase + (u32)(t * 8));                          // block b, row r: lane-contiguous; read once: nontemporal
        i32 tv[8];
        row_pass_packed(row, tv);                                 // == row_pass<8>(unpack_row(row)), 34 instructions instead of 56
        i32* dst = T1 + b * BLK_STRIDE + r * 8;
        *reinterpret_cast<int4*>(dst)     = make_int4(tv[0], tv[1], tv[2], tv[3]);
        *reinterpret_cast<int4*>(dst + 4) = make_int4(tv[4], tv[5], tv[6], tv[7]);
    }
    wave_sync();                                                  // a block's 8 threads sit in one wave
    if (t < NBLK * 8) {
        i32 tv[8], sm[8];
        const i32* src = T1 + b * BLK_STRIDE + r;
        #pragma unroll
        for (int i = 0; i < 8; ++i) tv[i] = src[i * 8];
        col_pass<8>(tv, sm);
        if (a.max_zag) {                                          // wave-uniform: Col!(1) shortcut, see the 4:2:0 kernel
            bool col1 = false;
            if (blk_live) col1 = a.max_zag[(int64_t)img * a.zag_stride + blk0 + b] <= 2;
            const i32 v = col1_sample(tv[0]);
            #pragma unroll
            for (int i = 0; i < 8; ++i) sm[i] = col1 ? v : sm[i];
        }
        #pragma unroll
        for (int i = 0; i < 8; ++i) S[b * 64 + i * 8 + r] = (uint8_t)sm[i];
    }
    __syncthreads();

    uint8_t* obase = a.out + (int64_t)img * a.out_stride + (int64_t)(mcu_y * MH) * a.out_pitch + (int64_t)mcu_x0 * (MW * OC);
    const int rows_here = min(MH, a.height - mcu_y * MH);
    const int px_here = min(mcus_here * MW, a.width - mcu_x0 * MW);             // live pixels of a strip row
    constexpr int GPR = SW / 4;                                   // groups of 4 pixels per strip row
    for (int g = t; g < GPR * MH; g += NTHR) {
        const int y = g / GPR, x0 = (g - y * GPR) * 4;
        if (y >= rows_here || x0 >= px_here) continue;
        const int m = x0 / MW, xin = x0 - m * MW;
        u32 ys, px[4];
        if constexpr (ST == GAMUT_JPGD_GRAYSCALE) {
            ys = *reinterpret_cast<const u32*>(S + m * 64 + y * 8 + xin);
        } else if constexpr (ST == GAMUT_JPGD_YH1V1) {
            const uint8_t* sp = S + m * 192 + y * 8 + xin;
            ys = *reinterpret_cast<const u32*>(sp);
            const u32 cbs = *reinterpret_cast<const u32*>(sp + 64), crs = *reinterpret_cast<const u32*>(sp + 128);
            #pragma unroll
            for (int j = 0; j < 4; ++j) px[j] = ycc_to_rgba((ys >> (8 * j)) & 255, (cbs >> (8 * j)) & 255, (crs >> (8 * j)) & 255);
        } else if constexpr (ST == GAMUT_JPGD_YH1V2) {            // Y block y >> 3, chroma row y >> 1
            const uint8_t* mp = S + m * 256;
            ys = *reinterpret_cast<const u32*>(mp + (y >> 3) * 64 + (y & 7) * 8 + xin);
            const u32 cbs = *reinterpret_cast<const u32*>(mp + 128 + (y >> 1) * 8 + xin), crs = *reinterpret_cast<const u32*>(mp + 192 + (y >> 1) * 8 + xin);
            #pragma unroll
            for (int j = 0; j < 4; ++j) px[j] = ycc_to_rgba((ys >> (8 * j)) & 255, (cbs >> (8 * j)) & 255, (crs >> (8 * j)) & 255);
        } else {
            const uint8_t* mp = S + m * 256;
            ys = *reinterpret_cast<const u32*>(mp + (xin >> 3) * 64 + y * 8 + (xin & 7));
            const u32 cbs = *reinterpret_cast<const uint16_t*>(mp + 128 + y * 8 + (xin >> 1));
            const u32 crs = *reinterpret_cast<const uint16_t*>(mp + 192 + y * 8 + (xin >> 1));
            #pragma unroll
            for (int j = 0; j < 4; ++j) px[j] = ycc_to_rgba((ys >> (8 * j)) & 255, (cbs >> (8 * (j >> 1))) & 255, (crs >> (8 * (j >> 1))) & 255);
        }
        uint8_t* o = obase + (int64_t)y * a.out_pitch + (int64_t)x0 * OC;
        const int npx = min(4, px_here - x0);
        u32 w[4];                                                 // the group's OC * 4 output bytes
        if constexpr (ST == GAMUT_JPGD_GRAYSCALE) {               // grey replicated, alpha 255 (:3761-3801)
            if constexpr (OC == 1) w[0] = ys;
            else if constexpr (OC == 3) { w[0] = __builtin_amdgcn_perm(ys, ys, 0x01000000u); w[1] = __builtin_amdgcn_perm(ys, ys, 0x02020101u); w[2] = __builtin_amdgcn_perm(ys, ys, 0x03030302u); }
            else { w[0] = __builtin_amdgcn_perm(ys, ys, 0x0d000000u); w[1] = __builtin_amdgcn_perm(ys, ys, 0x0d010101u); w[2] = __builtin_amdgcn_perm(ys, ys, 0x0d020202u); w[3] = __builtin_amdgcn_perm(ys, ys, 0x0d030303u); }
        } else {
            if constexpr (OC == 4) { w[0] = px[0]; w[1] = px[1]; w[2] = px[2]; w[3] = px[3]; }
            else if constexpr (OC == 3) { w[0] = __builtin_amdgcn_perm(px[1], px[0], 0x04020100u); w[1] = __builtin_amdgcn_perm(px[2], px[1], 0x05040201u); w[2] = __builtin_amdgcn_perm(px[3], px[2], 0x06050402u); }
            else w[0] = rgb_to_luma(px[0]) | (rgb_to_luma(px[1]) << 8) | (rgb_to_luma(px[2]) << 16) | (rgb_to_luma(px[3]) << 24);
        }
        if (npx == 4) {                                           // written once, never read back here: nontemporal
            typedef u32 u32x4a __attribute__((ext_vector_type(4), aligned(4)));
            typedef u32 u32x3a __attribute__((ext_vector_type(3), aligned(1)));
            typedef u32 u32x1a __attribute__((aligned(1)));
            if constexpr (NT) {
                if constexpr (OC == 4)      __builtin_nontemporal_store(u32x4a{ w[0], w[1], w[2], w[3] }, reinterpret_cast<u32x4a*>(o));
                else if constexpr (OC == 3) __builtin_nontemporal_store(u32x3a{ w[0], w[1], w[2] }, reinterpret_cast<u32x3a*>(o));
                else                        __builtin_nontemporal_store((u32x1a)w[0], reinterpret_cast<u32x1a*>(o));
            } else {
                if constexpr (OC == 4)      *reinterpret_cast<u32x4a*>(o) = u32x4a{ w[0], w[1], w[2], w[3] };
                else if constexpr (OC == 3) *reinterpret_cast<u32x3a*>(o) = u32x3a{ w[0], w[1], w[2] };
                else                        *reinterpret_cast<u32x1a*>(o) = (u32x1a)w[0];
            }
        } else {
            for (int k = 0; k < npx * OC; ++k) o[k] = (uint8_t)(w[k >> 2] >> ((k & 3) * 8));
        }
    }
    __syncthreads();                                              // T1 / S are rewritten by the next strip
    }
}

// =============================================================================
// grey and 4:4:4: a thread per (MCU, block row), then per (MCU, pixel column)
// =============================================================================
// k_jpeg_plain's 4:4:4 form is bound by the vector ALUs (92 % busy, 72 instructions per pixel; grey 95 %: profiles/r04_plain_pmc.txt): besides
// the IDCTs a pixel needs it writes every sample to LDS as a byte, reads it back in another thread, pulls the bytes apart again and leaves a
// third of the workgroup idle in the colour stage.  Here the thread that ran the column pass of column c of an MCU's Y block runs those of its
// Cb and Cr blocks as well and has the three samples of its eight pixels in registers: colour conversion and the stores follow without another
// trip through LDS -- a dword per row for rgba8 (a wave's 64 lanes cover 256 contiguous bytes of a row); for rgb8 and l8 four neighbouring
// lanes first pass their pixels along (DPP row shifts) so that the group's 12 or 4 bytes leave as whole dwords.
// Pass 1: the same thread takes row r of the MCU's blocks (H1V1Convert :2528-2561 consumes the Y, Cb, Cr blocks of one MCU; gray_convert
// :2715-2728 one Y block).  One LDS block slot per MCU, the components take turns in it: registers, not LDS, decide how many waves a SIMD holds.
// one pixel per lane of a row, lanes along x: rgba8 a dword each; rgb8 / l8: the four lanes of a quad (j = place in it) pass their pixels
// along so that whole dwords leave -- `whole`: the quad lies inside the image, else every lane writes its own bytes.  All lanes must call.
// NT: the nontemporal hint on the pixel stores -- for rows on 128-byte lines (a wave's run is whole lines); rows off the lines store plain, so that the
// partial lines of neighbouring strips meet in the L2 their XCD shares (the workgroup order below keeps an image's strips on one XCD).  Round 6, one box
// (profiles/r06_jpeg_cols_xr_ab.txt): 4:4:4 -> rgba8 2048 x 1366x768 5.0-5.4 -> 4.38 ms, 1024 x 1080x1920 4.65-4.72 -> 4.31-4.34; 1920x1080 (on lines) 3.59 either way.
#define GAMUT_EMIT_STORE(v, p) do { if constexpr (NT) __builtin_nontemporal_store((v), (p)); else *(p) = (v); } while (0)
template <int OC, bool NT>
__device__ __forceinline__ void emit_px(uint8_t* orow, int x, u32 px, u32 grey, int j, u32 sel3, bool mine, bool whole)
{
    typedef u32 u32x1a __attribute__((aligned(1)));
    if constexpr (OC == 4) {
        if (mine) GAMUT_EMIT_STORE(px, reinterpret_cast<u32*>(orow + (int64_t)x * 4));
    } else if constexpr (OC == 3) {
        const u32 nxt = (u32)__builtin_amdgcn_update_dpp(0, (int)px, 0x101, 0xF, 0xF, false);                 // row_shl:1 -- the pixel to the right
        const u32 d = __builtin_amdgcn_perm(nxt, px, sel3);
        if (whole) { if (mine && j < 3) GAMUT_EMIT_STORE((u32x1a)d, reinterpret_cast<u32x1a*>(orow + (int64_t)(x - j) * 3 + j * 4)); }
        else if (mine) { uint8_t* q = orow + (int64_t)x * 3; q[0] = (uint8_t)px; q[1] = (uint8_t)(px >> 8); q[2] = (uint8_t)(px >> 16); }
    } else {
        u32 g = grey;
        g |= (u32)__builtin_amdgcn_update_dpp(0, (int)g, 0x101, 0xF, 0xF, false) << 8;                      // + the pixel to the right
        g |= (u32)__builtin_amdgcn_update_dpp(0, (int)g, 0x102, 0xF, 0xF, false) << 16;                     // + the pair two to the right
        if (whole) { if (mine && j == 0) GAMUT_EMIT_STORE((u32x1a)g, reinterpret_cast<u32x1a*>(orow + x)); }
        else if (mine) orow[x] = (uint8_t)g;
    }
}

template <int ST, int OC, int MCUS, bool NT = true>     // MCUs per strip = threads / 8
__global__ __launch_bounds__(MCUS * 8) void k_jpeg_cols(JpegArgs a)
{
    static_assert(ST == GAMUT_JPGD_GRAYSCALE || ST == GAMUT_JPGD_YH1V1, "one block per component and MCU");
    constexpr int NC = ST == GAMUT_JPGD_GRAYSCALE ? 1 : 3;
    constexpr int STRIPS = 2;
    __shared__ __attribute__((aligned(16))) i32 T1[MCUS * BLK_STRIDE];
    const int t = threadIdx.x, m = t >> 3, r = t & 7;
    // XCD-aware order, as k_jpeg_h2v2: gridDim.x is a multiple of 8, the XCD is blockIdx.x & 7 and takes whole images (image = 8 z + xcd)
    const int img = blockIdx.z * 8 + (blockIdx.x & 7), mcu_y = blockIdx.y, bx = blockIdx.x >> 3;
    if (img >= a.count) return;
    const ColourConsts cc = colour_consts();
    const int rows_here = min(8, a.height - mcu_y * 8);
    const int j = t & 3;                                          // place in the group of four pixels that shares its output dwords (rgb8, l8)
    const u32 sel3 = j == 0 ? 0x04020100u : j == 1 ? 0x05040201u : 0x06050402u;
    for (int strip = 0; strip < STRIPS; ++strip) {
        const int mcu_x0 = (bx * STRIPS + strip) * MCUS;
        if (mcu_x0 >= a.mcus_per_row) break;                      // workgroup-uniform
        const int64_t blk0 = ((int64_t)mcu_y * a.mcus_per_row + mcu_x0) * NC;
        const int16_t* cbase = a.coeffs + (int64_t)img * a.coeff_stride + blk0 * 64;
        const bool live = mcu_x0 + m < a.mcus_per_row;
        uint4 rows[NC];
        #pragma unroll
        for (int comp = 0; comp < NC; ++comp) {
            rows[comp] = make_uint4(0, 0, 0, 0);
            if (live) rows[comp] = load_coeffs16(cbase + (u32)((m * NC + comp) * 64 + r * 8));
        }
        i32 smp[NC][8];
        #pragma unroll
        for (int comp = 0; comp < NC; ++comp) {
            i32 tv[8];
            row_pass_packed(rows[comp], tv);
            i32* dst = T1 + m * BLK_STRIDE + r * 8;
            *reinterpret_cast<int4*>(dst)     = make_int4(tv[0], tv[1], tv[2], tv[3]);
            *reinterpret_cast<int4*>(dst + 4) = make_int4(tv[4], tv[5], tv[6], tv[7]);
            wave_sync();                                          // an MCU's 8 threads sit in one wave
            const i32* src = T1 + m * BLK_STRIDE + r;
            #pragma unroll
            for (int i = 0; i < 8; ++i) tv[i] = src[i * 8];
            wave_sync();                                          // the slot is rewritten by the next component's pass 1
            col_pass<8>(tv, smp[comp]);
            if (a.max_zag) {                                      // Col!(1) shortcut, as in k_jpeg_plain
                bool col1 = false;
                if (live) col1 = a.max_zag[(int64_t)img * a.zag_stride + blk0 + m * NC + comp] <= 2;
                const i32 v = col1_sample(tv[0]);
                #pragma unroll
                for (int i = 0; i < 8; ++i) smp[comp][i] = col1 ? v : smp[comp][i];
            }
        }
        const int x = (mcu_x0 + m) * 8 + r;
        const bool mine = live && x < a.width;
        const bool whole = x - j + 4 <= a.width;                  // the group of four pixels lies inside the image: dword stores
        uint8_t* const o0 = a.out + (int64_t)img * a.out_stride + (int64_t)(mcu_y * 8) * a.out_pitch;
        #pragma unroll
        for (int y = 0; y < 8; ++y) {
            if (y >= rows_here) break;                            // workgroup-uniform
            uint8_t* const orow = o0 + (int64_t)y * a.out_pitch;
            u32 px;                                               // R, G, B, 255
            if constexpr (NC == 1) px = __builtin_amdgcn_perm((u32)smp[0][y], (u32)smp[0][y], 0x0d000000u);         // grey replicated (:3761-3801)
            else                   px = ycc_to_rgba(smp[0][y], smp[1][y], smp[NC - 1][y], cc.kr, cc.kb, cc.kg);
            u32 g = 0;
            if constexpr (OC == 1) { if constexpr (NC == 1) g = (u32)smp[0][y]; else g = rgb_to_luma(px); }
            emit_px<OC, NT>(orow, x, px, g, j, sel3, mine, whole);
        }
    }
}

// 4:2:2 (H2V1Convert :2558-2600: an MCU is two Y blocks side by side, 16 x 8 pixels, one chroma sample per pixel pair) and 4:4:0 (H1V2Convert
// :2603-2647: two Y blocks stacked, 8 x 16 pixels, one chroma sample per pair of rows) the same way: eight threads per MCU, each runs pass 1 on
// row r of the four blocks, then four column passes -- 4:4:0: column r of Y-top, Y-bottom, Cb, Cr = its 16 pixels; 4:2:2: the Y columns 2r and
// 2r + 1 (both in block r >> 2) and chroma column r = its 8 x 2 pixels.  Two LDS block slots per MCU: the Y pair, then the chroma pair.
template <int ST, int OC, int MCUS, bool NT = true>
__global__ __launch_bounds__(MCUS * 8) void k_jpeg_cols4(JpegArgs a)
{
    static_assert(ST == GAMUT_JPGD_YH2V1 || ST == GAMUT_JPGD_YH1V2, "four blocks per MCU");
    constexpr bool WIDE = ST == GAMUT_JPGD_YH2V1;
    constexpr int MH = WIDE ? 8 : 16;
    constexpr int STRIPS = 2;
    __shared__ __attribute__((aligned(16))) i32 T1[MCUS * 2 * BLK_STRIDE];
    const int t = threadIdx.x, m = t >> 3, r = t & 7;
    const int img = blockIdx.z * 8 + (blockIdx.x & 7), mcu_y = blockIdx.y, bx = blockIdx.x >> 3;      // XCD-aware order, as k_jpeg_cols
    if (img >= a.count) return;
    const ColourConsts cc = colour_consts();
    const int rows_here = min(MH, a.height - mcu_y * MH);
    for (int strip = 0; strip < STRIPS; ++strip) {
        const int mcu_x0 = (bx * STRIPS + strip) * MCUS;
        if (mcu_x0 >= a.mcus_per_row) break;                      // workgroup-uniform
        const int64_t blk0 = ((int64_t)mcu_y * a.mcus_per_row + mcu_x0) * 4;
        const int16_t* cbase = a.coeffs + (int64_t)img * a.coeff_stride + blk0 * 64;
        const bool live = mcu_x0 + m < a.mcus_per_row;
        uint4 rows[4];
        #pragma unroll
        for (int b = 0; b < 4; ++b) {
            rows[b] = make_uint4(0, 0, 0, 0);
            if (live) rows[b] = load_coeffs16(cbase + (u32)((m * 4 + b) * 64 + r * 8));
        }
        i32 smp[4][8];                                            // 4:2:2: Y column 2r, Y column 2r + 1, Cb, Cr; 4:4:0: Y top, Y bottom, Cb, Cr
        #pragma unroll
        for (int turn = 0; turn < 2; ++turn) {
            #pragma unroll
            for (int k = 0; k < 2; ++k) {
                i32 tv[8];
                row_pass_packed(rows[turn * 2 + k], tv);
                i32* dst = T1 + (m * 2 + k) * BLK_STRIDE + r * 8;
                *reinterpret_cast<int4*>(dst)     = make_int4(tv[0], tv[1], tv[2], tv[3]);
                *reinterpret_cast<int4*>(dst + 4) = make_int4(tv[4], tv[5], tv[6], tv[7]);
            }
            wave_sync();                                          // an MCU's 8 threads sit in one wave
            i32 tv[2][8];
            int blk[2];                                           // which block of the MCU each of the two columns belongs to (Col!(1) shortcut)
            #pragma unroll
            for (int k = 0; k < 2; ++k) {
                const bool pair_in_one = WIDE && turn == 0;       // 4:2:2 luma: both columns come from block r >> 2
                const int slot = pair_in_one ? (r >> 2) : k, col = pair_in_one ? ((2 * r) & 7) + k : r;
                blk[k] = turn * 2 + slot;
                const i32* src = T1 + (m * 2 + slot) * BLK_STRIDE + col;
                #pragma unroll
                for (int i = 0; i < 8; ++i) tv[k][i] = src[i * 8];
            }
            wave_sync();                                          // the slots are rewritten by the next turn / strip
            #pragma unroll
            for (int k = 0; k < 2; ++k) {
                col_pass<8>(tv[k], smp[turn * 2 + k]);
                if (a.max_zag) {                                  // Col!(1) shortcut, as in k_jpeg_plain
                    bool col1 = false;
                    if (live) col1 = a.max_zag[(int64_t)img * a.zag_stride + blk0 + m * 4 + blk[k]] <= 2;
                    const i32 v = col1_sample(tv[k][0]);
                    #pragma unroll
                    for (int i = 0; i < 8; ++i) smp[turn * 2 + k][i] = col1 ? v : smp[turn * 2 + k][i];
                }
            }
        }
        uint8_t* const o0 = a.out + (int64_t)img * a.out_stride + (int64_t)(mcu_y * MH) * a.out_pitch;
        if constexpr (!WIDE) {
            const int x = (mcu_x0 + m) * 8 + r, j = t & 3;
            const u32 sel3 = j == 0 ? 0x04020100u : j == 1 ? 0x05040201u : 0x06050402u;
            const bool mine = live && x < a.width, whole = x - j + 4 <= a.width;
            #pragma unroll
            for (int y = 0; y < 16; ++y)
                if (y < rows_here) {                              // workgroup-uniform (no break: the loop must unroll, smp[] is registers)
                    const u32 px = ycc_to_rgba(smp[y >> 3][y & 7], smp[2][y >> 1], smp[3][y >> 1], cc.kr, cc.kb, cc.kg);
                    u32 g = 0;
                    if constexpr (OC == 1) g = rgb_to_luma(px);
                    emit_px<OC, NT>(o0 + (int64_t)y * a.out_pitch, x, px, g, j, sel3, mine, whole);
                }
        } else {
            const int x0 = (mcu_x0 + m) * 16 + 2 * r;             // this thread's two pixels: x0, x0 + 1; a lane PAIR shares the dwords of four pixels
            const bool odd = r & 1;
            const int gx = x0 - (odd ? 2 : 0);
            const bool mine = live && x0 < a.width, both = x0 + 1 < a.width, whole = gx + 4 <= a.width;
            typedef u32 u32x1a __attribute__((aligned(1)));
            typedef u32 u32x2a __attribute__((ext_vector_type(2), aligned(1)));
            #pragma unroll
            for (int y = 0; y < 8; ++y) {
                if (y >= rows_here) break;                        // workgroup-uniform
                uint8_t* const orow = o0 + (int64_t)y * a.out_pitch;
                const u32 p0 = ycc_to_rgba(smp[0][y], smp[2][y], smp[3][y], cc.kr, cc.kb, cc.kg), p1 = ycc_to_rgba(smp[1][y], smp[2][y], smp[3][y], cc.kr, cc.kb, cc.kg);
                if constexpr (OC == 4) {
                    typedef u32 u32x2n __attribute__((ext_vector_type(2), aligned(4)));
                    if (mine) { if (both) GAMUT_EMIT_STORE((u32x2n{ p0, p1 }), reinterpret_cast<u32x2n*>(orow + (int64_t)x0 * 4));
                                else      GAMUT_EMIT_STORE(p0, reinterpret_cast<u32*>(orow + (int64_t)x0 * 4)); }
                } else if constexpr (OC == 3) {
                    const u32 left1 = (u32)__builtin_amdgcn_update_dpp(0, (int)p1, 0x111, 0xF, 0xF, false);           // row_shr:1 -- the left lane's second pixel
                    if (whole) {
                        if (mine) {
                            if (!odd) GAMUT_EMIT_STORE((u32x1a)__builtin_amdgcn_perm(p1, p0, 0x04020100u), reinterpret_cast<u32x1a*>(orow + (int64_t)gx * 3));
                            else      GAMUT_EMIT_STORE((u32x2a{ __builtin_amdgcn_perm(p0, left1, 0x05040201u), __builtin_amdgcn_perm(p1, p0, 0x06050402u) }),
                                                       reinterpret_cast<u32x2a*>(orow + (int64_t)gx * 3 + 4));
                        }
                    } else if (mine) {
                        uint8_t* q = orow + (int64_t)x0 * 3; q[0] = (uint8_t)p0; q[1] = (uint8_t)(p0 >> 8); q[2] = (uint8_t)(p0 >> 16);
                        if (both) { q[3] = (uint8_t)p1; q[4] = (uint8_t)(p1 >> 8); q[5] = (uint8_t)(p1 >> 16); }
                    }
                } else {
                    u32 g = rgb_to_luma(p0) | (rgb_to_luma(p1) << 8);
                    g |= (u32)__builtin_amdgcn_update_dpp(0, (int)g, 0x101, 0xF, 0xF, false) << 16;                 // + the right lane's two
                    if (whole) { if (mine && !odd) GAMUT_EMIT_STORE((u32x1a)g, reinterpret_cast<u32x1a*>(orow + gx)); }
                    else if (mine) { orow[x0] = (uint8_t)g; if (both) orow[x0 + 1] = (uint8_t)(g >> 8); }
                }
            }
        }
    }
}

// =============================================================================
// generic: all sampling modes, all output formats
// =============================================================================
// full dense IDCT of one block held by one thread (jpegload.d:308-376 without the sparse dispatch)
__device__ void idct_block(const int16_t* __restrict__ src, uint8_t* __restrict__ dst, bool col1)
{
    i32 tmp[64];
    #pragma unroll
    for (int r = 0; r < 8; ++r) {
        i32 x[8], tv[8];
        unpack_row(*reinterpret_cast<const uint4*>(src + r * 8), x);
        row_pass<8>(x, tv);
        #pragma unroll
        for (int c = 0; c < 8; ++c) tmp[r * 8 + c] = tv[c];
    }
    #pragma unroll
    for (int c = 0; c < 8; ++c) {
        i32 tv[8], s[8];
        #pragma unroll
        for (int r = 0; r < 8; ++r) tv[r] = tmp[r * 8 + c];
        col_pass<8>(tv, s);
        if (col1) { const i32 v = col1_sample(tv[0]); for (int r = 0; r < 8; ++r) s[r] = v; }
        #pragma unroll
        for (int r = 0; r < 8; ++r) dst[r * 8 + c] = (uint8_t)s[r];
    }
}

// one chroma block -> four expanded sample blocks (transform_mcu_expand :2152-2254)
__device__ void upsample_block(const int16_t* __restrict__ src, uint8_t* __restrict__ dst4)
{
    i32 H[8][8];      // H[k][m]: m<4 = E map, m>=4 = O map of source row k
    #pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        i32 u[8], e[4], o[4];
        unpack_row(*reinterpret_cast<const uint4*>(src + kk * 8), u);
        map_E(u, e); map_O(u, o);
        #pragma unroll
        for (int i = 0; i < 4; ++i) { H[kk][i] = e[i]; H[kk][4 + i] = o[i]; }
    }
    i32 V[8][8];      // V[n][m]
    #pragma unroll
    for (int mm = 0; mm < 8; ++mm) {
        i32 u[8], e[4], o[4];
        #pragma unroll
        for (int kk = 0; kk < 8; ++kk) u[kk] = H[kk][mm];
        map_E(u, e); map_O(u, o);
        #pragma unroll
        for (int i = 0; i < 4; ++i) { V[i][mm] = e[i]; V[4 + i][mm] = o[i]; }
    }
    for (int qq = 0; qq < 4; ++qq) {
        i32 tmp[4][8];
        #pragma unroll
        for (int j = 0; j < 4; ++j) {
            i32 x[8], tv[8];
            #pragma unroll
            for (int i = 0; i < 4; ++i) {
                const i32 p = (qq & 2) ? wsub(V[j][i], V[4 + j][i]) : wadd(V[j][i], V[4 + j][i]);
                const i32 s = (qq & 2) ? wsub(V[j][4 + i], V[4 + j][4 + i]) : wadd(V[j][4 + i], V[4 + j][4 + i]);
                x[i] = (i32)(short)((qq & 1) ? wsub(p, s) : wadd(p, s));
            }
            x[4] = x[5] = x[6] = x[7] = 0;
            row_pass<4>(x, tv);
            #pragma unroll
            for (int c = 0; c < 8; ++c) tmp[j][c] = tv[c];
        }
        #pragma unroll
        for (int c = 0; c < 8; ++c) {
            i32 tv[8] = { tmp[0][c], tmp[1][c], tmp[2][c], tmp[3][c], 0, 0, 0, 0 }, s[8];
            col_pass<4>(tv, s);
            #pragma unroll
            for (int rr = 0; rr < 8; ++rr) dst4[qq * 64 + rr * 8 + c] = (uint8_t)s[rr];
        }
    }
}

__global__ __launch_bounds__(256) void k_jpeg_generic(JpegArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t samples[TILE_MCUS * 12 * 64];
    const int t = threadIdx.x;
    const int img = blockIdx.z, mcu_y = blockIdx.y, mcu_x0 = blockIdx.x * TILE_MCUS;
    const int st = a.scan_type;
    const int nb = st == GAMUT_JPGD_GRAYSCALE ? 1 : st == GAMUT_JPGD_YH1V1 ? 3 : st == GAMUT_JPGD_YH2V2 ? 6 : 4;
    const int sb = st == GAMUT_JPGD_YH2V2 ? 12 : nb;                     // sample blocks per MCU (m_expanded_blocks_per_mcu)
    const int mcu_w = (st == GAMUT_JPGD_YH2V1 || st == GAMUT_JPGD_YH2V2) ? 16 : 8;
    const int mcu_h = (st == GAMUT_JPGD_YH1V2 || st == GAMUT_JPGD_YH2V2) ? 16 : 8;
    const int mcus_here = min(TILE_MCUS, a.mcus_per_row - mcu_x0);
    const int64_t blk0 = ((int64_t)mcu_y * a.mcus_per_row + mcu_x0) * nb;
    const int16_t* cbase = a.coeffs + (int64_t)img * a.coeff_stride + blk0 * 64;
    const uint8_t* zbase = a.max_zag ? a.max_zag + (int64_t)img * a.zag_stride + blk0 : nullptr;

    // phase 1: one thread per coefficient block
    if (t < mcus_here * nb) {
        const int m = t / nb, bi = t - m * nb;
        const int16_t* src = cbase + (int64_t)t * 64;
        uint8_t* dst = samples + m * sb * 64;
        if (st == GAMUT_JPGD_YH2V2 && bi >= 4) upsample_block(src, dst + (4 + (bi - 4) * 4) * 64);
        else idct_block(src, dst + bi * 64, zbase && zbase[t] <= 2);
    }
    __syncthreads();

    // phase 2: one thread per pixel of the strip (row-major inside the strip)
    const int strip_w = mcus_here * mcu_w;
    const int npx = strip_w * mcu_h;
    uint8_t* obase = a.out + (int64_t)img * a.out_stride;
    for (int p = t; p < npx; p += 256) {
        const int ly = p / strip_w, lx = p - ly * strip_w;
        const int m = lx / mcu_w, xin = lx - m * mcu_w;
        const int gx = mcu_x0 * mcu_w + lx, gy = mcu_y * mcu_h + ly;
        if (gx >= a.width || gy >= a.height) continue;
        const uint8_t* s = samples + m * sb * 64;
        u32 rgba;
        i32 yv;
        if (st == GAMUT_JPGD_GRAYSCALE) {                                   // gray_convert :2715-2728
            yv = s[ly * 8 + xin]; rgba = 0;
        } else if (st == GAMUT_JPGD_YH1V1) {                                // H1V1Convert :2528-2555
            const int o = ly * 8 + xin;
            yv = s[o]; rgba = ycc_to_rgba(yv, s[64 + o], s[128 + o]);
        } else if (st == GAMUT_JPGD_YH2V1) {                                // H2V1Convert :2558-2600
            yv = s[(xin >> 3) * 64 + ly * 8 + (xin & 7)];
            const int co = 2 * 64 + ly * 8 + (xin >> 1);
            rgba = ycc_to_rgba(yv, s[co], s[co + 64]);
        } else if (st == GAMUT_JPGD_YH1V2) {                                // H1V2Convert :2603-2647
            yv = s[(ly >> 3) * 64 + (ly & 7) * 8 + xin];
            const int co = 2 * 64 + (ly >> 1) * 8 + xin;
            rgba = ycc_to_rgba(yv, s[co], s[co + 64]);
        } else {                                                            // expanded_convert :2731-2823
            const int o = ((ly >> 3) * 2 + (xin >> 3)) * 64 + (ly & 7) * 8 + (xin & 7);
            yv = s[o]; rgba = ycc_to_rgba(yv, s[4 * 64 + o], s[8 * 64 + o]);
        }
        // output packing :3761-3801
        uint8_t* o = obase + (int64_t)gy * a.out_pitch + (int64_t)gx * a.out_comps;
        if (st == GAMUT_JPGD_GRAYSCALE) {
            o[0] = (uint8_t)yv;
            if (a.out_comps >= 3) { o[1] = o[2] = (uint8_t)yv; if (a.out_comps == 4) o[3] = 255; }
        } else if (a.out_comps == 4) {
            o[0] = (uint8_t)rgba; o[1] = (uint8_t)(rgba >> 8); o[2] = (uint8_t)(rgba >> 16); o[3] = 255;
        } else if (a.out_comps == 3) {
            o[0] = (uint8_t)rgba; o[1] = (uint8_t)(rgba >> 8); o[2] = (uint8_t)(rgba >> 16);
        } else {
            o[0] = (uint8_t)rgb_to_luma(rgba);
        }
    }
}

} // namespace

int jpeg_reconstruct_launch(const int16_t* coeffs, int64_t coeff_stride,
                            const uint8_t* max_zag, int64_t zag_stride,
                            uint8_t* out, int64_t out_pitch, int64_t out_stride,
                            int width, int height, int scan_type, int out_comps,
                            int count, hipStream_t stream)
{
    // limits of the reference decoder: jpegload.d:101-102, 1355-1381, 3134-3195
    if (width < 1 || height < 1 || width > 16384 || height > 16384)
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_reconstruct: bad image size %dx%d", width, height);
    if (scan_type < GAMUT_JPGD_GRAYSCALE || scan_type > GAMUT_JPGD_YH2V2)
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_reconstruct: bad scan type %d", scan_type);
    if (out_comps != 1 && out_comps != 3 && out_comps != 4)
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_reconstruct: out_comps must be 1, 3 or 4");
    if (count < 0) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_reconstruct: negative count");
    if (count == 0) return GAMUT_HIP_OK;
    if (!coeffs || !out) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_reconstruct: null pointer");
    if (((uintptr_t)coeffs & 15) || (count > 1 && (coeff_stride & 7)))
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_reconstruct: coefficient buffers must be 16-byte aligned");

    JpegArgs a{};
    a.coeffs = coeffs; a.coeff_stride = coeff_stride; a.max_zag = max_zag; a.zag_stride = zag_stride;
    a.out = out; a.out_pitch = out_pitch; a.out_stride = out_stride;
    a.width = width; a.height = height; a.scan_type = scan_type; a.out_comps = out_comps;
    const int mcu_w = (scan_type == GAMUT_JPGD_YH2V1 || scan_type == GAMUT_JPGD_YH2V2) ? 16 : 8;
    const int mcu_h = (scan_type == GAMUT_JPGD_YH1V2 || scan_type == GAMUT_JPGD_YH2V2) ? 16 : 8;
    a.mcus_per_row = (width + mcu_w - 1) / mcu_w;          // jpegload.d:3197-3198
    a.mcus_per_col = (height + mcu_h - 1) / mcu_h;
    const int tiles = (a.mcus_per_row + TILE_MCUS - 1) / TILE_MCUS;

    for (int i0 = 0; i0 < count; i0 += 65535) {             // gridDim.z limit
        const int n = count - i0 < 65535 ? count - i0 : 65535;
        JpegArgs c = a;
        c.count = n;
        c.coeffs += (int64_t)i0 * coeff_stride; c.out += (int64_t)i0 * out_stride;
        if (c.max_zag) c.max_zag += (int64_t)i0 * zag_stride;
        const dim3 grid(tiles, a.mcus_per_col, n);
        // tuned kernels: rgba8 needs dword-aligned rows; rgb8 / l8 rows may start anywhere (unaligned dword stores)
        // Rows stored bottom-up (a negative pitch: LAYOUT_VERT_FLIPPED is first-class in the reference, internals/types.d:498-501) take the
        // tuned kernels too: k_jpeg_plain addresses rows with the signed pitch as it is; k_jpeg_h2v2, whose row offsets are unsigned 32-bit
        // lane offsets behind a scalar base, gets the image's lowest row and |pitch| and counts its rows down (JpegArgs.flip).
        const int64_t apitch = out_pitch < 0 ? -out_pitch : out_pitch;
        const bool tuned = apitch > 0 && apitch < (1 << 27) &&
                           (out_comps != 4 || (((uintptr_t)out & 3) == 0 && (apitch & 3) == 0 && (out_stride & 3) == 0));
        if (tuned && out_pitch < 0 && scan_type == GAMUT_JPGD_YH2V2) { c.flip = 1; c.out_pitch = apitch; c.out += (int64_t)(height - 1) * out_pitch; }
        const bool on_lines = (((uintptr_t)out | (uintptr_t)apitch | (uintptr_t)(count > 1 ? (out_stride < 0 ? -out_stride : out_stride) : 0)) & 127u) == 0;     // every row of every image on a 128-byte line
        c.nt = on_lines ? 1 : 0;
        const char* const cols_env = getenv("GAMUT_HIP_JPEG_COLS");                                 // A/B and tests: "plain" = k_jpeg_plain (rounds 1-3) for every mode but 4:2:0
        const bool cols_tuned = !(cols_env && !strcmp(cols_env, "plain"));
        const int ps = plain_strips(scan_type);
        const int pm = plain_mcus(scan_type, out_comps);
        const dim3 grid32(8u * (unsigned)(((a.mcus_per_row + 31) / 32 + ps - 1) / ps), a.mcus_per_col, (n + 7) / 8);           // grey: 32 MCUs per strip; x & 7 = the image's place among eight
        const dim3 grid_plain(8u * (unsigned)(((a.mcus_per_row + pm - 1) / pm + ps - 1) / ps), a.mcus_per_col, (n + 7) / 8);
        const int strips420 = out_comps == 4 ? JPEG_STRIPS : JPEG_STRIPS_PACKED;
        const unsigned groups420 = (unsigned)(((a.mcus_per_row + H2V2_MCUS - 1) / H2V2_MCUS + strips420 - 1) / strips420);
#if JPEG_XCD_REMAP
        const dim3 grid420(8u * groups420, a.mcus_per_col, (n + 7) / 8);
#else
        const dim3 grid420(groups420, a.mcus_per_col, n);
#endif
#define GAMUT_JPEG_PLAIN(ST, G) do { \
            if (out_comps == 4)      hipLaunchKernelGGL((k_jpeg_plain<ST, 4>), G, dim3(plain_threads(ST)), 0, stream, c);      /* (rgba8 off the lines: plain stores 3 % slower here) */ \
            else if (out_comps == 3) { if (on_lines) hipLaunchKernelGGL((k_jpeg_plain<ST, 3>), G, dim3(plain_threads(ST)), 0, stream, c); else hipLaunchKernelGGL((k_jpeg_plain<ST, 3, false>), G, dim3(plain_threads(ST)), 0, stream, c); } \
            else                     { if (on_lines) hipLaunchKernelGGL((k_jpeg_plain<ST, 1>), G, dim3(plain_threads(ST)), 0, stream, c); else hipLaunchKernelGGL((k_jpeg_plain<ST, 1, false>), G, dim3(plain_threads(ST)), 0, stream, c); } } while (0)
        if (!tuned)                                   hipLaunchKernelGGL(k_jpeg_generic, grid, dim3(256), 0, stream, c);
        else if (scan_type == GAMUT_JPGD_GRAYSCALE)   GAMUT_JPEG_PLAIN(GAMUT_JPGD_GRAYSCALE, grid32);
        else if ((scan_type == GAMUT_JPGD_YH1V1 || scan_type == GAMUT_JPGD_GRAYSCALE) && cols_tuned) {
            // strips of 32 or 24 MCUs, whichever leaves fewer idle threads at the end of a row (1080p: 240 MCUs = 10 x 24)
            const int waste32 = (32 - a.mcus_per_row % 32) % 32, waste24 = (24 - a.mcus_per_row % 24) % 24;
            const bool m24 = waste24 * 32 < waste32 * 24;
            const dim3 g(8u * (unsigned)(((a.mcus_per_row + (m24 ? 23 : 31)) / (m24 ? 24 : 32) + 1) / 2), a.mcus_per_col, (n + 7) / 8);      // (x & 7 = the XCD = the image's place among eight)
#define GAMUT_JPEG_COLS(ST, OC) do { if (m24) { if (on_lines) hipLaunchKernelGGL((k_jpeg_cols<ST, OC, 24>), g, dim3(192), 0, stream, c); else hipLaunchKernelGGL((k_jpeg_cols<ST, OC, 24, false>), g, dim3(192), 0, stream, c); } \
                                     else     { if (on_lines) hipLaunchKernelGGL((k_jpeg_cols<ST, OC, 32>), g, dim3(256), 0, stream, c); else hipLaunchKernelGGL((k_jpeg_cols<ST, OC, 32, false>), g, dim3(256), 0, stream, c); } } while (0)
            if (scan_type == GAMUT_JPGD_YH1V1) { if (out_comps == 4) GAMUT_JPEG_COLS(GAMUT_JPGD_YH1V1, 4); else if (out_comps == 3) GAMUT_JPEG_COLS(GAMUT_JPGD_YH1V1, 3); else GAMUT_JPEG_COLS(GAMUT_JPGD_YH1V1, 1); }
            else                               { if (out_comps == 4) GAMUT_JPEG_COLS(GAMUT_JPGD_GRAYSCALE, 4); else if (out_comps == 3) GAMUT_JPEG_COLS(GAMUT_JPGD_GRAYSCALE, 3); else GAMUT_JPEG_COLS(GAMUT_JPGD_GRAYSCALE, 1); }
#undef GAMUT_JPEG_COLS
        }
        else if (scan_type == GAMUT_JPGD_YH1V1)       GAMUT_JPEG_PLAIN(GAMUT_JPGD_YH1V1, grid_plain);
        else if ((scan_type == GAMUT_JPGD_YH2V1 || scan_type == GAMUT_JPGD_YH1V2) && cols_tuned) {
            const int waste32 = (32 - a.mcus_per_row % 32) % 32, waste24 = (24 - a.mcus_per_row % 24) % 24;
            const bool m24 = waste24 * 32 < waste32 * 24;
            const dim3 g(8u * (unsigned)(((a.mcus_per_row + (m24 ? 23 : 31)) / (m24 ? 24 : 32) + 1) / 2), a.mcus_per_col, (n + 7) / 8);
#define GAMUT_JPEG_COLS4(ST, OC) do { if (m24) { if (on_lines) hipLaunchKernelGGL((k_jpeg_cols4<ST, OC, 24>), g, dim3(192), 0, stream, c); else hipLaunchKernelGGL((k_jpeg_cols4<ST, OC, 24, false>), g, dim3(192), 0, stream, c); } \
                                      else     { if (on_lines) hipLaunchKernelGGL((k_jpeg_cols4<ST, OC, 32>), g, dim3(256), 0, stream, c); else hipLaunchKernelGGL((k_jpeg_cols4<ST, OC, 32, false>), g, dim3(256), 0, stream, c); } } while (0)
            if (scan_type == GAMUT_JPGD_YH2V1) { if (out_comps == 4) GAMUT_JPEG_COLS4(GAMUT_JPGD_YH2V1, 4); else if (out_comps == 3) GAMUT_JPEG_COLS4(GAMUT_JPGD_YH2V1, 3); else GAMUT_JPEG_COLS4(GAMUT_JPGD_YH2V1, 1); }
            else                               { if (out_comps == 4) GAMUT_JPEG_COLS4(GAMUT_JPGD_YH1V2, 4); else if (out_comps == 3) GAMUT_JPEG_COLS4(GAMUT_JPGD_YH1V2, 3); else GAMUT_JPEG_COLS4(GAMUT_JPGD_YH1V2, 1); }
#undef GAMUT_JPEG_COLS4
        }
        else if (scan_type == GAMUT_JPGD_YH2V1)       GAMUT_JPEG_PLAIN(GAMUT_JPGD_YH2V1, grid_plain);
        else if (scan_type == GAMUT_JPGD_YH1V2)       GAMUT_JPEG_PLAIN(GAMUT_JPGD_YH1V2, grid_plain);
        else if (out_comps == 4 && on_lines) hipLaunchKernelGGL(k_jpeg_h2v2<4>, grid420, dim3(H2V2_THREADS), 0, stream, c);
        else if (out_comps == 4) hipLaunchKernelGGL((k_jpeg_h2v2<4, false, false>), grid420, dim3(H2V2_THREADS), 0, stream, c);
        else if (out_comps == 3) hipLaunchKernelGGL(k_jpeg_h2v2<3>, grid420, dim3(H2V2_THREADS), 0, stream, c);
        else                     hipLaunchKernelGGL(k_jpeg_h2v2<1>, grid420, dim3(H2V2_THREADS), 0, stream, c);
#undef GAMUT_JPEG_PLAIN
        if (int rc = launch_status("jpeg_reconstruct")) return rc;
    }
    return GAMUT_HIP_OK;
}

// 4:2:0 images whose coefficients arrive as token streams (the compact hand-off of gamut_hip_jpeg_decode_batch_device: jpeg_host.hip,
// SUB_TOKENS): same kernel, the strip's tile assembled in LDS from its tokens.  tok_offs / strip_offs: per-image offsets (device arrays).
int jpeg_reconstruct_tokens_launch(const uint32_t* tokens, const uint32_t* strip_tab, const int64_t* tok_offs, const int64_t* strip_offs,
                                   const uint8_t* max_zag, int64_t zag_stride, uint8_t* out, int64_t out_pitch, int64_t out_stride,
                                   int width, int height, int out_comps, int count, hipStream_t stream)
{
    if (width < 1 || height < 1 || width > 16384 || height > 16384 || count < 0 || (out_comps != 1 && out_comps != 3 && out_comps != 4) || !max_zag)
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_reconstruct_tokens: bad arguments");
    if (count == 0) return GAMUT_HIP_OK;
    const int64_t apitch = out_pitch < 0 ? -out_pitch : out_pitch;
    if (apitch <= 0 || apitch >= (1 << 27) || (out_comps == 4 && ((((uintptr_t)out & 3) != 0) || (apitch & 3) || (out_stride & 3))))
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg_reconstruct_tokens: output rows must be dword-aligned for rgba8");
    JpegArgs a{};
    a.tokens = tokens; a.strip_tab = strip_tab; a.max_zag = max_zag; a.zag_stride = zag_stride;
    a.out = out; a.out_pitch = apitch; a.out_stride = out_stride;
    a.width = width; a.height = height; a.scan_type = GAMUT_JPGD_YH2V2; a.out_comps = out_comps;
    a.mcus_per_row = (width + 15) / 16; a.mcus_per_col = (height + 15) / 16;
    if (out_pitch < 0) { a.flip = 1; a.out += (int64_t)(height - 1) * out_pitch; }
    for (int i0 = 0; i0 < count; i0 += 65528) {               // gridDim.z limit (a multiple of 8: the XCD-aware image order)
        const int n = count - i0 < 65528 ? count - i0 : 65528;
        JpegArgs c = a;
        c.count = n; c.out += (int64_t)i0 * out_stride; c.max_zag += (int64_t)i0 * zag_stride; c.tok_offs = tok_offs + i0; c.strip_offs = strip_offs + i0;
        const int strips420 = out_comps == 4 ? JPEG_STRIPS : JPEG_STRIPS_PACKED;
        const unsigned groups420 = (unsigned)(((a.mcus_per_row + H2V2_MCUS - 1) / H2V2_MCUS + strips420 - 1) / strips420);
#if JPEG_XCD_REMAP
        const dim3 grid420(8u * groups420, a.mcus_per_col, (n + 7) / 8);
#else
        const dim3 grid420(groups420, a.mcus_per_col, n);
#endif
        const bool on_lines = (((uintptr_t)out | (uintptr_t)apitch | (uintptr_t)(count > 1 ? (out_stride < 0 ? -out_stride : out_stride) : 0)) & 127u) == 0;
        c.nt = on_lines ? 1 : 0;
        if (out_comps == 4 && on_lines) hipLaunchKernelGGL((k_jpeg_h2v2<4, true>), grid420, dim3(H2V2_THREADS), 0, stream, c);
        else if (out_comps == 4) hipLaunchKernelGGL((k_jpeg_h2v2<4, true, false>), grid420, dim3(H2V2_THREADS), 0, stream, c);
        else if (out_comps == 3) hipLaunchKernelGGL((k_jpeg_h2v2<3, true>), grid420, dim3(H2V2_THREADS), 0, stream, c);
        else                     hipLaunchKernelGGL((k_jpeg_h2v2<1, true>), grid420, dim3(H2V2_THREADS), 0, stream, c);
        if (int rc = launch_status("jpeg_reconstruct_tokens")) return rc;
    }
    return GAMUT_HIP_OK;
}

} // namespace gamut
