// jpeg_prog.hpp -- progressive (SOF2) entropy decode ON THE GPU.  Included by jpeg_host.hip inside its namespace (it shares the
// marker parser, the Huffman table forms and the bit reader with the baseline device decoder).
//
// A progressive file is a sequence of scans (jpegload.d:3296-3664): DC first / DC refinement (all components, coefficient 0)
// and, per component, AC first / AC refinement scans over a band [Ss, Se] of the zig-zag order, each refining the bits the
// scans before it left (successive approximation Ah / Al).  Inside a scan the stream is one chain -- a symbol's length tells
// where the next begins, an AC refinement symbol is followed by one correction bit per ALREADY non-zero coefficient it
// passes, so even a decoder that knew a codeword boundary could not start there -- but:
//   * scans that share no coefficient (other component, or disjoint bands) are independent of each other, and a scan that stands on
//     another one (refines its bits) only needs the BLOCKS it is about to read: ONE launch decodes every scan of every file of the
//     batch, a wave per scan (per restart segment), and a scan follows the scans it stands on a few dozen units behind (progress
//     words in memory, ProgItem::dep) instead of waiting for them to end -- the decode time of a file is its longest scan, not the
//     sum over the levels of its script (round 4; until then: one launch per level of the script);
//   * restart intervals cut a scan into independent segments, as in baseline files;
//   * a DC refinement scan is one bit per block: a lane per bit;
//   * an AC refinement block is a wave's work, not a lane's: lane k holds the coefficient at zig-zag position k, the
//     positions with history are a ballot, "skip `run` zeros, correcting what is passed" is a rank search in that mask
//     (v_mbcnt), and every passed coefficient picks its own correction bit out of a 128-bit window of the stream -- the cost
//     is per SYMBOL (a few per block), not per coefficient (63 per block).
// First scans (one symbol per step, nothing to look at in memory) are decoded by lane 0 of a wave.  The scans write straight
// into the dense MCU-ordered coefficient buffer the reconstruction kernels read (block (c, bx, by) of a component is block
// mcu * nb + ... of the buffer), not yet de-quantised; k_prog_finalize then multiplies by the quantisation table and finds
// max_zag per block (load_next_row :2259-2333).  The host walks the markers, cuts the scans at RSTn, unstuffs them into one
// pinned image and uploads it once.  Same results as Progressive::run above (the host feeder), which is its oracle.

struct ProgImage {
    int64_t coeff_off, zag_off;        // int16 elements / bytes from the start of the caller's buffers
    int32_t nb, ny, comps, mcus_per_row, mcus_per_col, n_blocks;
    int32_t hs[3], vs[3], quant[3];    // quant: index of the natural-order factor table
    int32_t index;                     // the image's index in the caller's arrays (status word)
};
enum { PROG_DC_FIRST = 0, PROG_DC_REFINE = 1, PROG_AC_FIRST = 2, PROG_AC_REFINE = 3 };
struct ProgItem {                      // one restart interval (or whole scan) of one scan of one image
    uint64_t begin, end;               // the unstuffed segment in the blob (followed by 64 bytes of 0xFF)
    int32_t image, kind, ncomp, ss, se, al;
    int32_t comp[3], tab[3];           // components in scan order, their DC (DC scans) or AC table
    int32_t first_unit, n_units;       // units: MCUs of an interleaved scan, blocks (raster order of nbx x nby) of a single-component one
    int32_t nbx, level;                // level: host only (the order of a file's items)
    // One launch decodes every scan of every file; what a scan needs from earlier ones it waits for, item by item (indices into the
    // launch's item list, always lower than the item's own: items are taken in list order, so what an item waits for is running or done).
    // dep[i] >= 0: a scan that wrote coefficients this one reads or overwrites.  Bit i of dep_pipe: that scan walks the same units in
    // the same order (same components, same restart segment), so unit u may go ahead once the dependency has published u + 1 --
    // the scans of a file run a few dozen units behind each other instead of one after the other.  Otherwise: wait for all of it.
    // ctr >= 0 (scripts whose scans do not line up): the file's count of finished items; the item starts when it has reached ctr_need
    // (= the items of the file's lower levels) and adds one when it is done.
    int32_t dep[3], dep_pipe, ctr, ctr_need;
    int32_t scan, seg;                 // host only: the scan of the file, the restart segment of the scan
    int32_t prio, tail;                // prio: s_setprio of the item's wave: the long chains of a file first (they are its decode time), the short ones in the gaps;
                                       // tail: as DevItem.tail (-1: the scan ends behind the segment; >= 0: an RSTn follows, behind that many fill bytes)
};
constexpr uint32_t kProgDone = 0x7FFFFFFFu;     // an item's progress word once it has returned (whatever its verdict)
constexpr int kProgBatch = 64;                  // units between two looks at / reports of progress

// block (bx, by) of component c in the MCU-ordered buffer
__device__ __forceinline__ int64_t prog_block(const ProgImage& im, int c, int bx, int by)
{
    const int hs = c == 0 ? im.hs[0] : c == 1 ? im.hs[1] : im.hs[2], vs = c == 0 ? im.vs[0] : c == 1 ? im.vs[1] : im.vs[2];
    const int mx = bx / hs, my = by / vs;
    const int off = c == 0 ? 0 : im.ny + c - 1;
    return ((int64_t)my * im.mcus_per_row + mx) * im.nb + off + (by - my * vs) * hs + (bx - mx * hs);
}
// the b-th block of unit u of the item's scan (interleaved: component order of the scan, rows of the component inside the MCU)
__device__ __forceinline__ int64_t prog_unit_block(const ProgImage& im, const ProgItem& it, int u, int b, int& ci)
{
    if (it.ncomp == 1) { ci = 0; const int by = u / it.nbx; return prog_block(im, it.comp[0], u - by * it.nbx, by); }
    const int my = u / im.mcus_per_row, mx = u - my * im.mcus_per_row;
    int i = 0, rest = b;
    for (; i < it.ncomp - 1; ++i) {
        const int c = i == 0 ? it.comp[0] : it.comp[1];
        const int n = (c == 0 ? im.hs[0] : c == 1 ? im.hs[1] : im.hs[2]) * (c == 0 ? im.vs[0] : c == 1 ? im.vs[1] : im.vs[2]);
        if (rest < n) break;
        rest -= n;
    }
    ci = i;
    const int c = i == 0 ? it.comp[0] : i == 1 ? it.comp[1] : it.comp[2];
    const int hs = c == 0 ? im.hs[0] : c == 1 ? im.hs[1] : im.hs[2], vs = c == 0 ? im.vs[0] : c == 1 ? im.vs[1] : im.vs[2];
    const int v = rest / hs, h = rest - v * hs;
    return prog_block(im, c, mx * hs + h, my * vs + v);
}

// The bits of an unstuffed segment from any bit position, for a whole wave: lane L keeps the big-endian dword at dword
// chunk + L of the segment (256 bytes per wave, reloaded when the position leaves them); 33 or more valid bits at the top of 64
// are two v_readlane and one shift -- enough for the longest code (16) plus the longest value behind it (16).
// Everything about the position is wave-uniform and kept in scalar registers (readfirstlane where the compiler cannot
// see it): the chain of a scan runs on the scalar unit.  A lone wave issues a dependent instruction every ~8 clocks, scalar or
// vector, a v_readlane round trip costs ~20 and a taken branch ~30 (tools/microbench/chain_latency.hip, profiles/r04_chain_latency.txt):
// the chains below are counted in instructions.
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t rfl(uint64_t v) { return ((uint64_t)rfl((uint32_t)(v >> 32)) << 32) | rfl((uint32_t)v); }
__device__ __forceinline__ int64_t rfl(int64_t v) { return (int64_t)rfl((uint64_t)v); }
struct WaveBits {
    const uint8_t* seg; uint32_t chunk; uint32_t cw;            // chunk: in dwords
    __device__ __forceinline__ void open(const uint8_t* s) { seg = s; chunk = 0; load(); }
    __device__ __forceinline__ void load()
    {
        uint32_t raw; __builtin_memcpy(&raw, seg + 4 * (size_t)(chunk + (threadIdx.x & 63)), 4);
        cw = __builtin_bswap32(raw);
    }
    __device__ __forceinline__ uint64_t at(uint32_t pos)        // pos: uniform, never before the chunk
    {
        uint32_t li = (pos >> 5) - chunk;
        if (__builtin_expect(li >= 63u, 0)) { chunk = pos >> 5; load(); li = 0; }
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)cw, (int)li), d1 = (uint32_t)__builtin_amdgcn_readlane((int)cw, (int)li + 1);
        return (((uint64_t)d0 << 32) | d1) << (pos & 31u);
    }
};

constexpr int kProgThreads = 64;

// A Huffman table for a wave on the scalar unit: the 256 entries of an 8-bit look-ahead ((length << 8) | symbol, 0 = longer
// code) live in four registers across the lanes and are fetched with v_readlane -- no LDS round trip on the chain; the rare
// longer codes take the canonical search through the table in LDS.
struct WaveHuff {
    uint32_t t0, t1, t2, t3;
    const DevHuff* lds;
    __device__ __forceinline__ void load(const DevHuff* h, int lane)
    {
        lds = h;
        auto entry = [&](int idx8) -> uint32_t { const uint32_t e = h->fast[idx8 << 1]; return (e >> 8) <= 8 ? e : 0u; };
        t0 = entry(lane); t1 = entry(64 + lane); t2 = entry(128 + lane); t3 = entry(192 + lane);
    }
    // the symbol at the top of the window w0, its length in `len`; -1: no such code
    __device__ __forceinline__ int decode(uint64_t w0, int& len) const
    {
        const int idx = (int)(w0 >> 56), l = idx & 63, j = idx >> 6;
        const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)t0, l), e1 = (uint32_t)__builtin_amdgcn_readlane((int)t1, l);
        const uint32_t e2 = (uint32_t)__builtin_amdgcn_readlane((int)t2, l), e3 = (uint32_t)__builtin_amdgcn_readlane((int)t3, l);
        const uint32_t e = j == 0 ? e0 : j == 1 ? e1 : j == 2 ? e2 : e3;
        if (e) { len = (int)(e >> 8); return (int)(e & 0xFF); }
        int32_t code = (int32_t)(w0 >> 55); len = 9;
        while (code > rfl(lds->maxcode[len])) { if (++len > 16) return -1; code = (int32_t)(w0 >> (64 - len)); }
        return rfl((int)lds->vals[(code + rfl(lds->delta[len])) & 0xFF]);
    }
};

// sampling factors are 1 or 2 (the marker walk rejects anything else): divisions by them are shifts
__device__ __forceinline__ int64_t prog_block_fast(int mcus_per_row, int nb, int hs, int vs, int off, int bx, int by)
{
    const int mx = bx >> (hs - 1), my = by >> (vs - 1);
    return (int64_t)((my * mcus_per_row + mx) * nb + off + (by & (vs - 1)) * hs + (bx & (hs - 1)));      // a frame has < 2^24 blocks (16384 x 16384, three components)
}

// Coefficients cross compute units while the launch runs (a scan reads what an earlier scan of the file wrote microseconds ago, from
// wherever that one runs): every store is write-through and every load reads past the caches (agent-scope relaxed atomics = sc1), a
// progress word is stored behind `s_waitcnt vmcnt(0)` -- the placement-independent hand-off of k_png_defilter_queue, no fences.
__device__ __forceinline__ int  co_load(const int16_t* p) { return (int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void co_store(int16_t* p, int v) { __hip_atomic_store(p, (int16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// An item's view of the launch's progress words: what it may touch, and what it tells the items behind it.
struct ProgSync {
    uint32_t* prog; uint32_t* st;
    int me, dep0, dep1, dep2, dep_pipe;
    uint32_t seen0, seen1, seen2;                               // the dependencies' progress as last read (it only grows)
    uint32_t waited;                                            // 100 MHz ticks spent waiting (GAMUT_HIP_TRACE)
    __device__ __forceinline__ void wait_word(const uint32_t* w, uint32_t need, uint32_t& seen)
    {
        if (seen >= need) return;
        const uint64_t t0 = wall_clock64();
        for (;;) {
            seen = rfl(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (seen >= need) { waited += (uint32_t)(wall_clock64() - t0); break; }
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > 2000000000ull) {          // 20 s of the 100 MHz clock: give up, say so, never hang
                if (threadIdx.x == 0) atomicOr(st, 16u);
                seen = 0xFFFFFFFFu;
                break;
            }
        }
    }
    __device__ __forceinline__ void wait_units(int upto)        // units [0, upto) of this item may be decoded
    {
        if (dep0 >= 0) wait_word(prog + dep0, (dep_pipe & 1) ? (uint32_t)upto : kProgDone, seen0);
        if (dep1 >= 0) wait_word(prog + dep1, (dep_pipe & 2) ? (uint32_t)upto : kProgDone, seen1);
        if (dep2 >= 0) wait_word(prog + dep2, (dep_pipe & 4) ? (uint32_t)upto : kProgDone, seen2);
    }
    __device__ __forceinline__ void publish(int units_done)     // the coefficients of units [0, units_done) are final as far as this scan goes
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) __hip_atomic_store(prog + me, (uint32_t)units_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

struct AcRefineArgs {
    const uint8_t* seg; int16_t* out; const DevHuff* huff;      // huff: the scan's table in LDS
    uint32_t limit_bit;
    int al, n_units, first_unit, mpr, nbm, ss, se, nbx, hs, vs, off;
};
__device__ __forceinline__ uint32_t prog_ac_refine(ProgSync sy, AcRefineArgs a, uint4* profile, uint32_t& end_pos);   // end_pos: set when the interval was decoded to its end

__device__ __forceinline__ void prog_scan_body(const ProgItem& it, const ProgImage& im, DevHuff* sh_huff, const DevHuff* huff_g,
                                               const uint8_t* blob, int16_t* coeffs, uint32_t* status, uint32_t* prog, const uint32_t* counters, int me, uint32_t& waited, uint4* profile)
{
    const int lane = threadIdx.x;
    const int kind = rfl(it.kind), ncomp = rfl(it.ncomp);
    if (kind != PROG_DC_REFINE)
        for (int i = 0; i < ncomp; ++i) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(huff_g + (i == 0 ? it.tab[0] : i == 1 ? it.tab[1] : it.tab[2]));
            uint32_t* dst = reinterpret_cast<uint32_t*>(&sh_huff[i]);
            for (int k = lane; k < (int)(sizeof(DevHuff) / 4); k += kProgThreads) dst[k] = src[k];
        }
    __syncthreads();
    const uint64_t seg_begin = rfl(it.begin);
    const uint8_t* seg = blob + seg_begin;
    const uint32_t seg_bytes = (uint32_t)(rfl(it.end) - seg_begin);
    const uint32_t limit_bit = seg_bytes * 8u + 64u * 8u;       // a decoder that runs past the padding is on a corrupt stream
    int16_t* out = coeffs + rfl(im.coeff_off);
    uint32_t* st = status + rfl(im.index);
    ProgSync sy{ prog, st, me, rfl(it.dep[0]), rfl(it.dep[1]), rfl(it.dep[2]), rfl(it.dep_pipe), 0u, 0u, 0u, 0u };
    struct Report { ProgSync& s; uint32_t& w; __device__ ~Report() { w = s.waited; } } report{ sy, waited };      // on every way out
    auto wait_units = [&](int upto) { sy.wait_units(upto); };
    auto publish = [&](int units_done) { sy.publish(units_done); };
    // an interval decoded to its end: may process_restart find the marker from where the reference's input stands?  (restart_leftover_bad)
    auto interval_done = [&](uint32_t used_bits) {
        const int tail = rfl(it.tail);
        if (tail >= 0 && lane == 0 && restart_leftover_bad(seg, seg_bytes, used_bits, tail)) atomicOr(st, kStatusBadRestart);
    };
    if (rfl(it.ctr) >= 0 && rfl(it.ctr_need) > 0) { uint32_t seen = 0; sy.wait_word(counters + rfl(it.ctr), (uint32_t)rfl(it.ctr_need), seen); }
    const int al = rfl(it.al), n_units = rfl(it.n_units), first_unit = rfl(it.first_unit);
    const int mpr = rfl(im.mcus_per_row), nbm = rfl(im.nb);        // scalars of their own: the structs are indexed by component elsewhere and live in scratch memory
    const uint32_t zag_reg = kZagDev[lane];                     // natural index of zig-zag position `lane`; readlane(zag_reg, k) for a uniform k
    // geometry of the scan's components (scan order)
    int c_[3], hs_[3], vs_[3], off_[3];
    #pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = rfl(i == 0 ? it.comp[0] : i == 1 ? it.comp[1] : it.comp[2]);
        c_[i] = c; hs_[i] = rfl(c == 0 ? im.hs[0] : c == 1 ? im.hs[1] : im.hs[2]); vs_[i] = rfl(c == 0 ? im.vs[0] : c == 1 ? im.vs[1] : im.vs[2]);
        off_[i] = c == 0 ? 0 : rfl(im.ny) + c - 1;
    }

    if (kind == PROG_DC_REFINE) {                               // decode_block_dc_refine :3321-3333: bit t of the segment belongs to block t
        int bpu = 0;
        for (int i = 0; i < ncomp; ++i) bpu += (i == 0 ? hs_[0] * vs_[0] : i == 1 ? hs_[1] * vs_[1] : hs_[2] * vs_[2]);
        if (ncomp == 1) bpu = 1;
        constexpr int kChunk = 1024;                            // units between two looks at the DC first scan's progress (a wait is bounded in time: it must
        for (int u0 = 0; u0 < n_units; u0 += kChunk) {           // be for a piece of that scan, not for all of a 16384 x 16384 frame's)
            const int u1 = min(u0 + kChunk, n_units);
            if (u0) publish(u0);
            wait_units(u1);                                     // the DC first scan of these units
            for (int t = u0 * bpu + lane; t < u1 * bpu; t += kProgThreads) {
                const int u = t / bpu, b = t - u * bpu;
                const uint32_t byte = (uint32_t)t >> 3;
                const uint32_t v = byte < seg_bytes ? seg[byte] : 0xFFu;                // past the data: ones (get_octet :683-696)
                if ((v >> (7 - (t & 7))) & 1u) {
                    int ci; const int64_t blk = prog_unit_block(im, it, first_unit + u, b, ci);
                    co_store(out + blk * 64, co_load(out + blk * 64) | (1 << al));
                }
            }
        }
        interval_done((uint32_t)n_units * (uint32_t)bpu);
        return;
    }

    WaveBits wb; wb.open(seg);
    uint32_t pos = 0;
    if (kind == PROG_DC_FIRST) {                                // decode_block_dc_first :3298-3319
        WaveHuff h0, h1, h2;
        h0.load(&sh_huff[0], lane); h1.load(&sh_huff[ncomp > 1 ? 1 : 0], lane); h2.load(&sh_huff[ncomp > 2 ? 2 : 0], lane);
        int pred0 = 0, pred1 = 0, pred2 = 0;
        // one block of a component of the scan (its table, its predictor: by reference -- a choice by index among the three would
        // keep all of them, and the bit reader with them, in scratch memory)
        auto dc_block = [&](const WaveHuff& h, int& pred, int64_t blk) -> bool {
            const uint64_t w0 = wb.at(pos);
            int len; const int s = h.decode(w0, len);
            if (s < 0) { if (lane == 0) atomicOr(st, 1u); return false; }
            const int n = s & 15;
            int v = n ? (int)((w0 << len) >> (64 - n)) : 0;
            if (n && v < (1 << (n - 1))) v += (int)(0xFFFFFFFFu << n) + 1;               // JPGD_HUFF_EXTEND :816-822
            pos += (uint32_t)(len + n);
            v += pred; pred = v;
            if (lane == 0) co_store(out + blk * 64, (int)((uint32_t)v << al));
            return true;
        };
        auto one = [&](int i, int64_t blk) -> bool { return i == 0 ? dc_block(h0, pred0, blk) : i == 1 ? dc_block(h1, pred1, blk) : dc_block(h2, pred2, blk); };
        if (ncomp == 1) {
            const int nbx = rfl(it.nbx);
            int by = first_unit / nbx, bx = first_unit - by * nbx;
            for (int u = 0; u < n_units; ++u) {
                if ((u & (kProgBatch - 1)) == 0) { if (u) publish(u); wait_units(min(u + kProgBatch, n_units)); }
                if (!one(0, prog_block_fast(mpr, nbm, hs_[0], vs_[0], off_[0], bx, by))) return;
                if (pos > limit_bit) { if (lane == 0) atomicOr(st, 4u); return; }
                if (++bx == nbx) { bx = 0; ++by; }
            }
        } else {
            int64_t base = (int64_t)first_unit * nbm;             // the MCU's first block: inside an MCU the scan order is the buffer's order
            for (int u = 0; u < n_units; ++u, base += nbm) {
                if ((u & (kProgBatch - 1)) == 0) { if (u) publish(u); wait_units(min(u + kProgBatch, n_units)); }
                for (int i = 0; i < ncomp; ++i) {
                    const int nblk = i == 0 ? hs_[0] * vs_[0] : i == 1 ? hs_[1] * vs_[1] : hs_[2] * vs_[2];
                    const int off = i == 0 ? off_[0] : i == 1 ? off_[1] : off_[2];
                    for (int b = 0; b < nblk; ++b) if (!one(i, base + off + b)) return;
                }
                if (pos > limit_bit) { if (lane == 0) atomicOr(st, 4u); return; }
            }
        }
        interval_done(pos);
        return;
    }

    // AC scans: one component, its blocks in raster order
    WaveHuff ac; ac.load(&sh_huff[0], lane);
    const int ss = rfl(it.ss), se = rfl(it.se), nbx = rfl(it.nbx);
    const int hs = hs_[0], vs = vs_[0], off = off_[0];
    int by = first_unit / nbx, bx = first_unit - by * nbx;
    int eobrun = 0;
    if (kind == PROG_AC_FIRST) {                                // decode_block_ac_first :3335-3398
        for (int u = 0; u < n_units; ++u) {
            if ((u & (kProgBatch - 1)) == 0) { if (u) publish(u); wait_units(min(u + kProgBatch, n_units)); }
            eobrun = rfl(eobrun);
            if (eobrun) --eobrun;
            else {
                int16_t* blk = out + prog_block_fast(mpr, nbm, hs, vs, off, bx, by) * 64;
                for (int k = ss; k <= se; ++k) {
                    k = rfl(k);
                    const uint64_t w0 = wb.at(pos);
                    int len; const int rs = ac.decode(w0, len);
                    if (rs < 0) { if (lane == 0) atomicOr(st, 1u); return; }
                    const int run = rs >> 4, size = rs & 15;
                    if (size) {
                        if ((k += run) > 63) { if (lane == 0) atomicOr(st, 2u); return; }
                        int v = (int)((w0 << len) >> (64 - size));
                        if (v < (1 << (size - 1))) v += (int)(0xFFFFFFFFu << size) + 1;
                        pos += (uint32_t)(len + size);
                        const int nat = __builtin_amdgcn_readlane((int)zag_reg, k);
                        if (lane == 0) co_store(blk + nat, (int)((uint32_t)v << al));
                    } else if (run == 15) {
                        pos += (uint32_t)len;
                        if ((k += 15) > 63) { if (lane == 0) atomicOr(st, 2u); return; }
                    } else {                                     // EOBn: this block and the next eobrun blocks end here
                        const int extra = run ? (int)((w0 << len) >> (64 - run)) : 0;
                        pos += (uint32_t)(len + run);
                        eobrun = (1 << run) + extra - 1;
                        break;
                    }
                }
                if (pos > limit_bit) { if (lane == 0) atomicOr(st, 4u); return; }
            }
            if (++bx == nbx) { bx = 0; ++by; }
        }
        interval_done(pos);
        return;
    }

    // AC refinement: a function of its own (its working set of scalar registers is the largest of the four kinds; inlined next to the
    // others the kernel spilled seventy of them)
    {
        const AcRefineArgs ra{ seg, out, &sh_huff[0], limit_bit, al, n_units, first_unit, mpr, nbm, ss, se, nbx, hs, vs, off };
        uint32_t end_pos = 0xFFFFFFFFu;
        sy.waited = prog_ac_refine(sy, ra, profile, end_pos);
        if (end_pos != 0xFFFFFFFFu) interval_done(end_pos);
    }
}

// decode_block_ac_refine :3400-3518, a wave per block: lane k holds the coefficient at zig-zag position k.
// The scalar unit walks the symbols; what it needs per symbol is short:
//   * the bits: a 128-bit window (w0 | w1) kept in scalar registers and SHIFTED by what a symbol used, a dword appended from the
//     wave's 256-byte piece of the stream (one v_readlane) whenever fewer than 97 bits are left -- not rebuilt from the bit position;
//   * the code: a 6-bit look-ahead (one v_readlane; refinement alphabets are a few (run, 0 / 1) symbols and EOBn), the 8-bit tables
//     and the canonical search behind it;
//   * where the run ends: per block, ONE permutation puts the positions without history (in band, ascending) into lanes 0 .. Z-1
//     (pz); a symbol's stop is v_readlane(pz, zeros passed so far + run);
//   * how many correction bits it passes: a population count of the history mask between the old and the new position.
// The correction bits themselves are not on that chain: every lane with history notes the BIT POSITION of its correction bit (the
// position behind the symbol + its rank among the lanes with history, known per block), fetches that one byte of the stream when
// the block is done, and the update + store of a block happens while the next block is being walked.
#ifndef PROG_WALK_ASM
#define PROG_WALK_ASM 1
#endif
#ifdef GAMUT_PROG_PROFILE
#define PROG_T(var) const uint64_t var = __builtin_amdgcn_s_memtime()
#define PROG_ACC(dst, a, b) dst += (uint32_t)((b) - (a))
#else
#define PROG_T(var)
#define PROG_ACC(dst, a, b)
#endif
__device__ __forceinline__ uint32_t prog_ac_refine(ProgSync sy, AcRefineArgs a, uint4* profile, uint32_t& end_pos)   // end_pos: set when the interval was decoded to its end
{
    uint32_t pf_setup = 0, pf_walk = 0, pf_tail = 0, pf_syms = 0;     // GAMUT_PROG_PROFILE: shader clocks per phase, symbols
    (void)pf_setup; (void)pf_walk; (void)pf_tail; (void)pf_syms;
    const int lane = threadIdx.x;
    const uint8_t* const seg = a.seg; int16_t* const out = a.out; uint32_t* const st = sy.st;
    const uint32_t limit_bit = a.limit_bit;
    const int al = rfl(a.al), n_units = rfl(a.n_units), first_unit = rfl(a.first_unit), mpr = rfl(a.mpr), nbm = rfl(a.nbm);
    const int ss = rfl(a.ss), se = rfl(a.se), nbx = rfl(a.nbx), hs = rfl(a.hs), vs = rfl(a.vs), off = rfl(a.off);
    const uint32_t zag_reg = kZagDev[lane];
    WaveHuff ac; ac.load(a.huff, lane);
    const DevHuff* const sh_huff = a.huff;
    int by = first_unit / nbx, bx = first_unit - by * nbx;
    int eobrun = 0;
    uint32_t pos = 0;
    auto wait_units = [&](int upto) { sy.wait_units(upto); };
    auto publish = [&](int units_done) { sy.publish(units_done); };
    {
        const int nat = (int)zag_reg;
        const uint64_t band = (se >= 63 ? ~0ull : (1ull << (se + 1)) - 1) & ~((1ull << ss) - 1);
        const uint64_t m_outside = se >= 63 ? 0ull : ~0ull << (se + 1);      // "positions from the stop on" of a run that leaves the band: none of the band's
        const int plus = 1 << al, minus = (int)(0xFFFFFFFFu << al);
        // Bit reader.  The scalar chain only ever needs the next code and, behind an EOBn code, up to 14 more bits: 30 bits at `pos`.
        // Sign and correction bits are fetched by the lanes they belong to, from memory, when the block is done.  So there is no window
        // to maintain: two v_readlane out of the wave's 256-byte piece of the stream (cwv: lane L holds big-endian dword cb + L) and one
        // 64-bit shift give 33+ valid bits at any position.  A dependent scalar instruction of a lone wave takes ~8 clocks, a
        // v_readlane round trip ~20, a taken branch ~30 (tools/microbench/chain_latency.hip): the loop below is counted in instructions.
        WaveBits rb; rb.open(seg);
        auto bits_at = [&](uint32_t at) -> uint64_t { return rb.at(at); };
        // Look-ahead of 6 bits (refinement alphabets are a few (run, 0 / 1) symbols and EOBn), everything the walk needs of a symbol in
        // one word: [1:0] 1 = coefficient or ZRL, 2 = EOBn, 3 = not a refinement symbol, 0 = longer code; [11:4] bits used (code + sign,
        // or code + the EOB run's extra bits); [15:12] run; [16] size; [25:20] code length.
        auto pack = [&](int sym, int len) -> uint32_t {
            const int run = sym >> 4, size = sym & 15;
            const uint32_t kind = size > 1 ? 3u : (size == 0 && run != 15) ? 2u : 1u;
            const int used = kind == 2u ? len + run : len + size;
            return kind | (uint32_t)used << 4 | (uint32_t)run << 12 | (uint32_t)(size & 1) << 16 | (uint32_t)len << 20;
        };
        uint32_t t6;
        { const uint32_t e = sh_huff[0].fast[(lane << 3)]; t6 = (e >> 8) <= 6 && e ? pack((int)(e & 0xFF), (int)(e >> 8)) : 0u; }      // fast[]: 9-bit look-ahead, (length << 8) | symbol
        // the blocks are fetched three blocks ahead of their turn (they come from memory: other scans wrote them, possibly microseconds ago)
        int fx = bx, fy = by;                                   // the next block to fetch
        // (the values stay 16-bit until their block's turn: a conversion right behind the load would wait for it there)
        int16_t *p0 = out, *p1 = out, *p2 = out; int16_t c0 = 0, c1 = 0, c2 = 0;
        auto fetch = [&](int16_t*& p, int16_t& c, bool live) {
            if (live) {
                p = out + prog_block_fast(mpr, nbm, hs, vs, off, fx, fy) * 64;
                c = __hip_atomic_load(p + nat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (++fx == nbx) { fx = 0; ++fy; }
            }
        };
        // the block whose bits are on their way
        int16_t* pend_ptr = out; int pend_orig = 0; uint32_t pend_byte = 0, pend_bit = 0xFFFFFFFFu;
        auto finish_pending = [&]() {                            // (selects, one branch for the store: per block, but on the chain all the same)
            const bool have = pend_bit != 0xFFFFFFFFu;
            const bool fresh = have && (pend_bit >> 31);         // the lane a symbol's run ended at: it is SET (decode_block_ac_refine :3485-3488 assigns p[k] for any k < 64 --
            const bool one = (pend_byte >> (7u - (pend_bit & 7u))) & 1u;      // behind the band too, where a damaged stream's run can end on a coefficient of another scan)
            const int step = pend_orig >= 0 ? plus : minus;
            const int grown = (one && (pend_orig & plus) == 0) ? (int)(int16_t)(pend_orig + step) : pend_orig;   // history: one more bit of magnitude
            int coef = fresh ? (one ? plus : minus) : grown;                                                     // a new coefficient: its sign
            coef = have ? coef : pend_orig;
            if (coef != pend_orig) co_store(pend_ptr + nat, coef);
        };
        bool ended = false;
        for (int ub = 0; ub < n_units; ub += kProgBatch) {      // a batch of units: everything the scans before left in them is there
            const int nu = min(kProgBatch, n_units - ub);
            if (ub) { finish_pending(); pend_bit = 0xFFFFFFFFu; publish(ub); }
            wait_units(ub + nu);
            fetch(p0, c0, nu > 0); fetch(p1, c1, nu > 1); fetch(p2, c2, nu > 2);
            // one block; the slot it came from is refilled with the block three further on (three copies of this body, one per slot: moving an
            // in-flight value from slot to slot would wait for it).  -> false: the scan is over (an impossible code, or past the data)
            auto one_block = [&](int16_t*& slot_p, int16_t& slot_c, int u) -> bool {
                PROG_T(t_a);
                const int orig = slot_c; int16_t* const cur = slot_p;
                fetch(slot_p, slot_c, u + 3 < nu);
                uint32_t mybit = 0xFFFFFFFFu;                    // the bit of the stream this lane wants: its correction bit (history) or its sign (new)
                const uint64_t nz = __ballot(orig != 0) & band;
                eobrun = rfl(eobrun);
                if (eobrun > 0 && nz == 0) { --eobrun; return true; }       // inside an EOB run, no history: not a bit of the stream belongs to this block
                const uint64_t zeros = ~nz & band;
                const int rk = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(nz >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nz, 0u));      // lanes with history below this one
                const bool has_hist = (nz >> lane) & 1ull;
                const uint64_t hist_mask = __ballot(has_hist);   // (= nz: as an exec-style mask for the walk's own instructions)
                uint32_t nc = 0;                                 // lanes with history already passed
                uint64_t nzr = nz;                              // ... and those still ahead
                int k = ss;
                PROG_T(t_b);
                if (eobrun == 0) {
                    const int zr = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(zeros >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)zeros, 0u));
                    const int Z = __popcll(zeros);
                    const int dest = ((zeros >> lane) & 1ull) ? zr : 63 - (lane - zr);
                    const int pz = __builtin_amdgcn_ds_permute(dest << 2, lane);                 // lane r: the r-th position without history
                    uint32_t zc = 0;                            // positions without history already passed
                    int way_out = 0;                             // 1: an EOBn code ended the block's walk, 2: not a code of a refinement scan
                    if (k <= se) do {                            // ONE way out of the loop (the condition at its foot): with breaks and jumps the compiler builds a
#ifdef GAMUT_PROG_PROFILE                                         // state machine of flags around every iteration, ~13 of 60 instructions per symbol
                        ++pf_syms;
#endif
                        const uint64_t w = bits_at(pos);
                        uint32_t ent = (uint32_t)__builtin_amdgcn_readlane((int)t6, (int)(w >> 58));
                        if (__builtin_expect((ent & 3u) != 1u, 0)) {
                            if (ent == 0) { int len; const int sym = ac.decode(w, len); ent = sym < 0 ? 3u : pack(sym, len); }
                            if ((ent & 3u) == 3u) { way_out = 2; k = 1 << 20; continue; }
                            if ((ent & 3u) == 2u) {              // EOBn: this block's band ends here, and that of the next eobrun - 1 blocks
                                const int len = (int)(ent >> 20), run = (int)((ent >> 12) & 15u);
                                eobrun = (1 << run) + (run ? (int)((w << len) >> (64 - run)) : 0);
                                pos += (ent >> 4) & 0xFFu;
                                way_out = 1; k = 1 << 20; continue;
                            }
                        }
#if PROG_WALK_ASM
                        {   // The step of a coefficient / ZRL symbol, written out: what the compiler makes of the C++ below is these instructions plus
                            // ~13 that materialise and test flags.  On from k over `run` positions without history to the one that ends the run (stop =
                            // lane R of pz, or se + 1 past the last of them); what has history on the way (corr) is corrected: those lanes note where
                            // their bit is (behind the symbol, in position order); a new coefficient's lane notes its sign bit (P - 1).
                            uint32_t stop, t0, t1, t2, t3; uint64_t m, corr; uint32_t v0, v1;
                            asm volatile(
                                "s_bfe_u32 %[t0], %[ent], 0x4000c\n"            // run
                                "s_bfe_u32 %[t1], %[ent], 0x80004\n"            // bits used
                                "s_add_u32 %[t0], %[zc], %[t0]\n"              // R = zc + run
                                "s_add_u32 %[zc], %[t0], 1\n"
                                "s_and_b32 %[t2], %[t0], 63\n"
                                "v_readlane_b32 %[stop], %[pz], %[t2]\n"
                                "s_add_u32 %[t1], %[pos], %[t1]\n"             // P = pos + used
                                "s_lshl_b64 %[m], -1, %[stop]\n"               // positions from stop on (a lane of pz: <= 63).  (Sets SCC: the comparison comes
                                "s_cmp_lt_u32 %[t0], %[Z]\n"                   // AFTER it -- inside?  Until round 4's fuzz_mixed_gpu.py run the shift stood behind
                                "s_cselect_b32 %[stop], %[stop], %[se1]\n"     // the comparison and the select below saw the shift's SCC: a run that left a band
                                "s_cselect_b64 %[m], %[m], %[mo]\n"            // ending at 63 (stop = 64, a shift by 0) corrected nothing of what was left.)
                                "s_andn2_b64 %[corr], %[nzr], %[m]\n"
                                "s_and_b64 %[nzr], %[nzr], %[m]\n"
                                "s_bcnt1_i32_b64 %[t2], %[corr]\n"             // c
                                "v_subrev_u32 %[v0], %[nc], %[rk]\n"           // t = rk - nc
                                "v_cmp_gt_u32 vcc, %[t2], %[v0]\n"             // t < c
                                "s_add_u32 %[pos], %[t1], %[t2]\n"             // pos = P + c
                                "s_and_b64 vcc, vcc, %[hist]\n"
                                "v_add_u32 %[v0], %[t1], %[v0]\n"              // P + t
                                "s_bitcmp1_b32 %[ent], 16\n"                   // a new coefficient?
                                "v_cndmask_b32 %[mybit], %[mybit], %[v0], vcc\n"
                                "s_cselect_b32 %[t3], %[stop], 64\n"
                                "s_add_u32 %[t1], %[t1], -1\n"                 // P - 1
                                "s_bitset1_b32 %[t1], 31\n"                    // ... of a NEW coefficient (whatever stood there: the stop may lie behind the band)
                                "v_cmp_eq_u32 vcc, %[t3], %[lane]\n"
                                "v_mov_b32 %[v1], %[t1]\n"
                                "s_add_u32 %[nc], %[nc], %[t2]\n"
                                "v_cndmask_b32 %[mybit], %[mybit], %[v1], vcc\n"
                                : [stop] "=&s"(stop), [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [t3] "=&s"(t3), [m] "=&s"(m), [corr] "=&s"(corr), [v0] "=&v"(v0), [v1] "=&v"(v1),
                                  [zc] "+s"(zc), [pos] "+s"(pos), [nzr] "+s"(nzr), [nc] "+s"(nc), [mybit] "+v"(mybit)
                                : [ent] "s"(ent), [pz] "v"(pz), [Z] "s"(Z), [se1] "s"(se + 1), [mo] "s"(m_outside), [rk] "v"(rk), [hist] "s"(hist_mask), [lane] "v"(lane)
                                : "scc", "vcc");
                            k = (int)stop + 1;
                        }
#else
                        {
                        const int used = (int)((ent >> 4) & 0xFFu), run = (int)((ent >> 12) & 15u);
                        const bool coefficient = (ent >> 16) & 1u;       // a new coefficient at the stop (its sign follows the code); else ZRL
                        // on from k over `run` positions without history to the one that ends the run; what has history on the way is corrected
                        const int R = (int)zc + run;
                        const int stop_r = __builtin_amdgcn_readlane(pz, R & 63);
                        const bool inside = R < Z;
                        const int stop = inside ? stop_r : se + 1;
                        zc = (uint32_t)R + 1u;
                        const uint64_t below = inside ? (1ull << stop) - 1 : ~0ull;          // stop <= 63 here
                        const uint64_t corr = nzr & below;
                        nzr &= ~below;
                        const int c = __popcll(corr);
                        const uint32_t P = pos + (uint32_t)used;                             // the correction bits follow the symbol, in position order
                        const uint32_t t = (uint32_t)rk - nc;
                        if (has_hist && t < (uint32_t)c) mybit = P + t;
                        if (lane == (coefficient ? stop : 64)) mybit = (P - 1u) | 0x80000000u;  // stop <= 64: no lane if the walk ran off the block; bit 31: a NEW coefficient
                        nc += (uint32_t)c;
                        pos = P + (uint32_t)c;
                        k = stop + 1;
                        }
#endif
                    } while (k <= se);
                    if (way_out == 2) { if (lane == 0) atomicOr(st, 1u); finish_pending(); return false; }
                }
                PROG_T(t_c);
                eobrun = rfl(eobrun);
                if (eobrun > 0) {                                // the rest of the block: correction bits only
                    const int c = __popcll(nzr);
                    const uint32_t t = (uint32_t)rk - nc;
                    if (has_hist && t < (uint32_t)c) mybit = pos + t;
                    pos += (uint32_t)c;
                    --eobrun;
                }
                // the block before this one: its correction bits have arrived; this block's are sent for
                finish_pending();
                pend_ptr = cur; pend_orig = orig; pend_bit = mybit; pend_byte = 0;
                if (mybit != 0xFFFFFFFFu) pend_byte = seg[(mybit & 0x7FFFFFFFu) >> 3];
                if (pos > limit_bit) { if (lane == 0) atomicOr(st, 4u); finish_pending(); return false; }
                PROG_T(t_d);
                PROG_ACC(pf_setup, t_a, t_b); PROG_ACC(pf_walk, t_b, t_c); PROG_ACC(pf_tail, t_c, t_d);
                return true;
            };
            bool going = true;
            for (int u = 0; u < nu && going; u += 3) {
                going = one_block(p0, c0, u);
                if (going && u + 1 < nu) going = one_block(p1, c1, u + 1);
                if (going && u + 2 < nu) going = one_block(p2, c2, u + 2);
            }
            if (!going) { ended = true; break; }
        }
        if (!ended) { finish_pending(); end_pos = pos; }
    }
#ifdef GAMUT_PROG_PROFILE
    if (profile && lane == 0) *profile = make_uint4(pf_setup, pf_walk, pf_tail, pf_syms);
#endif
    return sy.waited;
}

// One workgroup (a wave) per item, items taken in list order through a ticket: whatever an item waits for has a lower ticket and
// is therefore running or done -- no deadlock however few wave slots the launch gets.
__global__ __launch_bounds__(kProgThreads) void k_prog_scan(const ProgItem* items, const ProgImage* images, const DevHuff* huff_g,
                                                            const uint8_t* blob, int16_t* coeffs, uint32_t* status, uint32_t* prog, uint32_t* ticket, uint4* times)
{
    __shared__ DevHuff sh_huff[3];
    int me = 0;
    if (threadIdx.x == 0) me = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    me = rfl(me);
    const ProgItem& it = items[me];                             // read where they lie: copies of the structs (their arrays indexed by component) would live in scratch memory, per lane
    const ProgImage& im = images[rfl(it.image)];
    switch (rfl(it.prio)) {                                    // the instruction takes an immediate
        case 3: __builtin_amdgcn_s_setprio(3); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        default: break;
    }
    const uint64_t t_in = wall_clock64();
    uint32_t waited = 0;
    prog_scan_body(it, im, sh_huff, huff_g, blob, coeffs, status, prog, ticket + 1, me, waited, times ? times + gridDim.x + me : (uint4*)nullptr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
        __hip_atomic_store(prog + me, kProgDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (rfl(it.ctr) >= 0) __hip_atomic_fetch_add(ticket + 1 + rfl(it.ctr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (times) { times[me] = make_uint4((uint32_t)t_in, (uint32_t)(t_in >> 32), (uint32_t)(wall_clock64() - t_in), waited); }   // GAMUT_HIP_TRACE: when it ran, how long, how much of it waiting
    }
}

// load_next_row :2259-2333 for every block of the listed images: max_zag = last non-zero zig-zag position + 1, coefficients
// times the quantisation factor of their position.  Eight lanes per block (16 bytes each).
__global__ __launch_bounds__(256) void k_prog_finalize(const ProgImage* images, int n_images, const int16_t* qnat /* [table][64], natural order */,
                                                       int16_t* coeffs, uint8_t* max_zag)
{
    __shared__ uint8_t sh_izag[64];
    if (threadIdx.x < 64) sh_izag[kZagDev[threadIdx.x]] = (uint8_t)threadIdx.x;
    __syncthreads();
    const ProgImage im = images[blockIdx.y];
    const int part = threadIdx.x & 7;
    for (int64_t b = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); b < im.n_blocks; b += (int64_t)gridDim.x * 32) {
        const int w = (int)(b % im.nb), c = w < im.ny ? 0 : w - im.ny + 1;
        const int16_t* q = qnat + (c == 0 ? im.quant[0] : c == 1 ? im.quant[1] : im.quant[2]) * 64 + part * 8;
        int16_t* p = coeffs + im.coeff_off + b * 64 + part * 8;
        union { uint4 v; int16_t s[8]; } x, f;
        x.v = *reinterpret_cast<const uint4*>(p);
        f.v = *reinterpret_cast<const uint4*>(q);
        int last = 0;
        #pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (x.s[i]) { const int z = sh_izag[part * 8 + i]; last = z > last ? z : last; }
            x.s[i] = (int16_t)((uint32_t)(int32_t)x.s[i] * (uint32_t)(int32_t)f.s[i]);
        }
        *reinterpret_cast<uint4*>(p) = x.v;
        last = max(last, __shfl_xor(last, 1)); last = max(last, __shfl_xor(last, 2)); last = max(last, __shfl_xor(last, 4));
        if (part == 0) max_zag[im.zag_off + b] = (uint8_t)(last + 1);
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------

// The highest zig-zag position a scan may WRITE: an AC first scan stores a coefficient wherever its runs take it up to 63 (decode_block_ac_first
// :3361-3380 checks k against 63, not against Se), an AC refinement scan sets the position its run ends at if that is < 64 (:3485-3488), i.e.
// up to Se + 1.  Well-formed files stay inside [Ss, Se]; a damaged one that does not must still come out as the reference decodes it -- scan
// after scan -- so the order of two scans of a component is kept wherever their REACH overlaps, not only their bands.  (Round 4, found by the
// fuzzer: Y 1-5 and Y 6-63, "independent", ran side by side; a run of the damaged first one ended at position 7, and its value landed on
// top of the second one's.)  The scans follow each other block by block, so the extra order costs next to nothing.
inline int prog_scan_reach(int kind, int se) { return kind == PROG_AC_FIRST ? 63 : kind == PROG_AC_REFINE ? (se < 63 ? se + 1 : 63) : se; }

struct ProgScanPrep {
    Scan sc; int kind = 0, level = 0, restart_interval = 0, units = 0, nbx = 0, nby = 0;
    int tab[3] = { 0, 0, 0 };            // file-local table indices (scan order)
    size_t begin = 0, end = 0;           // entropy-coded bytes of the scan in the file
};
struct ProgPrep {
    int rc = GAMUT_HIP_OK; char msg[200] = { 0 };
    int comps = 0, nb = 0, ny = 0, hs[3] = { 1, 1, 1 }, vs[3] = { 1, 1, 1 };
    QuantTab quant[3];                   // natural order (factor of the coefficient at natural index n)
    std::vector<ProgScanPrep> scans;
    std::vector<DevHuff> tabs;
    size_t cap = 0, used = 0;
    std::vector<ProgItem> items;         // begin / end relative to the file's slot in the blob; tab = file-local
};

// end of the entropy-coded data that starts at `q`: the first marker that is neither a stuffed FF00, a fill byte nor RSTn
inline size_t scan_data_end(const uint8_t* base, size_t q, size_t n)
{
    while (q < n) {
        const uint8_t* hit = (const uint8_t*)memchr(base + q, 0xFF, n - q);
        if (!hit || hit + 1 >= base + n) return n;
        q = (size_t)(hit - base);
        const uint8_t m = hit[1];
        if (m == 0x00 || (m >= 0xD0 && m <= 0xD7)) { q += 2; continue; }
        if (m == 0xFF) { q += 1; continue; }
        return q;
    }
    return n;
}

// marker walk of a whole progressive file: every scan with its tables as they stand when it begins (init_progressive :3585-3664)
void prog_prepare(int i, const uint8_t* base, size_t n, gamut_hip_jpeg_frame& f, ProgPrep& out, Parser& P)
{
    auto bad = [&](const char* why) { out.rc = GAMUT_HIP_ERR_DECODE; snprintf(out.msg, sizeof(out.msg), "image %d: jpeg: %s", i, why); fail(&f, why); };
    P = Parser();
    int marker = open_frame(P, base, n, &f);
    if (marker < 0) { out.rc = GAMUT_HIP_ERR_DECODE; snprintf(out.msg, sizeof(out.msg), "image %d: %s", i, last_error_buf()); return; }
    if (marker != 0xDA) { out.rc = kHostRedo; return; }        // a frame without a scan (every coefficient 0): nothing for the kernels, the host feeder writes the zeros
    out.comps = f.comps; out.nb = f.blocks_per_mcu; out.ny = f.comps == 1 ? 1 : P.hs[0] * P.vs[0];
    int max_h = 1, max_v = 1;
    for (int c = 0; c < f.comps; ++c) { out.hs[c] = P.hs[c]; out.vs[c] = P.vs[c]; if (P.hs[c] > max_h) max_h = P.hs[c]; if (P.vs[c] > max_v) max_v = P.vs[c]; }
    while (marker == 0xDA) {
        ProgScanPrep s; s.sc = P.scan;
        const Scan& sc = s.sc;
        const bool dc_scan = sc.ss == 0, refine = sc.ah != 0;
        if (scan_lists_a_component_twice(sc)) return bad("the scan lists a component twice");   // decode_scan's walk leaves the component's plane (coeff_buf_getp :3293)
        if (!scan_tables_ok(P, &f)) { out.rc = GAMUT_HIP_ERR_DECODE; snprintf(out.msg, sizeof(out.msg), "image %d: %s", i, last_error_buf()); return; }
        if (sc.ss > sc.se || sc.se > 63 || (dc_scan && sc.se != 0)) return bad("bad SOS spectral selection");
        if (!dc_scan && sc.ncomp != 1) return bad("AC scans can only contain one component");
        if (refine && sc.al != sc.ah - 1) return bad("bad SOS successive approximation");
        // (Al is a nibble: the standard stops at 13, read_sos_marker :1466-1540 does not, and every use of it here wraps in 16 bits as the reference's
        //  jpgd_block_t arithmetic does -- tests/golden/jpeg_fuzz/prog_al14_r04.jpg, found by tools/fuzz_mixed_gpu.py when this line still rejected Al > 13)
        for (int k = 0; k < sc.ncomp; ++k) {
            const int c = sc.comp[k];
            if (!(dc_scan && refine)) {
                // not for the kernels (kHostRedo): a table that carries a symbol value twice (the two-argument huff_decode takes code_size[symbol] bits for a
                // short code word: the LAST length that value was given), a DC category above 15
                const HuffTable& ht = P.huff[dc_scan ? P.td[c] : P.ta[c]];
                if (ht.twice || (dc_scan && ht.dc_limit > 15)) { out.rc = kHostRedo; return; }
                DevHuff d; to_dev_huff(ht, d);
                s.tab[k] = intern(out.tabs, d);
            }
        }
        s.kind = dc_scan ? (refine ? PROG_DC_REFINE : PROG_DC_FIRST) : (refine ? PROG_AC_REFINE : PROG_AC_FIRST);
        s.restart_interval = P.restart_interval;
        if (sc.ncomp == 1) {                                   // calc_mcu_block_order :3052-3066: the component's own blocks
            const int c = sc.comp[0];
            s.nbx = ((f.width  * P.hs[c] + max_h - 1) / max_h + 7) / 8;
            s.nby = ((f.height * P.vs[c] + max_v - 1) / max_v + 7) / 8;
            if (s.nbx > f.mcus_per_row * P.hs[c] || s.nby > f.mcus_per_col * P.vs[c]) return bad("decode error in a progressive scan");
            s.units = s.nbx * s.nby;
        } else s.units = f.mcus_per_row * f.mcus_per_col;
        // level: behind every earlier scan that touches one of its coefficients
        for (const ProgScanPrep& e : out.scans) {
            bool shares = false;
            for (int a = 0; a < sc.ncomp; ++a) for (int b = 0; b < e.sc.ncomp; ++b) shares = shares || sc.comp[a] == e.sc.comp[b];
            if (shares && sc.ss <= prog_scan_reach(e.kind, e.sc.se) && e.sc.ss <= prog_scan_reach(s.kind, sc.se) && e.level + 1 > s.level) s.level = e.level + 1;
        }
        if (P.pos >= n) { out.rc = kHostRedo; return; }            // a scan header that ends in the padding behind the file
        s.begin = P.pos; s.end = scan_data_end(base, P.pos, n);
        out.scans.push_back(s);
        if (out.scans.size() > 256) return bad("too many scans");
        P.pos = s.end;
        marker = next_scan(P, &f, Walk::kNextScan);
        if (marker < 0) { out.rc = GAMUT_HIP_ERR_DECODE; snprintf(out.msg, sizeof(out.msg), "image %d: %s", i, last_error_buf()); return; }
    }
    // de-quantisation happens after the last scan, with the tables as they stand then (load_next_row :2306-2323) -- of every component of the frame
    for (int c = 0; c < f.comps; ++c) if (!P.quant_def[P.tq[c]]) return bad("undefined quant table");
    for (int c = 0; c < f.comps; ++c) for (int k = 0; k < 64; ++k) out.quant[c].q[kZag[k]] = P.quant[P.tq[c]][k];
    out.cap = 0;
    size_t all_segs = 0;
    for (const ProgScanPrep& s : out.scans) {
        const size_t segs = s.restart_interval ? (size_t)(s.units / s.restart_interval + 1) : 1;
        if ((all_segs += segs) > ((size_t)1 << 22)) return bad("too many restart intervals");      // 256 scans x one interval per block of a 16384 x 16384 frame: gigabytes of segment records
        out.cap += (s.end - s.begin) + (64 + 4) * segs + 16;    // per segment: 64 bytes of padding, then up to the next dword
    }
}

// the scans of one file, unstuffed and cut at the restart markers, into the file's slot of the pinned upload image
void prog_unstuff(int i, const uint8_t* base, gamut_hip_jpeg_frame& f, ProgPrep& out, uint8_t* dst)
{
    size_t w = 0;
    bool bad = false;
    for (size_t si = 0; si < out.scans.size(); ++si) {
        const ProgScanPrep& s = out.scans[si];
        const int total = s.units, ri = s.restart_interval;
        int next_unit = 0, expect = 0, n_seg = 0;
        size_t q = s.begin, copy_from = s.begin, seg_begin = w;
        const size_t n = s.end;
        bool copying = true;
        auto flush = [&](size_t upto) {
            if (copying && upto > copy_from) {
                const size_t k = upto - copy_from;
                if (w + k > out.cap) { bad = true; return; }
                memcpy(dst + w, base + copy_from, k); w += k;
            }
        };
        int fill = 0;                                          // 0xFF bytes in front of the marker under way, besides its own
        auto close_segment = [&](int nu, int tail) {
            if (w + 64 + 16 > out.cap) { bad = true; return; }
            ProgItem it; memset(&it, 0, sizeof(it));
            it.image = i; it.kind = s.kind; it.ncomp = s.sc.ncomp; it.ss = s.sc.ss; it.se = s.sc.se; it.al = s.sc.al;
            for (int k = 0; k < s.sc.ncomp; ++k) { it.comp[k] = s.sc.comp[k]; it.tab[k] = s.tab[k]; }
            it.first_unit = next_unit; it.n_units = nu; it.nbx = s.nbx > 0 ? s.nbx : 1; it.level = s.level;
            it.begin = seg_begin; it.end = w;
            it.scan = (int32_t)si; it.seg = n_seg++; it.tail = tail;
            it.dep[0] = it.dep[1] = it.dep[2] = -1;
            out.items.push_back(it); next_unit += nu;
            memset(dst + w, 0xFF, 64); w += 64;
            w = (w + 3) & ~(size_t)3;                          // segments start on dword boundaries (the wave reader loads dwords)
            seg_begin = w;
        };
        while (!bad) {
            const uint8_t* hit = q < n ? (const uint8_t*)memchr(base + q, 0xFF, n - q) : nullptr;
            if (!hit || hit + 1 >= base + n) { flush(n); q = n; break; }
            const uint8_t m = hit[1];
            q = (size_t)(hit - base);
            if (m == 0x00) {
                if (!copying) { bad = true; break; }                                    // FF .. FF 00: the input stopped at the first FF (see unstuff_file)
                flush(q + 1); copy_from = q + 2; q += 2; continue;                      // stuffed 0xFF: keep the FF, drop the 00
            }
            flush(q); copying = false;
            if (m == 0xFF) { q += 1; fill = std::min(fill + 1, 4096); continue; }
            if (m >= 0xD0 && m <= 0xD7 && ri && next_unit + ri < total) {
                if (m != 0xD0 + expect) { bad = true; break; }
                close_segment(ri, fill); fill = 0;
                expect = (expect + 1) & 7; q += 2; copy_from = q; copying = true;
                continue;
            }
            break;
        }
        if (!bad && q < n) bad = true;                         // an RSTn the scan has no use for: the reference's input stops THERE, prog_prepare looked for the next scan behind it
        if (!bad && next_unit < total) {
            if (ri && total - next_unit > ri) bad = true;      // a restart marker is missing
            else close_segment(total - next_unit, -1);
        }
        if (bad) break;
    }
    out.used = w;
    (void)f;
    if (bad) { out.items.clear(); out.used = 0; out.rc = kHostRedo; }       // what the reference makes of such a restart structure is the host feeder's to say
}

// Decodes the progressive files data[idx[0..n)] of a batch (indices into the caller's arrays) into the same buffers the
// baseline files of the batch went to.  host_status[i] / info[i] are filled for those files; returns the status of the
// lowest-numbered failing one (message in `first_msg`).
int progressive_decode_device(const uint8_t* const* data, const size_t* len, const std::vector<int>& idx,
                              const int64_t* coeff_offset, const int64_t* zag_offset,
                              int16_t* d_coeffs, uint8_t* d_max_zag, uint32_t* d_status,
                              gamut_hip_jpeg_frame* info, int* host_status, hipStream_t stream, int* first_index, char* first_msg, size_t msg_cap)
{
    const bool trace = getenv("GAMUT_HIP_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    const int n = (int)idx.size();
    *first_index = -1;
    if (n == 0) return GAMUT_HIP_OK;
    int workers = host_threads();
    workers = workers < 1 ? 1 : workers > 16 ? 16 : workers;
    if (workers > n) workers = n;
    // A scan is a chain: the GPU decodes a 1080p file's longest one in 63 ms (round 4; whatever the batch, up to a few hundred files) where a
    // host core needs 20 ms for the whole file, and wins by decoding every scan of every file at the same time.  Below five files per host
    // thread the host feeder (Progressive::run on the thread pool) plus an upload of the coefficients is the faster way to the same buffers
    // (profiles/r04_prog_threshold.txt, files -> rgba8: 16 threads 64 files 54 ms on the host / 66 ms on the GPU, 96 files 76 / 68 ms,
    // 192 files 185 / 71 ms; 2 threads 16 files 70 / 65 ms).  GAMUT_HIP_JPEG_PROGRESSIVE = host / device forces either (tests, measurements).
    const char* how = getenv("GAMUT_HIP_JPEG_PROGRESSIVE");
    const bool on_host = how && !strcmp(how, "host") ? true : how && !strcmp(how, "device") ? false : n < 5 * workers;
    // the host feeder's share: every file (on_host), or afterwards the files the kernels flagged / were not given (kHostRedo)
    auto on_the_host = [&](const std::vector<int>& which, std::vector<int>& rcs, std::vector<std::string>& msgs) -> int {
        return host_redo(data, len, which, info, deliver_dense(coeff_offset, zag_offset, d_coeffs, d_max_zag, d_status, stream), rcs, msgs);
    };
    if (on_host) {
        std::vector<int> rcs; std::vector<std::string> msgs;
        if (int rc = on_the_host(idx, rcs, msgs)) { *first_index = -1; return rc; }
        int first_rc = GAMUT_HIP_OK;
        for (int k = 0; k < n; ++k) {
            const int i = idx[(size_t)k];
            if (host_status) host_status[i] = rcs[(size_t)k];
            if (rcs[(size_t)k] != GAMUT_HIP_OK && (*first_index < 0 || i < *first_index)) {
                *first_index = i; first_rc = rcs[(size_t)k]; snprintf(first_msg, msg_cap, "image %d: %s", i, msgs[(size_t)k].c_str());
            }
        }
        return first_rc;
    }
    std::vector<ProgPrep> prep((size_t)n);
    {
        std::vector<Parser*> parsers((size_t)workers, nullptr);
        for (Parser*& p : parsers) p = new Parser();
        parallel_for(n, workers, [&](int w, int k) { const int i = idx[(size_t)k]; prog_prepare(i, data[i], len[i], info[i], prep[(size_t)k], *parsers[(size_t)w]); });
        for (Parser* p : parsers) delete p;
    }
    // layout: blob slots, global tables
    std::vector<ProgImage> images((size_t)n);
    std::vector<DevHuff> huffs; std::vector<QuantTab> quants;
    std::vector<std::vector<int>> tab_map((size_t)n);
    std::vector<size_t> blob_off((size_t)n, 0);
    size_t blob_size = 0; int max_level = -1; int64_t max_blocks = 0;
    for (int k = 0; k < n; ++k) {
        ProgPrep& pp = prep[(size_t)k]; ProgImage& im = images[(size_t)k];
        memset(&im, 0, sizeof(im));
        if (pp.rc != GAMUT_HIP_OK) continue;
        const int i = idx[(size_t)k];
        if ((coeff_offset[i] & 7) != 0) { pp.rc = GAMUT_HIP_ERR_INVALID_ARG; snprintf(pp.msg, sizeof(pp.msg), "image %d: coefficient offsets must be multiples of 8 elements", i); continue; }
        im.coeff_off = coeff_offset[i]; im.zag_off = zag_offset[i]; im.nb = pp.nb; im.ny = pp.ny; im.comps = pp.comps;
        im.mcus_per_row = info[i].mcus_per_row; im.mcus_per_col = info[i].mcus_per_col; im.n_blocks = im.mcus_per_row * im.mcus_per_col * im.nb;
        im.index = i;
        for (int c = 0; c < 3; ++c) { im.hs[c] = pp.hs[c]; im.vs[c] = pp.vs[c]; im.quant[c] = c < pp.comps ? intern(quants, pp.quant[c]) : 0; }
        for (const DevHuff& d : pp.tabs) tab_map[(size_t)k].push_back(intern(huffs, d));
        for (const ProgScanPrep& s : pp.scans) if (s.level > max_level) max_level = s.level;
        if (im.n_blocks > max_blocks) max_blocks = im.n_blocks;
        blob_off[(size_t)k] = blob_size;
        blob_size += (pp.cap + 255) & ~(size_t)255;
    }
    const double ms_parse = ms_since(t_begin);
    double ms_unstuff = 0, ms_kernels = 0;
    if (max_level >= 0) {
        static thread_local PerDevice<DeviceScratch> scratch_pd, tab_scratch_pd;
        static thread_local PerDevice<PinnedScratch> pinned_pd, tab_pinned_pd;
        DeviceScratch& scratch = scratch_pd.cur(); DeviceScratch& tab_scratch = tab_scratch_pd.cur();
        PinnedScratch& pinned = pinned_pd.cur(); PinnedScratch& tab_pinned = tab_pinned_pd.cur();
        uint8_t* d_blob = (uint8_t*)scratch.get(blob_size + kBlobSlack, stream);
        uint8_t* h_blob = pinned.get(blob_size + kBlobSlack, stream);
        if (!d_blob || !h_blob) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: staging allocation of %zu bytes failed", blob_size + kBlobSlack);
        const auto t_u = std::chrono::steady_clock::now();
        parallel_for(n, workers, [&](int, int k) {
            ProgPrep& pp = prep[(size_t)k];
            if (pp.rc == GAMUT_HIP_OK) { const int i = idx[(size_t)k]; prog_unstuff(i, data[i], info[i], pp, h_blob + blob_off[(size_t)k]); }
        });
        memset(h_blob + blob_size, 0xFF, kBlobSlack);
        // the coefficient blocks of the images start at zero (scans only write what they decode); adjacent images in one call
        {
            auto ok = [&](int k) { return prep[(size_t)k].rc == GAMUT_HIP_OK; };
            for (int k = 0; k < n; ) {
                if (!ok(k)) { ++k; continue; }
                const int64_t begin = images[(size_t)k].coeff_off; int64_t endo = begin + (int64_t)images[(size_t)k].n_blocks * 64;
                int j = k + 1;
                while (j < n && ok(j) && images[(size_t)j].coeff_off == endo) { endo += (int64_t)images[(size_t)j].n_blocks * 64; ++j; }
                GAMUT_HIP_CHECK(hipMemsetAsync(d_coeffs + begin, 0, (size_t)(endo - begin) * sizeof(int16_t), stream));
                k = j;
            }
        }
        // One item list for the launch.  Items are taken in list order and whatever an item waits for must come before it; beyond that
        // the order decides which waves share a SIMD (workgroup i of a launch lands on SIMD i mod 1024 while slots are free), and the
        // decode time of a file is the time of its longest chain of scans.  So: per file the scans by the length of the longest chain
        // they feed (bytes; a scan that others stand on is at least as "long" as they are, so this is also a valid order), and the
        // list takes the first scan of every file, then the second of every file, ...: the long chains of the batch start first and
        // spread over the SIMDs evenly instead of meeting on some of them.
        // Tables -> global indices, segments -> blob offsets; images that failed while unstuffing drop out.
        std::vector<ProgItem> all_items;
        const char* prio_env = getenv("GAMUT_HIP_PROG_PRIO");
        const int prio_mode = prio_env ? atoi(prio_env) : 1;
        const char* order_env = getenv("GAMUT_HIP_PROG_ORDER");
        const bool file_major = order_env && !strcmp(order_env, "file");          // measurements: a file's items next to each other
        std::vector<ProgImage> live; std::vector<int> live_of((size_t)n, -1);
        std::vector<std::vector<ProgItem>> per_file;                               // in the file's own order, dep[] = local indices
        for (int k = 0; k < n; ++k) {
            const ProgPrep& pp = prep[(size_t)k];
            if (pp.rc != GAMUT_HIP_OK) continue;
            live_of[(size_t)k] = (int)live.size(); live.push_back(images[(size_t)k]);
            const size_t n_scans = pp.scans.size();
            // per scan: the scans it stands on -- for every coefficient it touches, the LAST earlier scan that touched it
            std::vector<std::vector<int>> scan_deps(n_scans);
            {
                int last[3][64];
                for (auto& r : last) for (int& v : r) v = -1;
                for (size_t si = 0; si < n_scans; ++si) {
                    const Scan& sc = pp.scans[si].sc;
                    const int hi = prog_scan_reach(pp.scans[si].kind, sc.se);
                    for (int a = 0; a < sc.ncomp; ++a) for (int z = sc.ss; z <= hi; ++z) {
                        const int e = last[sc.comp[a]][z];
                        if (e >= 0 && std::find(scan_deps[si].begin(), scan_deps[si].end(), e) == scan_deps[si].end()) scan_deps[si].push_back(e);
                        last[sc.comp[a]][z] = (int)si;
                    }
                }
            }
            std::vector<int> n_segs(n_scans, 0);
            std::vector<uint64_t> bytes(n_scans, 0), chain(n_scans, 0);
            for (const ProgItem& m : pp.items) { ++n_segs[(size_t)m.scan]; bytes[(size_t)m.scan] += m.end - m.begin; }
            // do the scans line up?  (same components, units and restart segments as everything they stand on: then unit u may follow unit u)
            bool lined_up = true;
            for (size_t si = 0; si < n_scans && lined_up; ++si) {
                const ProgScanPrep& sp = pp.scans[si];
                lined_up = scan_deps[si].size() <= 3;
                for (size_t d = 0; d < scan_deps[si].size() && lined_up; ++d) {
                    const ProgScanPrep& e = pp.scans[(size_t)scan_deps[si][d]];
                    bool same = e.sc.ncomp == sp.sc.ncomp && e.units == sp.units && e.restart_interval == sp.restart_interval && n_segs[(size_t)scan_deps[si][d]] == n_segs[si];
                    for (int c = 0; c < sp.sc.ncomp && same; ++c) same = e.sc.comp[c] == sp.sc.comp[c];
                    lined_up = same;
                }
            }
            uint64_t longest = 1;
            for (size_t si = n_scans; si-- > 0; ) {
                chain[si] = std::max(chain[si], bytes[si]);
                for (int e : scan_deps[si]) chain[(size_t)e] = std::max(chain[(size_t)e], chain[si]);
                longest = std::max(longest, chain[si]);
            }
            std::vector<int> order(n_scans);                                         // the file's scans in launch order
            for (size_t si = 0; si < n_scans; ++si) order[si] = (int)si;
            if (lined_up) std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return chain[(size_t)a] > chain[(size_t)b]; });
            else          std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pp.scans[(size_t)a].level < pp.scans[(size_t)b].level; });
            std::vector<int> first_item(n_scans, -1);
            { int at = 0; for (int si : order) { first_item[(size_t)si] = at; at += n_segs[(size_t)si]; } }
            std::vector<ProgItem> mine(pp.items.size());
            std::vector<int> lower_levels;                                           // not lined up: items of the levels below, per level
            if (!lined_up) {
                int max_lv = 0; for (const ProgScanPrep& sp : pp.scans) max_lv = std::max(max_lv, sp.level);
                lower_levels.assign((size_t)max_lv + 2, 0);
                for (const ProgItem& m : pp.items) ++lower_levels[(size_t)m.level + 1];
                for (size_t l = 1; l < lower_levels.size(); ++l) lower_levels[l] += lower_levels[l - 1];
            }
            for (ProgItem it : pp.items) {
                const int local = first_item[(size_t)it.scan] + it.seg;
                it.dep_pipe = 0; it.ctr = -1; it.ctr_need = 0;
                if (lined_up) {
                    const std::vector<int>& deps = scan_deps[(size_t)it.scan];
                    for (size_t d = 0; d < deps.size(); ++d) { it.dep[d] = first_item[(size_t)deps[d]] + it.seg; it.dep_pipe |= 1 << d; }
                } else { it.ctr = live_of[(size_t)k]; it.ctr_need = lower_levels[(size_t)it.level]; }       // a barrier per level of the file: a count of finished items
                it.prio = 0;
                if (prio_mode) {
                    const uint64_t c = chain[(size_t)it.scan];
                    it.prio = c * 2 >= longest ? 3 : c * 5 >= longest ? 2 : c * 20 >= longest ? 1 : 0;
                }
                it.image = live_of[(size_t)k];
                it.begin += blob_off[(size_t)k]; it.end += blob_off[(size_t)k];
                if (it.kind != PROG_DC_REFINE) for (int c = 0; c < it.ncomp; ++c) it.tab[c] = tab_map[(size_t)k][(size_t)it.tab[c]];
                mine[(size_t)local] = it;
            }
            per_file.push_back(std::move(mine));
        }
        {
            size_t total_items = 0, deepest = 0;
            for (const auto& v : per_file) { total_items += v.size(); deepest = std::max(deepest, v.size()); }
            std::vector<std::vector<int32_t>> place(per_file.size());                // (file, local index) -> place in the launch's list
            for (size_t f = 0; f < per_file.size(); ++f) place[f].resize(per_file[f].size());
            int32_t at = 0;
            if (file_major) { for (size_t f = 0; f < per_file.size(); ++f) for (size_t j = 0; j < per_file[f].size(); ++j) place[f][j] = at++; }
            else for (size_t j = 0; j < deepest; ++j) for (size_t f = 0; f < per_file.size(); ++f) if (j < per_file[f].size()) place[f][j] = at++;
            all_items.resize(total_items);
            for (size_t f = 0; f < per_file.size(); ++f)
                for (size_t j = 0; j < per_file[f].size(); ++j) {
                    ProgItem it = per_file[f][j];
                    for (int d = 0; d < 3; ++d) if (it.dep[d] >= 0) it.dep[d] = place[f][(size_t)it.dep[d]];
                    all_items[(size_t)place[f][j]] = it;
                }
        }
        ms_unstuff = ms_since(t_u);
        if (!live.empty()) {
            auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
            const size_t n_items = all_items.size();
            std::vector<QuantTab>& qn = quants;
            const size_t o_img = 0, o_huff = align(live.size() * sizeof(ProgImage)), o_quant = align(o_huff + huffs.size() * sizeof(DevHuff)),
                         o_items = align(o_quant + qn.size() * sizeof(QuantTab)), o_prog = align(o_items + n_items * sizeof(ProgItem)),
                         o_times = align(o_prog + (n_items + 1 + live.size()) * sizeof(uint32_t)), total = o_times + (trace ? 2 * n_items * sizeof(uint4) : 0) + 256;
            uint8_t* d = (uint8_t*)tab_scratch.get(total, stream);
            uint8_t* h = tab_pinned.get(total, stream);
            if (!d || !h) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: staging allocation of %zu bytes failed", total);
            memcpy(h + o_img, live.data(), live.size() * sizeof(ProgImage));
            memcpy(h + o_huff, huffs.data(), huffs.size() * sizeof(DevHuff));
            memcpy(h + o_quant, qn.data(), qn.size() * sizeof(QuantTab));
            memcpy(h + o_items, all_items.data(), n_items * sizeof(ProgItem));
            GAMUT_HIP_CHECK(hipMemcpyAsync(d, h, o_prog, hipMemcpyHostToDevice, stream));
            GAMUT_HIP_CHECK(hipMemsetAsync(d + o_prog, 0, (n_items + 1 + live.size()) * sizeof(uint32_t), stream));      // progress words, the ticket, a count of finished items per file
            GAMUT_HIP_CHECK(hipMemcpyAsync(d_blob, h_blob, blob_size + kBlobSlack, hipMemcpyHostToDevice, stream));
            uint32_t* st = d_status;
            if (!st) {
                static thread_local PerDevice<DeviceScratch> sink_pd;
                DeviceScratch& sink = sink_pd.cur();
                int top = 0; for (int i : idx) top = i > top ? i : top;
                st = (uint32_t*)sink.get((size_t)(top + 1) * sizeof(uint32_t), stream);
                if (!st) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: status allocation failed");
            }
            for (int k = 0; k < n; ++k) if (prep[(size_t)k].rc == GAMUT_HIP_OK) GAMUT_HIP_CHECK(hipMemsetAsync(st + idx[(size_t)k], 0, sizeof(uint32_t), stream));
            const auto t_k = std::chrono::steady_clock::now();
            if (n_items) {
                hipLaunchKernelGGL(k_prog_scan, dim3((unsigned)n_items), dim3(kProgThreads), 0, stream,
                                   (const ProgItem*)(d + o_items), (const ProgImage*)(d + o_img), (const DevHuff*)(d + o_huff), d_blob, d_coeffs, st,
                                   (uint32_t*)(d + o_prog), (uint32_t*)(d + o_prog) + n_items, trace ? (uint4*)(d + o_times) : (uint4*)nullptr);
                if (int rc = launch_status("jpeg_prog_scan")) return rc;
                if (trace) {
                    const auto t_l = std::chrono::steady_clock::now();
                    (void)hipStreamSynchronize(stream);
                    fprintf(stderr, "[gamut_hip]   scans: %zu segments in one launch, %.1f ms (with what was queued before it)\n", n_items, ms_since(t_l));
                    std::vector<uint4> tm(2 * n_items);
                    if (hipMemcpy(tm.data(), d + o_times, 2 * n_items * sizeof(uint4), hipMemcpyDeviceToHost) == hipSuccess) {
                        uint64_t t_min = ~0ull;
                        for (const uint4& t : tm) t_min = std::min(t_min, ((uint64_t)t.y << 32) | t.x);
                        // per scan number (of its file): mean start, mean duration, mean time spent waiting, over the files of the batch (100 MHz clock)
                        double start[16] = { 0 }, dur[16] = { 0 }, wt[16] = { 0 }, endmax[16] = { 0 }; int cnt[16] = { 0 };
                        for (size_t j = 0; j < n_items; ++j) {
                            const int sc = all_items[j].scan;
                            if (sc >= 16) continue;
                            const double st0 = (double)((((uint64_t)tm[j].y << 32) | tm[j].x) - t_min) / 1e5;
                            start[sc] += st0; dur[sc] += tm[j].z / 1e5; wt[sc] += tm[j].w / 1e5; ++cnt[sc];
                            endmax[sc] = std::max(endmax[sc], st0 + tm[j].z / 1e5);
                        }
#ifdef GAMUT_PROG_PROFILE
                        double pf[16][4] = { { 0 } };
                        for (size_t j = 0; j < n_items; ++j) { const int sc = all_items[j].scan; if (sc < 16) { const uint4& q = tm[n_items + j]; pf[sc][0] += q.x; pf[sc][1] += q.y; pf[sc][2] += q.z; pf[sc][3] += q.w; } }
                        for (int sc = 0; sc < 16; ++sc) if (cnt[sc] && pf[sc][3] > 0)
                            fprintf(stderr, "[gamut_hip]     scan %2d (AC refinement) per segment: block set-up %.0f, symbol walk %.0f, tail %.0f shader clocks; %.0f symbols\n",
                                    sc, pf[sc][0] / cnt[sc], pf[sc][1] / cnt[sc], pf[sc][2] / cnt[sc], pf[sc][3] / cnt[sc]);
#endif
                        for (int sc = 0; sc < 16; ++sc) if (cnt[sc])
                            fprintf(stderr, "[gamut_hip]     scan %2d: %5d segments, start %7.2f ms, runs %7.2f ms of which waiting %7.2f ms, last one ends at %7.2f ms\n",
                                    sc, cnt[sc], start[sc] / cnt[sc], dur[sc] / cnt[sc], wt[sc] / cnt[sc], endmax[sc]);
                    }
                }
            }
            const unsigned gx = (unsigned)std::min<int64_t>((max_blocks + 31) / 32, 4096);
            hipLaunchKernelGGL(k_prog_finalize, dim3(gx, (unsigned)live.size()), dim3(256), 0, stream,
                               (const ProgImage*)(d + o_img), (int)live.size(), (const int16_t*)(d + o_quant), d_coeffs, d_max_zag);
            if (int rc = launch_status("jpeg_prog_finalize")) return rc;
            GAMUT_HIP_CHECK(hipStreamSynchronize(stream));     // the per-thread staging buffers are reused by the next call
            // whatever a kernel flagged -- a bit pattern no code word begins, a run past coefficient 63, a segment that ran out, octets between an
            // interval's last bit and its RSTn (restart_leftover_bad) -- is the host feeder's to judge (kHostRedo, below)
            {
                int lo = idx[0], hi = idx[0];
                for (int i : idx) { lo = std::min(lo, i); hi = std::max(hi, i); }
                std::vector<uint32_t> flags((size_t)(hi - lo + 1));
                GAMUT_HIP_CHECK(hipMemcpy(flags.data(), st + lo, flags.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
                for (int k = 0; k < n; ++k) {
                    ProgPrep& pp = prep[(size_t)k];
                    if (pp.rc == GAMUT_HIP_OK && flags[(size_t)(idx[(size_t)k] - lo)]) pp.rc = kHostRedo;
                }
            }
            ms_kernels = ms_since(t_k);
        }
    }
    if (trace) fprintf(stderr, "[gamut_hip] progressive_decode_device: %d files, %d levels: headers %.1f ms, unstuff %.1f ms, upload + kernels %.1f ms\n",
                       n, max_level + 1, ms_parse, ms_unstuff, ms_kernels);
    {
        std::vector<int> redo, at;
        for (int k = 0; k < n; ++k) if (prep[(size_t)k].rc == kHostRedo) { redo.push_back(idx[(size_t)k]); at.push_back(k); }
        if (!redo.empty()) {
            std::vector<int> rcs; std::vector<std::string> msgs;
            if (int rc = on_the_host(redo, rcs, msgs)) { *first_index = -1; return rc; }
            for (size_t j = 0; j < redo.size(); ++j) {
                ProgPrep& pp = prep[(size_t)at[j]];
                pp.rc = rcs[j];
                if (rcs[j] != GAMUT_HIP_OK) snprintf(pp.msg, sizeof(pp.msg), "image %d: %s", redo[j], msgs[j].c_str());
            }
        }
    }
    int first_rc = GAMUT_HIP_OK;
    for (int k = 0; k < n; ++k) {
        const ProgPrep& pp = prep[(size_t)k];
        const int i = idx[(size_t)k];
        if (host_status) host_status[i] = pp.rc;
        if (pp.rc != GAMUT_HIP_OK && (*first_index < 0 || i < *first_index)) { *first_index = i; first_rc = pp.rc; snprintf(first_msg, msg_cap, "%s", pp.msg); }
    }
    return first_rc;
}
