// qoi.hip -- QOI decode (needed for the mixed JPEG/PNG/QOI batch of BASELINE.json config 5).
//
// Replaces qoi_decode (source/gamut/codecs/qoi.d:448-550).  The format is a byte-serial state machine (previous pixel,
// 64-entry colour hash, run counter): nothing to parallelise inside a stream, so the parallelism is the batch -- one
// lane per image.  The lane's hash table lives in LDS (lane-interleaved, so the 64 lanes of a wave hit 64 different
// banks x 2), the stream is read byte-wise through L1, pixels are written as they come.  The host only validates the
// 14-byte header (same checks as :472-480) and uploads the streams as they are.
#include "common.hpp"
#include <vector>

namespace gamut {
namespace {

constexpr uint32_t kQoiMagic = 0x716F6966u, kQoiPixelsMax = 400000000u;      // qoi.d:244, :251
constexpr int kQoiHeader = 14, kQoiPadding = 8;                                // :245, :268

struct QoiItem { uint64_t begin; int64_t out_off; uint32_t size, npx; int32_t channels, pad; };

// A lane must not touch global memory per pixel: every wait for a load also waits for the stores issued before it (one
// in-order counter), so byte-wise loads and per-pixel stores made a lane sit out a memory round trip several times per
// pixel (1.3 us per pixel measured).  And it must not branch on the op: the six op kinds of 64 unrelated streams put
// every branch on the wave's path in nearly every iteration (0.37 us per pixel measured with per-op branches: ~200
// instructions, one wave per SIMD, one instruction per ~4 cycles).  So:
//   * everything a lane touches per pixel is in registers or LDS -- the hash table, a 256-byte window of the stream and
//     a 64-pixel output buffer, all dword-interleaved over the lanes ([slot][lane]: 64 lanes, 64 banks);
//   * an op is decoded WITHOUT branches: its first byte indexes a 256-entry table (bytes used, run length, one bit per
//     op kind), the candidate pixels of all kinds are computed (the DIFF / LUMA deltas come from tables too and are
//     added bytewise, SWAR) and the right one is picked with bit-field masks; an iteration that only continues a run
//     is the same code with a zeroed table entry;
//   * the next 5..8 stream bytes sit in a 64-bit register topped up from a dword that was read from the window one
//     top-up earlier, so no LDS latency is on the byte path;
//   * the window is refilled for the whole wave at once: when SOME lane has less than two 64-byte blocks left, every lane
//     with a free block commits the block it has in flight (16 VGPRs) and requests the next -- a refill every ~15
//     iterations instead of one lane or another refilling in nine iterations out of ten;
//   * the pixel loop is uniform over the wave, so the output buffers fill up together and are flushed with dwordx4 stores.
constexpr int kQoiWinDwords = 64, kQoiOutPx = 64;             // 4 blocks of 64 bytes per lane
constexpr int kQoiSlack = GAMUT_HIP_QOI_SLACK;                // readable bytes guaranteed after every stream

__device__ __forceinline__ uint32_t qoi_add_bytes(uint32_t x, uint32_t y)       // bytewise (x + y) mod 256
{
    return ((x & 0x7f7f7f7fu) + (y & 0x7f7f7f7fu)) ^ ((x ^ y) & 0x80808080u);
}

__global__ __launch_bounds__(64) void k_qoi_decode(const QoiItem* items, int n, const uint8_t* blob, uint8_t* out)
{
    __shared__ uint32_t index[64 * 64];                       // [hash][lane]
    __shared__ uint32_t sh_in[kQoiWinDwords * 64];            // [dword of the window][lane]
    __shared__ uint32_t sh_out[kQoiOutPx * 64];               // [pixel][lane], always r|g<<8|b<<16|a<<24
    __shared__ uint32_t optab[256];                           // first byte -> bytes used | kind bits << 3 | run << 8
    __shared__ uint2 deltab[64];                              // low six bits -> (QOI_OP_DIFF delta, QOI_OP_LUMA green part), packed bytes
    __shared__ uint32_t luma2[256];                           // second byte of QOI_OP_LUMA -> (dr - dg, 0, db - dg, 0) + 8 removed
    constexpr uint32_t K_RGB = 1u << 3, K_RGBA = 1u << 4, K_INDEX = 1u << 5, K_DIFF = 1u << 6, K_LUMA = 1u << 7;
    const int lane = threadIdx.x;
    #pragma unroll 8
    for (int k = 0; k < 64; ++k) index[k * 64 + lane] = 0;    // memset(index, 0) :491  (a lane only touches its own column)
    #pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t v = (uint32_t)(k * 64 + lane);
        uint32_t e;
        if (v == 0xFE) e = 4 | K_RGB; else if (v == 0xFF) e = 5 | K_RGBA;
        else if ((v >> 6) == 0) e = 1 | K_INDEX; else if ((v >> 6) == 1) e = 1 | K_DIFF; else if ((v >> 6) == 2) e = 2 | K_LUMA;
        else e = 1 | (v & 63) << 8;                                                            // QOI_OP_RUN
        optab[v] = e;
        luma2[v] = (v >> 4) | (v & 15) << 16;
    }
    {
        const uint32_t k = (uint32_t)lane, vg = (k - 32) & 255, vg8 = (k - 40) & 255;          // vg = k - 32; vg - 8
        deltab[k] = make_uint2(((((k >> 4) & 3) - 2) & 255) | ((((k >> 2) & 3) - 2) & 255) << 8 | (((k & 3) - 2) & 255) << 16, vg8 | vg << 8 | vg8 << 16);
    }
    __syncthreads();
    const int i = blockIdx.x * 64 + lane;
    if (i >= n) return;
    const QoiItem it = items[i];
    const bool rgba = it.channels == 4;
    const int bpp = rgba ? 4 : 3;
    uint8_t* pixels = out + it.out_off;
    const uint8_t* stream = blob + it.begin + kQoiHeader;      // chunks start here
    uint32_t* win = sh_in + lane;
    uint32_t* obuf = sh_out + lane;
    uint32_t px = 0xFF000000u;                                // r = g = b = 0, a = 255 :492-495
    // bytes [14, size - 8) are chunks (p < chunks_len, :498); a chunk may read up to 4 bytes further (padding / slack)
    const int chunk_bytes = (int)it.size - kQoiPadding - kQoiHeader;
    const uint32_t fetch_limit = (uint32_t)(chunk_bytes > 0 ? chunk_bytes : 0) + 5;   // no byte at or beyond this is ever decoded
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    struct __attribute__((packed, aligned(1))) AnyVec { u32x4 v; };               // 16 bytes at any address
    u32x4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;      // the block in flight
    uint32_t fetched = 0, stored = 0, pulled = 0;            // stream bytes requested / in the window / read from the window
    auto issue = [&]() {
        if (fetched < fetch_limit) {
            const AnyVec* src = reinterpret_cast<const AnyVec*>(stream + fetched);
            q0 = src[0].v; q1 = src[1].v; q2 = src[2].v; q3 = src[3].v;
        }
        fetched += 64;
    };
    auto commit = [&]() {
        uint32_t* w = win + ((stored >> 2) & (kQoiWinDwords - 1)) * 64;
        w[0 * 64] = q0.x; w[1 * 64] = q0.y; w[2 * 64] = q0.z; w[3 * 64] = q0.w;
        w[4 * 64] = q1.x; w[5 * 64] = q1.y; w[6 * 64] = q1.z; w[7 * 64] = q1.w;
        w[8 * 64] = q2.x; w[9 * 64] = q2.y; w[10 * 64] = q2.z; w[11 * 64] = q2.w;
        w[12 * 64] = q3.x; w[13 * 64] = q3.y; w[14 * 64] = q3.z; w[15 * 64] = q3.w;
        stored += 64;
    };
    issue(); commit(); issue(); commit(); issue(); commit(); issue(); commit(); issue();      // four blocks in the window, the fifth in flight
    uint32_t lo = 0, hi = 0, ahead; int valid = 0;            // `valid` stream bytes in hi:lo, lowest byte first; `ahead` = the dword after them
    auto read_ahead = [&]() { ahead = win[((pulled >> 2) & (kQoiWinDwords - 1)) * 64]; pulled += 4; };
    read_ahead(); lo = ahead; read_ahead(); hi = ahead; valid = 8; read_ahead();
    int left = chunk_bytes > 0 ? chunk_bytes : 0;             // chunk bytes not decoded yet
    uint32_t flushed = 0;                                     // pixels already written out
    int run = 0, staged = 0;                                  // pixels waiting in obuf
    auto flush = [&](int npx) {                               // npx pixels from obuf to the image
        uint8_t* o = pixels + (size_t)flushed * bpp;
        int k = 0;
        if (rgba) {
            for (; k + 4 <= npx; k += 4) {
                const u32x4 v = {obuf[k * 64], obuf[(k + 1) * 64], obuf[(k + 2) * 64], obuf[(k + 3) * 64]};
                reinterpret_cast<AnyVec*>(o + k * 4)->v = v;
            }
            for (; k < npx; ++k) { const uint32_t v = obuf[k * 64]; o[k * 4] = (uint8_t)v; o[k * 4 + 1] = (uint8_t)(v >> 8); o[k * 4 + 2] = (uint8_t)(v >> 16); o[k * 4 + 3] = (uint8_t)(v >> 24); }
        } else {
            struct __attribute__((packed, aligned(1))) AnyU32 { uint32_t v; };
            for (; k + 4 <= npx; k += 4) {                    // 4 pixels = 3 dwords
                const uint32_t p0 = obuf[k * 64] & 0xFFFFFFu, p1 = obuf[(k + 1) * 64] & 0xFFFFFFu, p2 = obuf[(k + 2) * 64] & 0xFFFFFFu, p3 = obuf[(k + 3) * 64] & 0xFFFFFFu;
                AnyU32* d = reinterpret_cast<AnyU32*>(o + k * 3);
                d[0].v = p0 | p1 << 24; d[1].v = p1 >> 8 | p2 << 16; d[2].v = p2 >> 16 | p3 << 8;
            }
            for (; k < npx; ++k) { const uint32_t v = obuf[k * 64]; o[k * 3] = (uint8_t)v; o[k * 3 + 1] = (uint8_t)(v >> 8); o[k * 3 + 2] = (uint8_t)(v >> 16); }
        }
        flushed += (uint32_t)npx;
    };
    for (uint32_t p = 0; p < it.npx; ++p) {
        // window refill for the whole wave at once
        if (__any((int)(stored - pulled) <= 128)) {
            if (stored - (pulled & ~63u) < 4 * 64) { commit(); issue(); }         // a block of the lane's window has been read completely
        }
        // one op, or one more pixel of a run: table entry zeroed when no op is decoded (run going on, or the chunks are used up)
        const uint32_t b1 = lo & 255u;
        uint32_t e = optab[b1];
        e = (run == 0 && left > 0) ? e : 0u;
        const uint32_t iv = index[(b1 & 63u) * 64 + lane];
        const uint2 dl = deltab[b1 & 63u];
        const uint32_t l2 = luma2[(lo >> 8) & 255u];
        const uint32_t c_rgba = __builtin_amdgcn_alignbit(hi, lo, 8);                                   // stream bytes 1..4
        const uint32_t c_diff = qoi_add_bytes(px, dl.x);
        const uint32_t c_luma = qoi_add_bytes(px, qoi_add_bytes(dl.y, l2));
        auto pick = [](uint32_t mask, uint32_t yes, uint32_t no) { return (yes & mask) | (no & ~mask); };     // v_bfi_b32
        uint32_t v = px;
        v = pick((uint32_t)(((int32_t)(e << 28)) >> 31) & 0x00FFFFFFu, c_rgba, v);                     // QOI_OP_RGB keeps alpha
        v = pick((uint32_t)(((int32_t)(e << 27)) >> 31), c_rgba, v);                                   // QOI_OP_RGBA
        v = pick((uint32_t)(((int32_t)(e << 26)) >> 31), iv, v);                                       // QOI_OP_INDEX
        v = pick((uint32_t)(((int32_t)(e << 25)) >> 31), c_diff, v);                                   // QOI_OP_DIFF
        v = pick((uint32_t)(((int32_t)(e << 24)) >> 31), c_luma, v);                                   // QOI_OP_LUMA
        px = v;
        const int used = (int)(e & 7u);
        run = (run > 0 ? run - 1 : 0) + (int)((e >> 8) & 63u);
        left -= used;
        // consume `used` bytes; top the register up to >= 5 bytes again
        {
            const uint64_t bits = ((uint64_t)hi << 32 | lo) >> (8 * used);
            lo = (uint32_t)bits; hi = (uint32_t)(bits >> 32); valid -= used;
        }
        {
            const bool need = valid < 5;
            const uint64_t add = (uint64_t)ahead << (8 * (valid & 7));
            lo |= need ? (uint32_t)add : 0u; hi |= need ? (uint32_t)(add >> 32) : 0u;
            valid += need ? 4 : 0;
            const uint32_t nxt = win[((pulled >> 2) & (kQoiWinDwords - 1)) * 64];
            ahead = need ? nxt : ahead; pulled += need ? 4u : 0u;
        }
        if (__any(valid < 5)) {                               // only after a 5-byte chunk that left nothing
            if (valid < 5) {
                const uint64_t add = (uint64_t)ahead << (8 * valid);
                lo |= (uint32_t)add; hi |= (uint32_t)(add >> 32); valid += 4;
                read_ahead();
            }
        }
        index[(__builtin_amdgcn_udot4(px, 0x0B070503u, 0u, false) & 63u) * 64 + lane] = px;                 // QOI_COLOR_HASH :239-242 (r*3 + g*5 + b*7 + a*11)
        obuf[staged * 64] = px;
        if (++staged == kQoiOutPx) { flush(kQoiOutPx); staged = 0; }
    }
    if (staged) flush(staged);
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


// upload the item table and decode; returns when the decode has finished (the pageable item vector dies with the call)
int launch_items(const std::vector<QoiItem>& items, uint8_t* d_items, const uint8_t* d_blob, uint8_t* d_out, hipStream_t stream)
{
    const int n = (int)items.size();
    GAMUT_HIP_CHECK(hipMemcpyAsync(d_items, items.data(), items.size() * sizeof(QoiItem), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_qoi_decode, dim3((n + 63) / 64), dim3(64), 0, stream, (const QoiItem*)d_items, n, d_blob, d_out);
    if (int rc = launch_status("qoi_decode")) return rc;
    GAMUT_HIP_CHECK(hipStreamSynchronize(stream));
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
        static thread_local DeviceScratch staging;
        static thread_local PinnedScratch pinned;
        uint8_t* d = (uint8_t*)staging.get(total);
        uint8_t* h = pinned.get(total);
        if (!d || !h) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "qoi: staging allocation of %zu bytes failed", total);
        // the files are gathered into one pinned image on a few host threads and go up in one DMA
        int workers = host_threads();
        workers = workers < 1 ? 1 : workers > 16 ? 16 : workers;
        if ((size_t)workers > blob_size / (4u << 20) + 1) workers = (int)(blob_size / (4u << 20) + 1);
        parallel_for((int)items.size(), workers, [&](int, int k) {
            uint8_t* dst = h + o_blob + items[(size_t)k].begin;
            memcpy(dst, data[src[(size_t)k]], items[(size_t)k].size);
            memset(dst + items[(size_t)k].size, 0, kQoiSlack);
        });
        memcpy(h, items.data(), items.size() * sizeof(QoiItem));
        GAMUT_HIP_CHECK(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, stream));
        const int n = (int)items.size();
        hipLaunchKernelGGL(k_qoi_decode, dim3((n + 63) / 64), dim3(64), 0, stream, (const QoiItem*)d, n, (const uint8_t*)(d + o_blob), d_out);
        if (int rc = launch_status("qoi_decode")) return rc;
        GAMUT_HIP_CHECK(hipStreamSynchronize(stream));         // the per-thread staging buffers are reused by the next call
    }
    if (first != GAMUT_HIP_OK) return set_error(first, "image %d: qoi: bad header or arguments", first_idx);
    return GAMUT_HIP_OK;
}

} // namespace
} // namespace gamut

using namespace gamut;

extern "C" {

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
        static thread_local DeviceScratch table;
        uint8_t* d_items = (uint8_t*)table.get(items.size() * sizeof(QoiItem));
        if (!d_items) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "qoi: device allocation of the item table failed");
        return launch_items(items, d_items, blob, out, pick_stream(stream));
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
    bool ok = hipMalloc(&dout, bytes ? bytes : 1) == hipSuccess;
    if (!ok) set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "qoi: hipMalloc(%zu) failed", bytes);
    const uint8_t* ptr = (const uint8_t*)data; const int64_t off = 0;
    hipStream_t st = thread_stream();
    try { ok = ok && decode_batch(&ptr, &size, 1, channels, &off, (uint8_t*)dout, desc, nullptr, st) == GAMUT_HIP_OK; }
    catch (...) { set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "qoi: out of host memory"); ok = false; }
    if (ok && (hipMemcpyAsync(result, dout, bytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) {
        set_error(GAMUT_HIP_ERR_HIP, "qoi: copy back failed"); ok = false;
    }
    if (dout) (void)hipFree(dout);
    if (!ok) { free(result); return nullptr; }
    return result;
}

} // extern "C"
