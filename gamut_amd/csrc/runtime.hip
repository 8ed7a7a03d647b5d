// runtime.hip -- C-ABI runtime entry points of libgamut_hip (init, errors, memory, streams)
// and the host-pointer drop-in for scanlinesConvert / scanlinesCopy.
#include "common.hpp"
#include <algorithm>
#include <condition_variable>
#include <mutex>

namespace gamut {

char* last_error_buf()
{
    static thread_local char buf[512] = { 0 };
    return buf;
}

int set_error(int status, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return status;
}

int host_threads()
{
    static const int n = [] {
        int hw = (int)std::thread::hardware_concurrency();
        if (hw < 1) hw = 1;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char quota[32] = { 0 }; long long period = 0;
            if (fscanf(f, "%31s %lld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
                const long long q = atoll(quota);
                const int cores = (int)((q + period - 1) / period);
                if (cores >= 1 && cores < hw) hw = cores;
            }
            fclose(f);
        }
        // one process per GPU on a shared host (torchrun / mpirun export the number of ranks on this node): every rank takes
        // its share of the cores -- 8 ranks x 16 inflating threads on a 16-core quota run slower than 8 x 2.  GAMUT_HIP_HOST_THREADS
        // overrides everything.
        for (const char* var : { "LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS" })
            if (const char* v = getenv(var)) { const int ranks = atoi(v); if (ranks > 1) { hw = (hw + ranks - 1) / ranks; break; } }
        if (const char* v = getenv("GAMUT_HIP_HOST_THREADS")) { const int t = atoi(v); if (t >= 1) hw = t; }
        return hw < 1 ? 1 : hw;
    }();
    return n;
}

// The helper pool of parallel_for: host_threads() persistent threads (at most 64), started on first use, never joined (the process may
// end while they wait; their thread-local staging must not be torn down under a running DMA).
namespace {
struct PoolTask { void (*run)(void*, int); void* ctx; int worker; std::atomic<int>* left; };
struct HelperPool {
    std::mutex m; std::condition_variable cv; std::vector<PoolTask> q; size_t head = 0; int threads = 0;
    void start()
    {
        const int want = std::min(64, std::max(1, host_threads()));
        while (threads < want) {
            std::thread([this] {
                for (;;) {
                    PoolTask t;
                    { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [this] { return head < q.size(); }); t = q[head++]; if (head == q.size()) { q.clear(); head = 0; } }
                    t.run(t.ctx, t.worker);
                    t.left->fetch_sub(1, std::memory_order_release);
                }
            }).detach();
            ++threads;
        }
    }
};
HelperPool& helper_pool() { static HelperPool* p = new HelperPool(); return *p; }
} // namespace

void pool_submit(void (*run)(void*, int), void* ctx, int first_worker, int n_helpers, std::atomic<int>* left)
{
    // Nothing may leave this function once a task is visible to a helper: the tasks refer to the caller's frame.  Threads are started
    // and the queue's room is reserved FIRST (both may fail: then fewer helpers, or none, take part and the caller does the work);
    // `left` is published with the number of tasks that really go in, and push_back into reserved room cannot throw.
    HelperPool& P = helper_pool();
    int queued = 0;
    {
        std::lock_guard<std::mutex> lk(P.m);
        try { P.start(); } catch (...) { }
        if (P.threads > 0) {
            try { P.q.reserve(P.q.size() + (size_t)n_helpers); queued = n_helpers; } catch (...) { queued = 0; }
        }
        left->store(queued, std::memory_order_relaxed);
        for (int k = 0; k < queued; ++k) P.q.push_back(PoolTask{ run, ctx, first_worker + k, left });
    }
    if (queued) P.cv.notify_all();
}
// The caller has run out of indices: its tasks that no helper has picked up yet have nothing to do -- take them back instead of
// waiting for busy helpers (another leg's long copy tasks) to dequeue them one by one.
void pool_cancel(std::atomic<int>* left)
{
    HelperPool& P = helper_pool();
    std::lock_guard<std::mutex> lk(P.m);
    size_t k = P.head; int removed = 0;
    for (size_t i = P.head; i < P.q.size(); ++i) {
        if (P.q[i].left == left) { ++removed; continue; }
        P.q[k++] = P.q[i];
    }
    P.q.resize(k);
    if (P.head == P.q.size()) { P.q.clear(); P.head = 0; }
    if (removed) left->fetch_sub(removed, std::memory_order_release);
}

hipStream_t thread_stream()
{
    static thread_local PerDevice<hipStream_t> s_pd;           // a stream belongs to the device it was created on
    hipStream_t& s = s_pd.cur();
    if (!s) {
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;   // falls back to the null stream
    }
    return s;
}

namespace {

// RAII device buffer for the synchronous host entry points
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n)
    {
        if (hipMalloc(&p, n ? n : 1) != hipSuccess) { p = nullptr; return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed", n); }
        return GAMUT_HIP_OK;
    }
};

int require_device()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return set_error(GAMUT_HIP_ERR_NO_DEVICE, "no HIP device available (libgamut_hip has no CPU fallback)");
    return GAMUT_HIP_OK;
}

// Host drop-in shared by convert and copy: stage the rectangle through tight HBM buffers.
int host_convert(int srcType, const uint8_t* src, int srcPitch, int dstType, uint8_t* dst, int dstPitch, int width, int height)
{
    if (!valid_type(srcType) || !valid_type(dstType))
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "scanlinesConvert: invalid PixelType %d -> %d", srcType, dstType);
    if (width < 0 || height < 0) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "scanlinesConvert: negative size");
    if (width == 0 || height == 0) return GAMUT_HIP_OK;
    if (!src || !dst) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "scanlinesConvert: null pointer");
    if (int rc = require_device()) return rc;

    const size_t srow = (size_t)width * kPixelSize[srcType], drow = (size_t)width * kPixelSize[dstType];
    const size_t sabs = (size_t)(srcPitch < 0 ? -(int64_t)srcPitch : srcPitch), dabs = (size_t)(dstPitch < 0 ? -(int64_t)dstPitch : dstPitch);
    if (height > 1 && (sabs < srow || dabs < drow))
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "scanlinesConvert: pitch smaller than a scanline");
    // lowest-address row of each rectangle (pitch < 0 => stored bottom-up)
    const uint8_t* s0 = srcPitch < 0 ? src + (int64_t)(height - 1) * srcPitch : src;
    uint8_t*       d0 = dstPitch < 0 ? dst + (int64_t)(height - 1) * dstPitch : dst;

    DevBuf ds, dd;
    if (int rc = ds.alloc(srow * height)) return rc;
    if (int rc = dd.alloc(drow * height)) return rc;
    hipStream_t st = thread_stream();
    GAMUT_HIP_CHECK(hipMemcpy2DAsync(ds.p, srow, s0, height > 1 ? sabs : srow, srow, height, hipMemcpyHostToDevice, st));
    // device rectangles keep the memory order of the host ones, so flipped stays flipped
    const uint8_t* dsrc = static_cast<const uint8_t*>(ds.p) + (srcPitch < 0 ? (int64_t)(height - 1) * srow : 0);
    uint8_t*       ddst = static_cast<uint8_t*>(dd.p) + (dstPitch < 0 ? (int64_t)(height - 1) * drow : 0);
    if (int rc = convert_device(srcType, dsrc, srcPitch < 0 ? -(int64_t)srow : (int64_t)srow, 0,
                                dstType, ddst, dstPitch < 0 ? -(int64_t)drow : (int64_t)drow, 0, width, height, 1, st))
        return rc;
    GAMUT_HIP_CHECK(hipMemcpy2DAsync(d0, height > 1 ? dabs : drow, dd.p, drow, drow, height, hipMemcpyDeviceToHost, st));
    GAMUT_HIP_CHECK(hipStreamSynchronize(st));
    return GAMUT_HIP_OK;
}

} // namespace
} // namespace gamut

using namespace gamut;

extern "C" {

const char* gamut_hip_version(void) { return "gamut-hip 0.1 (gfx950)"; }

int gamut_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gamut_hip_init(int device)
{
    clear_error();
    if (int rc = require_device()) return rc;
    if (device >= 0) GAMUT_HIP_CHECK(hipSetDevice(device));
    GAMUT_HIP_CHECK(hipFree(nullptr));      // force context creation
    return GAMUT_HIP_OK;
}

void gamut_hip_shutdown(void) { (void)hipDeviceSynchronize(); }

const char* gamut_hip_last_error(void) { return last_error_buf(); }

void* gamut_hip_device_malloc(size_t bytes)
{
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
void gamut_hip_device_free(void* p) { if (p) (void)hipFree(p); }

void* gamut_hip_host_malloc_pinned(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
void gamut_hip_host_free_pinned(void* p) { if (p) (void)hipHostFree(p); }

int gamut_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream)
{
    GAMUT_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, pick_stream(stream)));
    return GAMUT_HIP_OK;
}
int gamut_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream)
{
    GAMUT_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, pick_stream(stream)));
    return GAMUT_HIP_OK;
}
void* gamut_hip_stream_create(void)
{
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { set_error(GAMUT_HIP_ERR_HIP, "hipStreamCreate failed"); return nullptr; }
    return s;
}
void gamut_hip_stream_destroy(void* stream) { if (stream) (void)hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)); }
int gamut_hip_stream_synchronize(void* stream)
{
    GAMUT_HIP_CHECK(hipStreamSynchronize(pick_stream(stream)));
    return GAMUT_HIP_OK;
}

int gamut_hip_pixel_type_size(int type) { return valid_type(type) ? kPixelSize[type] : 0; }

int gamut_hip_scanlines_inter_type(int srcType, int dstType)
{
    auto plain8 = [](int t) { return t == GAMUT_PIXEL_l8 || t == GAMUT_PIXEL_la8 || t == GAMUT_PIXEL_rgb8 || t == GAMUT_PIXEL_rgba8; };
    return (plain8(srcType) && plain8(dstType)) ? GAMUT_PIXEL_rgba8 : GAMUT_PIXEL_rgbaf32;
}

int gamut_hip_scanlines_convert(int srcType, const uint8_t* src, int srcPitch,
                                int dstType, uint8_t* dst, int dstPitch, int width, int height)
{
    clear_error();
    return host_convert(srcType, src, srcPitch, dstType, dst, dstPitch, width, height);
}

int gamut_hip_scanlines_copy(int type, const uint8_t* src, int srcPitch, uint8_t* dst, int dstPitch, int width, int height)
{
    clear_error();
    return host_convert(type, src, srcPitch, type, dst, dstPitch, width, height);
}

int gamut_hip_scanlines_convert_device(int srcType, const void* src, int64_t srcPitch, int64_t srcLayerOffset,
                                       int dstType, void* dst, int64_t dstPitch, int64_t dstLayerOffset,
                                       int width, int height, int layers, void* stream)
{
    clear_error();
    return convert_device(srcType, src, srcPitch, srcLayerOffset, dstType, dst, dstPitch, dstLayerOffset,
                          width, height, layers, pick_stream(stream));
}

} // extern "C"
