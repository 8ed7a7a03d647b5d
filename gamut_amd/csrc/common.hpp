// common.hpp -- shared host-side plumbing of libgamut_hip (error strings, streams).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <memory>
#include <new>
#include <algorithm>
#include <thread>
#include <vector>
#include "../../include/gamut_hip.h"

namespace gamut {

// Thread-local message, same convention as Image._error in the reference
// (image.d:1563-1570): a zero-terminated C string, never thrown.
char* last_error_buf();
int   set_error(int status, const char* fmt, ...);
inline void clear_error() { last_error_buf()[0] = 0; }

// per-thread private stream used by the synchronous host drop-ins (created lazily)
hipStream_t thread_stream();
// device entry points: the caller's hipStream_t; NULL is HIP's null (legacy default) stream, so work
// enqueued by a caller that never touches streams is ordered with everything else it does
inline hipStream_t pick_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// A host process may drive several GPUs from one thread (hipSetDevice between calls) -- what a D host with "one host thread +
// one HIP stream per GPU"  does as well as a thread that simply serves two devices in turn.  Everything the
// library caches per thread -- staging buffers, private streams, events -- therefore lives in a slot of the CURRENT device:
// device memory and streams belong to the device they were created on.
inline int current_device()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
    return d < 0 ? 0 : d;
}
// The slot of a device is made the first time the thread uses the object on that device (a few pointers of thread-local storage per
// object; inline slots for 64 devices made the library's TLS segment 392 KB, paid by every short-lived worker thread on its first
// touch of any thread_local of the library).  Never freed, like the buffers the slots hold: the HIP runtime may be gone at thread exit.
template <class T> struct PerDevice {
    struct Slot { int dev; T* v; };
    Slot* slots = nullptr; int n = 0;
    T& cur()
    {
        const int d = current_device();
        for (int i = 0; i < n; ++i) if (slots[i].dev == d) return *slots[i].v;
        std::unique_ptr<T> v(new T());                       // (allocation failure here throws: the entry points catch and report it)
        Slot* grown = static_cast<Slot*>(realloc(slots, (size_t)(n + 1) * sizeof(Slot)));
        if (!grown) throw std::bad_alloc();                  // (`slots` is untouched by a failed realloc)
        slots = grown;                                       // nothing can throw between the realloc and this line
        slots[n] = Slot{ d, v.release() };
        return *slots[n++].v;
    }
};

// Per-thread, per-device staging buffers (declared `static thread_local PerDevice<...>` where they are used): grown on demand,
// never shrunk, and intentionally not freed at thread exit -- the HIP runtime may already be gone by then.
// Growth is stream-ordered by hand and never waits: the outgrown buffer is parked behind an event recorded on the stream that used
// it last and freed by a later call once that event has passed (the entry points that use these buffers end with a wait on
// their stream, so in practice the event has long passed; nothing here relies on it).  No hipDeviceSynchronize: other threads'
// streams are none of this thread's business.  (Not hipMallocAsync / hipFreeAsync: kernels of the next launch were seen reading a
// table as it had been BEFORE its stream-ordered re-allocation)
struct RetireList {
    struct Old { void* p; hipEvent_t ev; bool host; };
    Old old[8]; int n = 0;
    static void release(const Old& o) { if (o.host) (void)hipHostFree(o.p); else (void)hipFree(o.p); if (o.ev) (void)hipEventDestroy(o.ev); }
    void reap(bool make_room)
    {
        int k = 0;
        for (int i = 0; i < n; ++i) {
            const bool passed = !old[i].ev || hipEventQuery(old[i].ev) == hipSuccess;
            if (passed) release(old[i]); else old[k++] = old[i];
        }
        (void)hipGetLastError();                              // hipErrorNotReady is not an error
        n = k;
        if (make_room && n == 8) { (void)hipEventSynchronize(old[0].ev); release(old[0]); for (int i = 1; i < n; ++i) old[i - 1] = old[i]; --n; }
    }
    void park(void* p, bool host, hipStream_t last_user)
    {
        reap(true);
        hipEvent_t ev = nullptr;
        // A buffer whose user never named its stream (get() without one) may be in use on any of the library's non-blocking streams,
        // which an event on the legacy null stream does not order against: drain the device before the buffer is parked (growth is rare).
        if (!last_user) (void)hipDeviceSynchronize();
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, last_user) != hipSuccess) {
            (void)hipGetLastError();
            if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
            (void)hipStreamSynchronize(last_user);            // no event to be had: wait for that one stream
        }
        old[n++] = Old{ p, ev, host };
    }
};
struct DeviceScratch {                  // device memory
    void* p = nullptr; size_t cap = 0; hipStream_t last_user = nullptr; RetireList retired;
    // `stream`: where the caller will queue the work that reads / writes the buffer
    void* get(size_t n, hipStream_t stream = nullptr)
    {
        retired.reap(false);
        if (n > cap) {
            if (p) { retired.park(p, false, last_user); p = nullptr; cap = 0; }
            const size_t want = n + n / 4 + 4096;
            if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return nullptr; }
            cap = want;
        }
        last_user = stream;
        return p;
    }
};
struct PinnedScratch {                  // page-locked host memory: uploads from it are plain DMA
    uint8_t* p = nullptr; size_t cap = 0; hipStream_t last_user = nullptr; RetireList retired;
    uint8_t* get(size_t n, hipStream_t stream = nullptr)
    {
        retired.reap(false);
        if (n > cap) {
            if (p) { retired.park(p, true, last_user); p = nullptr; cap = 0; }
            void* np = nullptr;
            const size_t want = n + n / 4 + 4096;
            if (hipHostMalloc(&np, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            p = (uint8_t*)np; cap = want;
        }
        last_user = stream;
        return p;
    }
};

// Is [p, p + n) page-locked host memory the device can read by DMA as it stands (hipHostMalloc / gamut_hip_host_malloc_pinned /
// hipHostRegister)?  The file-level batch calls then skip their staging copy: a caller that reads its files into pinned buffers pays
// PCIe only.  GAMUT_HIP_PINNED_INPUTS=0 turns the test off (measurements).
inline bool host_range_is_pinned(const void* p, size_t n)
{
    static const bool off = [] { const char* e = getenv("GAMUT_HIP_PINNED_INPUTS"); return e && *e && atoi(e) == 0; }();
    if (off || !p || !n) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }      // plain pageable memory: "invalid value"
    if (at.type != hipMemoryTypeHost) return false;
    hipPointerAttribute_t at2;                                // the last byte too: inside the same kind of memory (a file is one allocation's worth)
    if (hipPointerGetAttributes(&at2, static_cast<const uint8_t*>(p) + n - 1) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at2.type == hipMemoryTypeHost;
}

// Host threads worth starting: the hardware threads, or fewer when a cgroup CPU quota (cpu.max) grants the process less --
// the GPU boxes show 256 threads and a quota of 16 cores, and 256 inflating threads on 16 cores' worth of time run slower
// than 16.
int host_threads();

// fn(worker, index) for index in [0, count) on `workers` host threads (the caller is worker 0); dynamic distribution.  The helpers
// are tasks on a persistent pool of the library (runtime.hip): round 3 started and joined workers - 1 std::threads per call -- ~0.5 ms
// of thread creation per call on the GPU box, five calls per JPEG batch -- and every short-lived thread paid for its thread-local
// staging again.  A helper that arrives after the indices are gone returns at once, so calls from several caller threads (the three
// legs of gamut_hip_decode_batch_device) share the pool without waiting for one another's helpers to START: a call returns when its
// own indices are done and its own helpers have left.
void pool_submit(void (*run)(void*, int), void* ctx, int first_worker, int n_helpers, std::atomic<int>* left);     // never throws
void pool_cancel(std::atomic<int>* left);                  // takes back the caller's tasks that no helper has started
template <class Fn> void parallel_for(int count, int workers, Fn fn)          // fn(worker, index); the caller runs worker 0
{
    struct Ctx { std::atomic<int> next{ 0 }; int count; Fn* fn; } c;
    c.count = count; c.fn = &fn;
    auto run = [](void* p, int w) { Ctx& x = *static_cast<Ctx*>(p); for (int i; (i = x.next.fetch_add(1, std::memory_order_relaxed)) < x.count; ) (*x.fn)(w, i); };
    std::atomic<int> left{ 0 };
    const int helpers = workers - 1 < count - 1 ? workers - 1 : count - 1;
    if (helpers > 0) pool_submit(run, &c, 1, helpers, &left);
    run(&c, 0);
    if (left.load(std::memory_order_acquire) > 0) pool_cancel(&left);
    while (left.load(std::memory_order_acquire) > 0) std::this_thread::yield();      // (the helpers still hold references to c and fn)
}

#define GAMUT_HIP_CHECK(expr)                                                                 \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess)                                                                \
            return ::gamut::set_error(GAMUT_HIP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

inline int launch_status(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GAMUT_HIP_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
    return GAMUT_HIP_OK;
}

// PixelType tables (types.d:62-86, internals/types.d:44-96)
constexpr int kPixelSize[GAMUT_PIXEL_COUNT]     = { 1,2,4, 2,4,8, 2,4,8, 3,6,12, 4,8,16, 4,8,16 };
constexpr int kPixelChannels[GAMUT_PIXEL_COUNT] = { 1,1,1, 2,2,2, 2,2,2, 3,3,3, 4,4,4, 4,4,4 };
inline bool valid_type(int t) { return t >= 0 && t < GAMUT_PIXEL_COUNT; }

// kernel launchers implemented in the .hip files
int convert_device(int srcType, const void* src, int64_t srcPitch, int64_t srcLayerOffset,
                   int dstType, void* dst, int64_t dstPitch, int64_t dstLayerOffset,
                   int width, int height, int layers, hipStream_t stream);

int flip_device(int type, void* data, int64_t pitch, int64_t layer_off, int w, int h, int layers, int vertical, hipStream_t st);   // flip.hip

int jpeg_reconstruct_launch(const int16_t* coeffs, int64_t coeff_stride,
                            const uint8_t* max_zag, int64_t zag_stride,
                            uint8_t* out, int64_t out_pitch, int64_t out_stride,
                            int width, int height, int scan_type, int out_comps,
                            int count, hipStream_t stream);

int jpeg_reconstruct_tokens_launch(const uint32_t* tokens, const uint32_t* strip_tab, const int64_t* tok_offs, const int64_t* strip_offs,
                                   const uint8_t* max_zag, int64_t zag_stride, uint8_t* out, int64_t out_pitch, int64_t out_stride,
                                   int width, int height, int out_comps, int count, hipStream_t stream);

// inflate.hip: DEFLATE streams in HBM -> bytes in HBM, one workgroup per stream (asynchronous on `stream`)
int inflate_launch(const gamut_hip_inflate_desc* descs, int count, uint32_t* out_len_dev, uint32_t* status_dev, hipStream_t stream);
// the same while the streams are still being uploaded: begin, then a step per slice (avail_host[i] = bytes of stream i that are in HBM once
// `stream` gets there), the last step with every stream whole; out_len / status are final after the last step
int inflate_sliced_begin(const gamut_hip_inflate_desc* descs, int count, hipStream_t stream);
int inflate_sliced_step(int count, const uint32_t* avail_host, uint32_t* out_len_dev, uint32_t* status_dev, hipStream_t stream);
int png_defilter_launch(const uint8_t* raw, int64_t raw_stride, uint32_t raw_len,
                        uint8_t* out, int64_t out_stride,
                        uint32_t x, uint32_t y, int img_n, int out_n, int depth, int color,
                        int count, uint32_t* status, hipStream_t stream,
                        const int64_t* raw_offs = nullptr, const int64_t* out_offs = nullptr, bool offs_dword_aligned = false, bool offs_line_aligned = false);

int png_transparency_launch(void* img, int64_t npx, int out_n, int depth16, const uint16_t tc[3], hipStream_t st);
int png_palette_launch(const uint8_t* idx, uint8_t* out, int64_t npx, int pal_n, const uint8_t* palette_dev, hipStream_t st);
int png_convert_format_launch(const void* src, void* dst, int64_t npx, int img_n, int req, int depth16, hipStream_t st);
int png_depth_convert_launch(const void* src, void* dst, int64_t n, int to16, hipStream_t st);
int png_adam7_scatter_launch(const uint8_t* pass, uint8_t* final_, uint32_t px, uint32_t py, uint32_t img_x, int out_bytes, int p, hipStream_t st);

} // namespace gamut
