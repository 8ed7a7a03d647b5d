// batch_host.hip -- ONE batch call for files of any of the three formats (BASELINE.json config 5 from files).
//
// The reference takes any file: Image.loadFromMemory -> identifyFormatFromStream (image.d:1045-1061: the plugins' detect procedures,
// a signature test each -- plugins/jpeg.d:106-110, png.d:165-169, qoi.d:143-147) -> g_plugins[fif].loadProc (image.d:1751-1772).
// The batched surface had three per-format calls that a caller had to sort its files into and that tools/e2e_mixed_bench.py ran one
// after the other -- 3072 files: 13 + 117 + 112 ms -- although the three pipelines lean on different resources at different times:
// the PNG leg is bound by the inflate kernels while the QOI leg is bound by PCIe (5 MB files), and the JPEG leg is short.
// gamut_hip_decode_batch_device sniffs every file's format the way the reference does and runs the three pipelines SIDE BY SIDE:
// each on a worker thread of the library (the per-format calls keep their staging buffers per thread, so the workers are persistent)
// and on a stream of its own behind the caller's stream.  Pixels land at out + out_offset[i] in the caller's order.
#include "common.hpp"
#include <condition_variable>
#include <functional>
#include <mutex>

namespace gamut {
namespace {

// One persistent helper thread per format pipeline.  Jobs are handed over one at a time; the thread never exits (its thread-local
// staging buffers must not die with it: they are device / pinned memory that is deliberately not released at thread exit).
struct Worker {
    std::mutex m; std::condition_variable cv;
    std::function<void()> job; bool pending = false, done = false, started = false;
    void ensure()
    {
        if (started) return;
        std::thread([this] {
            for (;;) {
                std::function<void()> j;
                { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [this] { return pending; }); j = std::move(job); pending = false; }
                try { j(); } catch (...) { }                 // (a leg reports through its LegResult; nothing may leave a detached thread)
                { std::lock_guard<std::mutex> lk(m); done = true; }
                cv.notify_all();
            }
        }).detach();
        started = true;                                          // (only once the thread exists: a failed start throws and is tried again)
    }
    void submit(std::function<void()> j) { { std::lock_guard<std::mutex> lk(m); job = std::move(j); pending = true; done = false; } cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [this] { return done; }); }
};
// A pool per DEVICE: a host that drives its GPUs from a thread each ("one host thread + one HIP stream per GPU") keeps
// the three-way overlap on every one of them (one process-wide pool let the first caller overlap its legs and made the seven others run
// theirs one after the other).  Two threads on ONE device: the second finds the pool taken and runs its legs in turn, as before.
struct Pool { std::mutex busy; Worker w[2]; };
Pool& pool(int dev)                                             // never destroyed: the threads outlive main()
{
    static std::mutex m;
    static Pool* pools[64] = { nullptr };
    std::lock_guard<std::mutex> lk(m);
    Pool*& p = pools[dev < 0 || dev >= 64 ? 0 : dev];
    if (!p) p = new Pool();
    return *p;
}
// What a worker runs refers to the caller's frame (leg[], res[]): the caller must not leave that frame -- by return or by exception --
// while a worker still runs.  This waits for every job that was handed over, on every way out.
struct SubmittedJobs {
    Pool& P; int n = 0;
    explicit SubmittedJobs(Pool& p) : P(p) {}
    ~SubmittedJobs() { for (int k = 0; k < n; ++k) P.w[k].wait(); }
};

struct LegResult { int rc = GAMUT_HIP_OK; char msg[256] = { 0 }; };

} // namespace
} // namespace gamut

using namespace gamut;

extern "C" int gamut_hip_identify_format(const uint8_t* b, size_t len)
{
    static const uint8_t png[8] = { 0x89, 0x50, 0x4e, 0x47, 0x0d, 0x0a, 0x1a, 0x0a };
    if (b && len >= 2 && b[0] == 0xFF && b[1] == 0xD8) return GAMUT_HIP_FORMAT_JPEG;        // detectJPEG plugins/jpeg.d:106-110
    if (b && len >= 8 && !memcmp(b, png, 8)) return GAMUT_HIP_FORMAT_PNG;                   // detectPNG plugins/png.d:165-169
    if (b && len >= 4 && !memcmp(b, "qoif", 4)) return GAMUT_HIP_FORMAT_QOI;                // detectQOI plugins/qoi.d:143-147
    return GAMUT_HIP_FORMAT_UNKNOWN;
}

extern "C" int gamut_hip_decode_batch_device(const uint8_t* const* data, const size_t* len, int count, int req_comps,
                                             const int64_t* out_offset, uint8_t* out, gamut_hip_image_info* info, int* status_host, void* stream)
{
    clear_error();
    if (count < 0 || (req_comps != 3 && req_comps != 4) || (count > 0 && (!data || !len || !out_offset || !out || !info)))
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "decode_batch_device: bad arguments (req_comps is 3 or 4: what all three decoders produce)");
    if (count == 0) return GAMUT_HIP_OK;
    int ndev = 0, dev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_error(GAMUT_HIP_ERR_NO_DEVICE, "no HIP device available (libgamut_hip has no CPU fallback)");
    (void)hipGetDevice(&dev);
    try {
        // the files by format, in the caller's order
        std::vector<int> idx[3];
        std::vector<int> own_status;
        int* hst = status_host;
        if (!hst) { own_status.assign((size_t)count, GAMUT_HIP_OK); hst = own_status.data(); }
        for (int i = 0; i < count; ++i) {
            memset(&info[i], 0, sizeof(info[i]));
            const int f = gamut_hip_identify_format(data[i], len[i]);
            info[i].format = f;
            hst[i] = f < 0 ? GAMUT_HIP_ERR_UNSUPPORTED : GAMUT_HIP_OK;                       // kStrImageFormatUnidentified (image.d:1758-1762)
            if (f >= 0) idx[f].push_back(i);
        }
        struct Leg {
            std::vector<const uint8_t*> ptr; std::vector<size_t> len; std::vector<int> isize; std::vector<int64_t> off; std::vector<int> st;
            std::vector<gamut_hip_jpeg_frame> jf; std::vector<gamut_hip_png_info> pf; std::vector<gamut_hip_qoi_desc> qf;
        } leg[3];
        for (int f = 0; f < 3; ++f) {
            const size_t n = idx[f].size();
            leg[f].ptr.resize(n); leg[f].len.resize(n); leg[f].off.resize(n); leg[f].st.assign(n, GAMUT_HIP_OK);
            for (size_t k = 0; k < n; ++k) { const int i = idx[f][k]; leg[f].ptr[k] = data[i]; leg[f].len[k] = len[i]; leg[f].off[k] = out_offset[i]; }
        }
        leg[GAMUT_HIP_FORMAT_JPEG].jf.resize(idx[GAMUT_HIP_FORMAT_JPEG].size());
        leg[GAMUT_HIP_FORMAT_PNG].pf.resize(idx[GAMUT_HIP_FORMAT_PNG].size());
        leg[GAMUT_HIP_FORMAT_QOI].qf.resize(idx[GAMUT_HIP_FORMAT_QOI].size());
        leg[GAMUT_HIP_FORMAT_QOI].isize.resize(idx[GAMUT_HIP_FORMAT_QOI].size());
        for (size_t k = 0; k < idx[GAMUT_HIP_FORMAT_QOI].size(); ++k) {
            const size_t n = leg[GAMUT_HIP_FORMAT_QOI].len[k];
            leg[GAMUT_HIP_FORMAT_QOI].isize[k] = n > 0x7fffffffu ? 0x7fffffff : (int)n;     // qoi_decode takes an int size (qoi.d:448)
        }

        // every leg behind what the caller's stream holds now, on a stream of its own
        hipStream_t st = pick_stream(stream);
        static thread_local PerDevice<hipEvent_t> fork_pd;
        hipEvent_t& fork = fork_pd.cur();
        if (!fork) GAMUT_HIP_CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        GAMUT_HIP_CHECK(hipEventRecord(fork, st));

        LegResult res[3];
        auto run_leg = [&](int f, bool own_stream) {
            LegResult& r = res[f];
            if (idx[f].empty()) return;
            if (hipSetDevice(dev) != hipSuccess) { r.rc = GAMUT_HIP_ERR_HIP; snprintf(r.msg, sizeof(r.msg), "decode_batch_device: hipSetDevice failed"); return; }
            hipStream_t ls = st;
            if (own_stream) {
                static thread_local PerDevice<hipStream_t> leg_stream_pd;
                hipStream_t& s = leg_stream_pd.cur();
                if (!s && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { r.rc = GAMUT_HIP_ERR_HIP; snprintf(r.msg, sizeof(r.msg), "decode_batch_device: stream creation failed"); return; }
                if (hipStreamWaitEvent(s, fork, 0) != hipSuccess) { r.rc = GAMUT_HIP_ERR_HIP; snprintf(r.msg, sizeof(r.msg), "decode_batch_device: hipStreamWaitEvent failed"); return; }
                ls = s;
            }
            Leg& L = leg[f];
            const int n = (int)idx[f].size();
            if (f == GAMUT_HIP_FORMAT_JPEG)
                r.rc = gamut_hip_jpeg_decode_batch_device(L.ptr.data(), L.len.data(), n, req_comps, L.off.data(), out, L.jf.data(), L.st.data(), nullptr, ls);
            else if (f == GAMUT_HIP_FORMAT_PNG)
                r.rc = gamut_hip_png_decode_batch_device(L.ptr.data(), L.len.data(), n, req_comps, 8, L.off.data(), out, L.pf.data(), L.st.data(), 0, ls);
            else
                r.rc = gamut_hip_qoi_decode_batch_device(L.ptr.data(), L.isize.data(), n, req_comps, L.off.data(), out, L.qf.data(), L.st.data(), ls);
            if (r.rc != GAMUT_HIP_OK) snprintf(r.msg, sizeof(r.msg), "%s", gamut_hip_last_error());
            if (own_stream) (void)hipStreamSynchronize(ls);                                   // (the per-format calls return when their pixels are in place)
        };
        // The longest legs first on the workers (PNG: inflate-bound, QOI: PCIe-bound), JPEG on the calling thread.  When another
        // thread's mixed batch holds the workers, the legs run one after the other here: same results.
        static const bool serial = [] { const char* e = getenv("GAMUT_HIP_MIXED_SERIAL"); return e && *e && atoi(e) != 0; }();     // measurements
        Pool& P = pool(dev);
        const int legs = (int)!idx[0].empty() + (int)!idx[1].empty() + (int)!idx[2].empty();
        if (legs > 1 && !serial && P.busy.try_lock()) {
            std::lock_guard<std::mutex> hold(P.busy, std::adopt_lock);
            SubmittedJobs jobs(P);                                                            // (declared after the lock: waits before it is released)
            for (int f : { (int)GAMUT_HIP_FORMAT_PNG, (int)GAMUT_HIP_FORMAT_QOI }) {
                if (idx[f].empty()) continue;
                P.w[jobs.n].ensure();
                P.w[jobs.n].submit([&, f] {                                                   // (anything a leg throws -- bad_alloc from a per-device slot -- is that leg's verdict:
                    try { run_leg(f, true); }                                                 //  swallowed by the worker it would have left rc == OK over pixels never written)
                    catch (...) { res[f].rc = GAMUT_HIP_ERR_OUT_OF_MEMORY; snprintf(res[f].msg, sizeof(res[f].msg), "decode_batch_device: out of host memory in the %s leg", f == GAMUT_HIP_FORMAT_PNG ? "PNG" : "QOI"); }
                });
                ++jobs.n;                                                                     // only what was really handed over is waited for
            }
            run_leg(GAMUT_HIP_FORMAT_JPEG, true);
        } else {
            for (int f = 0; f < 3; ++f) run_leg(f, false);
        }
        (void)hipSetDevice(dev);

        // results back into the caller's order
        for (size_t k = 0; k < idx[GAMUT_HIP_FORMAT_JPEG].size(); ++k) {
            const int i = idx[GAMUT_HIP_FORMAT_JPEG][k]; const gamut_hip_jpeg_frame& f = leg[GAMUT_HIP_FORMAT_JPEG].jf[k];
            info[i].width = f.width; info[i].height = f.height; info[i].channels_in_file = f.comps; info[i].channels = req_comps;
            hst[i] = leg[GAMUT_HIP_FORMAT_JPEG].st[k];
        }
        for (size_t k = 0; k < idx[GAMUT_HIP_FORMAT_PNG].size(); ++k) {
            const int i = idx[GAMUT_HIP_FORMAT_PNG][k]; const gamut_hip_png_info& f = leg[GAMUT_HIP_FORMAT_PNG].pf[k];
            info[i].width = (int)f.width; info[i].height = (int)f.height; info[i].channels_in_file = f.channels_in_file; info[i].channels = req_comps;
            hst[i] = leg[GAMUT_HIP_FORMAT_PNG].st[k];
        }
        for (size_t k = 0; k < idx[GAMUT_HIP_FORMAT_QOI].size(); ++k) {
            const int i = idx[GAMUT_HIP_FORMAT_QOI][k]; const gamut_hip_qoi_desc& f = leg[GAMUT_HIP_FORMAT_QOI].qf[k];
            info[i].width = (int)f.width; info[i].height = (int)f.height; info[i].channels_in_file = f.channels; info[i].channels = req_comps;
            hst[i] = leg[GAMUT_HIP_FORMAT_QOI].st[k];
        }
        // a failure that is not a per-file verdict (allocation, HIP) is the call's; otherwise the lowest-numbered failing file's
        for (int f = 0; f < 3; ++f)
            if (res[f].rc != GAMUT_HIP_OK && res[f].rc != GAMUT_HIP_ERR_DECODE && res[f].rc != GAMUT_HIP_ERR_UNSUPPORTED && res[f].rc != GAMUT_HIP_ERR_INVALID_ARG)
                return set_error(res[f].rc, "%s", res[f].msg);
        for (int i = 0; i < count; ++i) {
            if (hst[i] == GAMUT_HIP_OK) continue;
            const int f = info[i].format;
            if (f < 0) return set_error(GAMUT_HIP_ERR_UNSUPPORTED, "image %d: format not identified (JPEG, PNG and QOI files are decoded)", i);
            return set_error(hst[i], "image %d: %s", i, res[f].msg[0] ? res[f].msg : "decoding failed");
        }
        return GAMUT_HIP_OK;
    } catch (...) {
        return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "decode_batch_device: out of host memory");
    }
}
