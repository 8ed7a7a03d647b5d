/* rccl_double.c -- a TEST DOUBLE of the eight RCCL entry points gamut_amd/csrc/comm.hip binds, for ranks that are THREADS OF ONE
 * PROCESS ON ONE DEVICE (real RCCL refuses two ranks on one device, and no box this suite has met had two).  It exists so that
 * gamut_hip_gather_outputs_device's multi-rank code -- which image goes to whom, grouped ncclSend / ncclRecv, groups of 256 -- runs
 * on a 1-GPU box: the library loads it through GAMUT_HIP_RCCL_LIB instead of librccl.  Test infrastructure, never shipped.
 *
 * Semantics kept from NCCL 2.x point-to-point: a Send to `peer` matches the peer's Recv from this rank in program order per
 * (source, destination) pair; both sides must call; operations inside ncclGroupStart / ncclGroupEnd are issued together (so a rank
 * may post sends and receives to the same peer in one group without deadlock); completion is stream-ordered -- ncclGroupEnd returns
 * once the copies are ENQUEUED, the data is there when the receiver's stream gets there, and the sender's stream does not pass the
 * call before its buffers have been read.  Bytes move with hipMemcpyAsync device-to-device on the receiver's stream.
 *   gcc -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include rccl_double.c -o librccl_double.so -L/opt/rocm/lib -lamdhip64 -lpthread */
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };

enum { MAX_RANKS = 64 };
typedef struct Posted {                 /* a send that waits for its receive */
    const void* src; size_t bytes; hipEvent_t ready;      /* recorded on the sender's stream: the data is final behind it */
    hipEvent_t taken; int matched;                         /* recorded on the receiver's stream behind its copy */
    struct Posted* next;
} Posted;
typedef struct World {
    char id[128]; int nranks, joined, left;
    Posted* head[MAX_RANKS][MAX_RANKS]; Posted* tail[MAX_RANKS][MAX_RANKS];      /* [source][destination]: FIFO */
    struct World* next;
} World;
struct ncclComm { World* w; int rank; };
typedef struct ncclComm* ncclComm_t;

static pthread_mutex_t g_m = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_cv = PTHREAD_COND_INITIALIZER;
static World* g_worlds;
static unsigned g_next_id = 1;

typedef struct { int send; void* buf; size_t bytes; int peer; ncclComm_t comm; hipStream_t stream; Posted* posted; } Op;
static __thread int t_depth;
static __thread Op* t_ops; static __thread int t_n, t_cap;

static size_t type_size(int t) { static const size_t s[] = { 1, 1, 4, 4, 8, 8, 2, 4, 8, 2 }; return t >= 0 && t < 10 ? s[t] : 0; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    pthread_mutex_lock(&g_m);
    const unsigned n = g_next_id++;
    pthread_mutex_unlock(&g_m);
    memcpy(id->internal, "rccl-double", 11); memcpy(id->internal + 16, &n, sizeof(n));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks || memcmp(id.internal, "rccl-double", 11)) return ncclInvalidArgument;
    pthread_mutex_lock(&g_m);
    World* w = g_worlds;
    while (w && memcmp(w->id, id.internal, 128)) w = w->next;
    if (!w) {
        w = (World*)calloc(1, sizeof(World));
        if (!w) { pthread_mutex_unlock(&g_m); return ncclSystemError; }
        memcpy(w->id, id.internal, 128); w->nranks = nranks; w->next = g_worlds; g_worlds = w;
    }
    if (w->nranks != nranks) { pthread_mutex_unlock(&g_m); return ncclInvalidArgument; }
    ++w->joined;
    pthread_cond_broadcast(&g_cv);
    while (w->joined < nranks) pthread_cond_wait(&g_cv, &g_m);              /* collective, like the real call */
    pthread_mutex_unlock(&g_m);
    ncclComm_t c = (ncclComm_t)calloc(1, sizeof(*c));
    if (!c) return ncclSystemError;
    c->w = w; c->rank = rank; *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclInvalidArgument;
    free(comm);                                                              /* (worlds are a few KB each and stay: tests are short) */
    return ncclSuccess;
}

static ncclResult_t run_ops(Op* ops, int n)
{
    ncclResult_t rc = ncclSuccess;
    /* 1: post every send of the group (no waiting: whoever receives finds them) */
    for (int i = 0; i < n; ++i) {
        Op* o = &ops[i];
        if (!o->send) continue;
        Posted* p = (Posted*)calloc(1, sizeof(Posted));
        if (!p) return ncclSystemError;
        p->src = o->buf; p->bytes = o->bytes;
        if (hipEventCreateWithFlags(&p->ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&p->taken, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(p->ready, o->stream) != hipSuccess) return ncclUnhandledCudaError;
        o->posted = p;
        World* w = o->comm->w; const int s = o->comm->rank, d = o->peer;
        pthread_mutex_lock(&g_m);
        if (w->tail[s][d]) w->tail[s][d]->next = p; else w->head[s][d] = p;
        w->tail[s][d] = p;
        pthread_cond_broadcast(&g_cv);
        pthread_mutex_unlock(&g_m);
    }
    /* 2: every receive takes the oldest send of its pair */
    for (int i = 0; i < n; ++i) {
        Op* o = &ops[i];
        if (o->send) continue;
        World* w = o->comm->w; const int s = o->peer, d = o->comm->rank;
        pthread_mutex_lock(&g_m);
        while (!w->head[s][d]) pthread_cond_wait(&g_cv, &g_m);
        Posted* p = w->head[s][d];
        w->head[s][d] = p->next; if (!p->next) w->tail[s][d] = NULL;
        pthread_mutex_unlock(&g_m);
        if (p->bytes != o->bytes) rc = ncclInvalidArgument;                  /* (real NCCL: undefined; here: said) */
        else if (hipStreamWaitEvent(o->stream, p->ready, 0) != hipSuccess ||
                 (p->bytes && hipMemcpyAsync(o->buf, p->src, p->bytes, hipMemcpyDeviceToDevice, o->stream) != hipSuccess)) rc = ncclUnhandledCudaError;
        if (hipEventRecord(p->taken, o->stream) != hipSuccess) rc = ncclUnhandledCudaError;
        pthread_mutex_lock(&g_m);
        p->matched = 1;
        pthread_cond_broadcast(&g_cv);
        pthread_mutex_unlock(&g_m);
    }
    /* 3: the sender's stream stays behind the copies that read its buffers */
    for (int i = 0; i < n; ++i) {
        Op* o = &ops[i];
        if (!o->send || !o->posted) continue;
        Posted* p = o->posted;
        pthread_mutex_lock(&g_m);
        while (!p->matched) pthread_cond_wait(&g_cv, &g_m);
        pthread_mutex_unlock(&g_m);
        if (hipStreamWaitEvent(o->stream, p->taken, 0) != hipSuccess) rc = ncclUnhandledCudaError;
        (void)hipEventDestroy(p->ready); (void)hipEventDestroy(p->taken);    /* (destruction is deferred by the runtime until the events have passed) */
        free(p);
    }
    return rc;
}

static ncclResult_t add_op(int send, void* buf, size_t count, int type, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || peer < 0 || peer >= comm->w->nranks || peer == comm->rank || !type_size(type) || (count && !buf)) return ncclInvalidArgument;
    Op o = { send, buf, count * type_size(type), peer, comm, stream, NULL };
    if (t_depth == 0) return run_ops(&o, 1);
    if (t_n == t_cap) {
        const int cap = t_cap ? 2 * t_cap : 64;
        Op* g = (Op*)realloc(t_ops, (size_t)cap * sizeof(Op));
        if (!g) return ncclSystemError;
        t_ops = g; t_cap = cap;
    }
    t_ops[t_n++] = o;
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, int type, int peer, ncclComm_t comm, hipStream_t stream) { return add_op(1, (void*)buf, count, type, peer, comm, stream); }
ncclResult_t ncclRecv(void* buf, size_t count, int type, int peer, ncclComm_t comm, hipStream_t stream) { return add_op(0, buf, count, type, peer, comm, stream); }
ncclResult_t ncclGroupStart(void) { ++t_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void)
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth) return ncclSuccess;
    const ncclResult_t rc = run_ops(t_ops, t_n);
    t_n = 0;
    return rc;
}
const char* ncclGetErrorString(ncclResult_t rc)
{
    static const char* s[] = { "no error", "unhandled HIP error", "system error", "internal error", "invalid argument", "invalid usage" };
    return rc >= 0 && rc < 6 ? s[rc] : "unknown error";
}
