// comm.hip -- multi-GPU for a C / D host: image-index sharding and the one exchange the path has, the gather of decoded
// outputs over RCCL / xGMI (BASELINE.json north_star).  One process per GPU.
//
//   gamut_hip_shard_owner / _count / _local_index   image i lives on rank i % world (round-robin); no data-path collective
//   gamut_hip_comm_*                                an RCCL communicator behind an opaque handle: rank 0 makes a 128-byte id,
//                                                   the host passes it to the other ranks by whatever it has (file, pipe, MPI)
//   gamut_hip_gather_outputs_device                 every rank's decoded images -> their slots of the full batch on `root`
//                                                   (or on every rank), as grouped ncclSend / ncclRecv per image
//
// librccl is loaded at run time (dlopen; an already loaded copy, e.g. the one PyTorch carries, is reused): the library has no
// link-time dependency on it and the single-GPU path never touches it.  xGMI is point to point (7 links x ~153 GB/s per GPU):
// a gather into one root is bound by the root's ingress, an all-gather by a link -- both far below the decode rate, so the
// exchange is a separate call the host overlaps or skips ("replicas only" when outputs are consumed where they are produced).
#include "common.hpp"
#include <dlfcn.h>
#include <mutex>

namespace gamut {
namespace {

// the part of rccl.h this file uses (ROCm 7.2 /opt/rocm/include/rccl/rccl.h: stable NCCL 2.x ABI)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                                         // ncclSuccess = 0
constexpr int ncclUint8 = 1;                                      // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = "symbols missing";                      // the loader's message, captured once (dlerror() clears itself)
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        // GAMUT_HIP_RCCL_LIB: the library to bind instead (a site's own RCCL build; tests/c/rccl_double.c, which lets the multi-rank
        // gather run with ranks that are threads on one device).  Set: that file or nothing -- no silent fall-back to another copy.
        const char* forced = getenv("GAMUT_HIP_RCCL_LIB");
        if (forced && *forced) r.h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        else {
            for (const char* n : names) if ((r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;      // reuse a loaded copy (PyTorch's)
            if (!r.h) for (const char* n : names) if ((r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        }
        if (!r.h) { const char* e = dlerror(); if (e) snprintf(r.why, sizeof(r.why), "%s", e); return; }
#define GAMUT_SYM(field, name) *(void**)(&r.field) = dlsym(r.h, name)
        GAMUT_SYM(GetUniqueId, "ncclGetUniqueId"); GAMUT_SYM(CommInitRank, "ncclCommInitRank"); GAMUT_SYM(CommDestroy, "ncclCommDestroy");
        GAMUT_SYM(Send, "ncclSend"); GAMUT_SYM(Recv, "ncclRecv"); GAMUT_SYM(GroupStart, "ncclGroupStart"); GAMUT_SYM(GroupEnd, "ncclGroupEnd");
        GAMUT_SYM(GetErrorString, "ncclGetErrorString");
#undef GAMUT_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Send && r.Recv && r.GroupStart && r.GroupEnd;
    });
    return r;
}

int need_rccl()
{
    if (!rccl().ok) return set_error(GAMUT_HIP_ERR_UNSUPPORTED, "librccl could not be loaded (%s)", rccl().why);
    return GAMUT_HIP_OK;
}
int nccl_fail(const char* what, ncclResult_t rc)
{
    return set_error(GAMUT_HIP_ERR_HIP, "%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error");
}

} // namespace
} // namespace gamut

using namespace gamut;

struct gamut_hip_comm { int world, rank; ncclComm_t nccl; };

extern "C" {

int gamut_hip_shard_owner(int64_t image_index, int world) { return world > 0 && image_index >= 0 ? (int)(image_index % world) : -1; }
int64_t gamut_hip_shard_count(int rank, int world, int64_t total_images)
{
    if (world <= 0 || rank < 0 || rank >= world || total_images < 0) return -1;
    return total_images / world + (rank < total_images % world ? 1 : 0);
}
int64_t gamut_hip_shard_local_index(int64_t image_index, int world) { return world > 0 && image_index >= 0 ? image_index / world : -1; }
int64_t gamut_hip_shard_global_index(int64_t local_index, int rank, int world) { return world > 0 && local_index >= 0 ? local_index * world + rank : -1; }

int gamut_hip_comm_get_unique_id(void* id128)
{
    clear_error();
    if (!id128) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "comm_get_unique_id: null pointer");
    if (int rc = need_rccl()) return rc;
    ncclUniqueId id;
    if (ncclResult_t rc = rccl().GetUniqueId(&id)) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(id128, id.internal, sizeof(id.internal));
    return GAMUT_HIP_OK;
}

int gamut_hip_comm_init(gamut_hip_comm** comm, int world, int rank, const void* id128)
{
    clear_error();
    if (!comm || world < 1 || rank < 0 || rank >= world) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "comm_init: bad arguments");
    *comm = nullptr;
    gamut_hip_comm* c = (gamut_hip_comm*)calloc(1, sizeof(gamut_hip_comm));
    if (!c) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "comm_init: out of memory");
    c->world = world; c->rank = rank;
    if (world > 1) {                                              // a single rank needs no communicator (and no librccl)
        if (!id128) { free(c); return set_error(GAMUT_HIP_ERR_INVALID_ARG, "comm_init: null id"); }
        if (int rc = need_rccl()) { free(c); return rc; }
        ncclUniqueId id; memcpy(id.internal, id128, sizeof(id.internal));
        if (ncclResult_t rc = rccl().CommInitRank(&c->nccl, world, id, rank)) { free(c); return nccl_fail("ncclCommInitRank", rc); }
    }
    *comm = c;
    return GAMUT_HIP_OK;
}

void gamut_hip_comm_destroy(gamut_hip_comm* comm)
{
    if (!comm) return;
    if (comm->nccl) (void)rccl().CommDestroy(comm->nccl);
    free(comm);
}
int gamut_hip_comm_rank(const gamut_hip_comm* comm) { return comm ? comm->rank : -1; }
int gamut_hip_comm_world(const gamut_hip_comm* comm) { return comm ? comm->world : -1; }

int gamut_hip_gather_outputs_device(gamut_hip_comm* comm, const void* local, int64_t local_stride, int64_t bytes_per_image,
                                    int64_t total_images, void* dst, int64_t dst_stride, int root, void* stream)
{
    clear_error();
    if (!comm || bytes_per_image < 0 || total_images < 0 || root >= comm->world || local_stride < bytes_per_image || dst_stride < bytes_per_image)
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "gather_outputs: bad arguments");
    const int world = comm->world, rank = comm->rank;
    const bool receives = root < 0 || root == rank;
    const int64_t mine = gamut_hip_shard_count(rank, world, total_images);
    if ((mine > 0 && !local) || (receives && total_images > 0 && !dst)) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "gather_outputs: null buffer");
    hipStream_t st = pick_stream(stream);
    if (bytes_per_image == 0 || total_images == 0) return GAMUT_HIP_OK;
    // own images: device-to-device, strided (round-robin slots k * world + rank of the destination)
    if (receives && mine > 0)
        GAMUT_HIP_CHECK(hipMemcpy2DAsync((uint8_t*)dst + (int64_t)rank * dst_stride, (size_t)(dst_stride * world), local, (size_t)local_stride,
                                         (size_t)bytes_per_image, (size_t)mine, hipMemcpyDeviceToDevice, st));
    if (world == 1) return GAMUT_HIP_OK;
    // the others: one ncclSend / ncclRecv per image, grouped (RCCL fuses a group's operations per peer); groups are kept to a
    // few hundred operations so that neither side queues unbounded work
    const Rccl& R = rccl();
    const int64_t kGroup = 256;
    for (int64_t i0 = 0; i0 < total_images; i0 += kGroup) {
        const int64_t i1 = i0 + kGroup < total_images ? i0 + kGroup : total_images;
        if (ncclResult_t rc = R.GroupStart()) return nccl_fail("ncclGroupStart", rc);
        ncclResult_t bad = 0;
        for (int64_t i = i0; i < i1 && !bad; ++i) {
            const int owner = (int)(i % world);
            if (owner == rank) {                                   // my image: to the root, or to everybody else
                const uint8_t* src = (const uint8_t*)local + (i / world) * local_stride;
                for (int peer = 0; peer < world && !bad; ++peer) {
                    if (peer == rank || (root >= 0 && peer != root)) continue;
                    bad = R.Send(src, (size_t)bytes_per_image, ncclUint8, peer, comm->nccl, st);
                }
            } else if (receives) {
                bad = R.Recv((uint8_t*)dst + i * dst_stride, (size_t)bytes_per_image, ncclUint8, owner, comm->nccl, st);
            }
        }
        const ncclResult_t end = R.GroupEnd();
        if (bad) return nccl_fail("ncclSend / ncclRecv", bad);
        if (end) return nccl_fail("ncclGroupEnd", end);
    }
    return GAMUT_HIP_OK;
}

int gamut_hip_host_threads(void) { return host_threads(); }

} // extern "C"
