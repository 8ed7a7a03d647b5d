/* A pure-C host of the multi-GPU path, the way a D host would drive it: no Python, no torch.distributed.
 *   shard_count -> "decode" this rank's images on its GPU -> gather_outputs_device -> check every image of the batch.
 *
 *   shard_host proc    <world> <rank> <idfile>    one process per GPU; rank 0 writes the RCCL id to <idfile>, the others read it
 *   shard_host threads <world> [total [pad]]      one process, one host thread + one device per rank ("one host thread + one HIP
 *                                                  stream per GPU"); the id travels through memory.  total = images of the batch
 *                                                  (default 11; 0 and 1 are cases), pad = bytes between the images of the local and
 *                                                  of the gathered buffer beyond an image's own (strides != bytes_per_image; the
 *                                                  bytes in between must come back untouched)
 * With GAMUT_HIP_RCCL_LIB=<tests/c/rccl_double.c built> the threads form runs on ONE device: the gather's multi-rank code with a
 * stand-in for the transport (tests/test_stream_comm.py::test_gather_many_ranks_on_one_device).
 * world 1 needs no id and no RCCL.  Rank r uses device r % gamut_hip_device_count().  The "decode" is the path's own pixel
 * conversion (rgba8 -> rgba16 on the device, scanline.d: v * 257), so every expected byte is known on the host.
 * Built and run by tests/test_stream_comm.py (-m gpu): world 1 always; world = min(devices, 4) as processes AND as threads
 * whenever two or more devices are visible -- the first time RCCL sees more than one rank. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "gamut_hip.h"

enum { W = 64, H = 48, SRC_BYTES = W * H * 4, IMG_BYTES = W * H * 8 };
static int TOTAL = 11, PAD = 0;

static unsigned char src_byte(int image, int k) { return (unsigned char)((image * 131 + k * 7 + (k >> 8)) & 255); }

#define CHECK(call) do { const int rc__ = (call); if (rc__ != GAMUT_HIP_OK) { fprintf(stderr, "rank %d: %s -> %d (%s)\n", rank, #call, rc__, gamut_hip_last_error()); return 10; } } while (0)

static int run_rank(int world, int rank, const void* id128)
{
    const int ndev = gamut_hip_device_count();
    if (ndev < 1) { fprintf(stderr, "no device\n"); return 9; }
    CHECK(gamut_hip_init(rank % ndev));
    gamut_hip_comm* comm = NULL;
    CHECK(gamut_hip_comm_init(&comm, world, rank, id128));
    if (gamut_hip_comm_rank(comm) != rank || gamut_hip_comm_world(comm) != world) return 11;

    void* stream = gamut_hip_stream_create();
    const int64_t mine = gamut_hip_shard_count(rank, world, TOTAL);
    const size_t LSTRIDE = IMG_BYTES + (size_t)PAD, DSTRIDE = IMG_BYTES + (size_t)(PAD ? PAD + 16 : 0), all_bytes = (size_t)(TOTAL ? TOTAL : 1) * DSTRIDE;
    unsigned char* h_src = (unsigned char*)malloc((size_t)(mine ? mine : 1) * SRC_BYTES);
    unsigned char* h_all = (unsigned char*)malloc(all_bytes);
    void* d_src = gamut_hip_device_malloc((size_t)(mine ? mine : 1) * SRC_BYTES);
    void* d_loc = gamut_hip_device_malloc((size_t)(mine ? mine : 1) * LSTRIDE);
    void* d_all = gamut_hip_device_malloc(all_bytes);
    if (!h_src || !h_all || !d_src || !d_loc || !d_all) return 12;
    for (int64_t k = 0; k < mine; ++k) {
        const int image = (int)gamut_hip_shard_global_index(k, rank, world);
        if (gamut_hip_shard_owner(image, world) != rank || gamut_hip_shard_local_index(image, world) != k) return 13;
        for (int b = 0; b < SRC_BYTES; ++b) h_src[k * SRC_BYTES + b] = src_byte(image, b);
    }
    if (mine) {
        CHECK(gamut_hip_memcpy_h2d(d_src, h_src, (size_t)mine * SRC_BYTES, stream));
        /* this rank's share, one layered launch: layer k = its k-th image */
        CHECK(gamut_hip_scanlines_convert_device(GAMUT_PIXEL_rgba8, d_src, W * 4, SRC_BYTES, GAMUT_PIXEL_rgba16, d_loc, W * 8, (int64_t)LSTRIDE, W, H, (int)mine, stream));
    }
    const int roots[3] = { -1, 0, world - 1 };                   /* an all-gather, a gather to rank 0, a gather to the last rank */
    for (int t = 0; t < (world > 1 ? 3 : 2); ++t) {
        const int root = roots[t];
        memset(h_all, 0xEE, all_bytes);
        CHECK(gamut_hip_memcpy_h2d(d_all, h_all, all_bytes, stream));
        CHECK(gamut_hip_gather_outputs_device(comm, d_loc, (int64_t)LSTRIDE, IMG_BYTES, TOTAL, d_all, (int64_t)DSTRIDE, root, stream));
        CHECK(gamut_hip_memcpy_d2h(h_all, d_all, all_bytes, stream));
        CHECK(gamut_hip_stream_synchronize(stream));
        const int receives = root < 0 || rank == root;
        for (int image = 0; image < TOTAL; ++image) {
            const unsigned char* p = h_all + (size_t)image * DSTRIDE;
            for (int k = 0; k < SRC_BYTES; ++k) {
                const unsigned v = src_byte(image, k) * 257u;
                if (receives ? (p[2 * k] != (v & 255u) || p[2 * k + 1] != (v >> 8)) : (p[2 * k] != 0xEE || p[2 * k + 1] != 0xEE)) {
                    fprintf(stderr, "rank %d root %d: image %d sample %d %s\n", rank, root, image, k, receives ? "differs" : "was written on a rank that receives nothing"); return 14;
                }
            }
            for (size_t k = IMG_BYTES; k < DSTRIDE; ++k) if (p[k] != 0xEE) { fprintf(stderr, "rank %d root %d: the gap behind image %d was written\n", rank, root, image); return 15; }
        }
    }
    gamut_hip_comm_destroy(comm);
    gamut_hip_device_free(d_src); gamut_hip_device_free(d_loc); gamut_hip_device_free(d_all);
    gamut_hip_stream_destroy(stream);
    free(h_src); free(h_all);
    return 0;
}

struct thread_arg { int world, rank, rc; const void* id; };
static void* thread_main(void* p)
{
    struct thread_arg* a = (struct thread_arg*)p;
    a->rc = run_rank(a->world, a->rank, a->id);
    return NULL;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: shard_host proc <world> <rank> <idfile> | threads <world>\n"); return 2; }
    const int world = atoi(argv[2]);
    if (world < 1 || world > 64) return 2;
    static unsigned char id[GAMUT_HIP_COMM_ID_BYTES];
    if (!strcmp(argv[1], "proc")) {
        if (argc < 5) return 2;
        const int rank = atoi(argv[3]);
        if (world > 1) {
            const int ndev = gamut_hip_device_count();
            if (ndev < 1) return 9;
            if (rank == 0) {
                if (gamut_hip_init(0) != GAMUT_HIP_OK || gamut_hip_comm_get_unique_id(id) != GAMUT_HIP_OK) { fprintf(stderr, "id: %s\n", gamut_hip_last_error()); return 3; }
                char tmp[4096];
                snprintf(tmp, sizeof(tmp), "%s.tmp", argv[4]);
                FILE* f = fopen(tmp, "wb");
                if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id) || fclose(f) || rename(tmp, argv[4])) return 3;
            } else {
                FILE* f = NULL;
                for (int tries = 0; tries < 2400 && !(f = fopen(argv[4], "rb")); ++tries) usleep(50000);
                if (!f || fread(id, 1, sizeof(id), f) != sizeof(id)) return 3;
                fclose(f);
            }
        }
        const int rc = run_rank(world, rank, world > 1 ? id : NULL);
        if (!rc) printf("rank %d of %d ok\n", rank, world);
        return rc;
    }
    if (!strcmp(argv[1], "threads")) {
        if (argc > 3) TOTAL = atoi(argv[3]);
        if (argc > 4) PAD = atoi(argv[4]);
        if (TOTAL < 0 || TOTAL > 100000 || PAD < 0 || PAD > 4096) return 2;
        if (world > 1 && (gamut_hip_init(0) != GAMUT_HIP_OK || gamut_hip_comm_get_unique_id(id) != GAMUT_HIP_OK)) { fprintf(stderr, "id: %s\n", gamut_hip_last_error()); return 3; }
        pthread_t th[64];
        struct thread_arg args[64];
        for (int r = 0; r < world; ++r) {
            args[r].world = world; args[r].rank = r; args[r].rc = -1; args[r].id = world > 1 ? id : NULL;
            if (pthread_create(&th[r], NULL, thread_main, &args[r])) return 4;
        }
        int bad = 0;
        for (int r = 0; r < world; ++r) { pthread_join(th[r], NULL); bad |= args[r].rc; }
        if (!bad) printf("%d threads ok (%d images)\n", world, TOTAL);
        return bad;
    }
    return 2;
}
