/* The callback-shaped drop-ins driven the way plugins/png.d and plugins/jpeg.d would drive them, from C: the structs below have the
 * reference's exact layouts -- IOStream (io.d:86-103: five fread / fwrite / fseek / ftell / feof-compatible extern(C) procs),
 * IOAndHandle (stbdec.d:136-140), JPEGIOHandle (plugins/jpeg.d:155-163), stbi_io_callbacks (stbdec.d:408-419) -- and the four callbacks are the
 * bodies of stb_read / stb_skip / stb_eof (stbdec.d:143-165) and stream_read_jpeg (plugins/jpeg.d:167-177) with C linkage, i.e. what
 * bindings/gamut_hip.d's trampolines forward to.  The IOStream here is stdio itself, as the reference's file streams are.
 *
 *   callback_consumer <file> <req_comp>          file = one image, or several back to back (PNG / JPEG in any order)
 *
 * Per image one line: kind, width, height, comp, bytes, FNV-1a of the pixels, stream position after the call, number of read / skip / eof calls.
 * Without a GPU the loads return NULL (GAMUT_HIP_ERR_NO_DEVICE) AFTER the stream was walked: positions and call counts are still printed. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "gamut_hip.h"

typedef void* IOHandle;
typedef struct IOStream {
    size_t (*read)(void* buffer, size_t size, size_t count, IOHandle handle);
    size_t (*write)(const void* buffer, size_t size, size_t count, IOHandle handle);
    int    (*seek)(IOHandle handle, long offset, int origin);
    long   (*tell)(IOHandle handle);
    int    (*eof)(IOHandle handle);
} IOStream;
typedef struct IOAndHandle { IOStream* io; IOHandle handle; } IOAndHandle;
typedef struct JPEGIOHandle { IOStream* wrapped; IOHandle handle; unsigned char errored; } JPEGIOHandle;

static size_t io_read(void* b, size_t s, size_t c, IOHandle h) { return fread(b, s, c, (FILE*)h); }
static size_t io_write(const void* b, size_t s, size_t c, IOHandle h) { return fwrite(b, s, c, (FILE*)h); }
static int    io_seek(IOHandle h, long o, int w) { return fseek((FILE*)h, o, w); }
static long   io_tell(IOHandle h) { return ftell((FILE*)h); }
static int    io_eof(IOHandle h) { return feof((FILE*)h); }

static long n_read, n_skip, n_eof;

static int stb_read(void* user, char* data, int size)                         /* stbdec.d:143-152 */
{
    IOAndHandle* ioh = (IOAndHandle*)user;
    ++n_read;
    return (int)ioh->io->read(data, 1, (size_t)size, ioh->handle);
}
static void stb_skip(void* user, int n)                                        /* stbdec.d:155-159: io.skipBytes = seek(handle, n, SEEK_CUR) */
{
    IOAndHandle* ioh = (IOAndHandle*)user;
    ++n_skip;
    ioh->io->seek(ioh->handle, n, SEEK_CUR);
}
static int stb_eof(void* user)                                                 /* stbdec.d:162-166 */
{
    IOAndHandle* ioh = (IOAndHandle*)user;
    ++n_eof;
    return ioh->io->eof(ioh->handle);
}
static int stream_read_jpeg(void* pBuf, int max_bytes_to_read, unsigned char* pEOF_flag, void* userData)   /* plugins/jpeg.d:167-177 */
{
    JPEGIOHandle* jio = (JPEGIOHandle*)userData;
    ++n_read;
    size_t got = jio->wrapped->read(pBuf, 1, (size_t)max_bytes_to_read, jio->handle);
    if (pEOF_flag) *pEOF_flag = jio->wrapped->eof(jio->handle) != 0;
    return (int)got;
}

static uint64_t fnv(const unsigned char* p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    const int req = atoi(argv[2]);
    IOStream io = { io_read, io_write, io_seek, io_tell, io_eof };
    fseek(f, 0, SEEK_END);
    const long total = ftell(f);
    fseek(f, 0, SEEK_SET);
    const int have_gpu = gamut_hip_device_count() > 0;
    int images = 0;
    while (ftell(f) < total && images < 16) {
        unsigned char sig[2] = { 0, 0 };
        const long at = ftell(f);
        if (fread(sig, 1, 2, f) != 2) break;
        fseek(f, at, SEEK_SET);
        clearerr(f);
        n_read = n_skip = n_eof = 0;
        if (sig[0] == 0x89 && sig[1] == 'P') {                                 /* loadPNG, plugins/png.d:45-89 */
            IOAndHandle ioh = { &io, f };
            gamut_hip_stbi_io_callbacks cb = { stb_read, stb_skip, stb_eof };
            const int is16 = gamut_hip_stbi_png_is16_from_callbacks(&cb, &ioh);
            if (io.seek(f, at, SEEK_SET) != 0) return 4;                       /* "rewind stream" -- to the image's start here */
            int w = 0, h = 0, comp = 0;
            float ppmx = -1, ppmy = -1, ratio = -1;
            void* px = is16 ? (void*)gamut_hip_stbi_load_16_from_callbacks(&cb, &ioh, &w, &h, &comp, req, &ppmx, &ppmy, &ratio)
                            : (void*)gamut_hip_stbi_load_from_callbacks(&cb, &ioh, &w, &h, &comp, req, &ppmx, &ppmy, &ratio);
            if (!px && have_gpu) { printf("png: load failed: %s\n", gamut_hip_last_error()); return 5; }
            const size_t bytes = px ? (size_t)w * h * (req ? req : comp) * (is16 ? 2 : 1) : 0;
            printf("png %d %d %d %d %zu %016llx %ld %ld %ld %ld\n", w, h, comp, is16, bytes, (unsigned long long)(px ? fnv(px, bytes) : 0), ftell(f), n_read, n_skip, n_eof);
            free(px);
        } else if (sig[0] == 0xFF && sig[1] == 0xD8) {                         /* loadJPEG, plugins/jpeg.d:42-61 */
            JPEGIOHandle jio = { &io, f, 0 };
            int w = 0, h = 0, comps = 0;
            float par = -1, dpi = -1;
            unsigned char* px = gamut_hip_decompress_jpeg_image_from_stream(stream_read_jpeg, &jio, &w, &h, &comps, &par, &dpi, req ? req : -1);
            if (!px && have_gpu) { printf("jpeg: load failed: %s\n", gamut_hip_last_error()); return 6; }
            const size_t bytes = px ? (size_t)w * h * (req ? req : comps) : 0;
            printf("jpeg %d %d %d 0 %zu %016llx %ld %ld %ld %ld\n", w, h, comps, bytes, (unsigned long long)(px ? fnv(px, bytes) : 0), ftell(f), n_read, n_skip, n_eof);
            free(px);
            /* jpgd over-reads by less than one 8 KiB piece: the next image is found by the caller's own framing; here: scan for it */
            long p = at + 2;
            fseek(f, p, SEEK_SET);
            int c, prev = 0;
            long next = total;
            while ((c = fgetc(f)) != EOF) { if (prev == 0xFF && c == 0xD9) { next = ftell(f); break; } prev = c; }
            fseek(f, next, SEEK_SET);
        } else break;
        ++images;
    }
    fclose(f);
    return images ? 0 : 7;
}
