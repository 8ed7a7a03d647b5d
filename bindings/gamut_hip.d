/**
 * gamut_hip.d -- D binding of include/gamut_hip.h (libgamut_hip.so), the file a Gamut maintainer adds as
 * `source/gamut/hip.d`.  The C header is the source of truth; tests/test_capi_cpu.py::test_d_binding_lists_every_export
 * checks that every function the header declares is declared here with the same return and parameter TYPES (a C -> D type map),
 * ::test_d_binding_struct_layouts that the structs declared here have the header's layout (size and every field offset).
 * (No D compiler exists in the image this was written in: the file is checked textually, not compiled; the `static assert`s
 * at the end of the extern(C) block are what a D compiler will check on the first build.)
 *
 * Linkage.  Everything inside the `extern(C)` block has C linkage -- including the function-pointer TYPES it declares.
 * The reference's own callbacks do NOT: `stream_read_jpeg` (plugins/jpeg.d:167), `stb_read`, `stb_skip`, `stb_eof`
 * (codecs/stbdec.d:143-165) and the aliases / struct members that hold them (`JpegStreamReadFunc` jpegload.d:70,
 * `stbi_io_callbacks` stbdec.d:408-419) are extern(D).  `&stream_read_jpeg` therefore does not convert to
 * `gamut_hip_jpeg_stream_read_func`, and a cast that makes it compile would call a D-ABI function with the C convention
 * (DMD and LDC pass extern(D) parameters in reverse order on x86-64): `stb_read(user, data, size)` would receive
 * `(size, data, user)`.  Two correct ways to make the swap, both shown at the end of this file:
 *   (1) mark the four reference functions `extern(C)` (a one-word change each; they keep working for stb / jpgd, whose
 *       callback types then have to be extern(C) as well), or
 *   (2) leave the reference untouched and pass the four one-line extern(C) trampolines below.
 */
module gamut.hip;

nothrow @nogc:

extern(C)
{
    enum { GAMUT_HIP_OK = 0, GAMUT_HIP_ERR_INVALID_ARG, GAMUT_HIP_ERR_UNSUPPORTED, GAMUT_HIP_ERR_OUT_OF_MEMORY,
           GAMUT_HIP_ERR_HIP, GAMUT_HIP_ERR_DECODE, GAMUT_HIP_ERR_NO_DEVICE }
    enum { GAMUT_JPGD_GRAYSCALE = 0, GAMUT_JPGD_YH1V1, GAMUT_JPGD_YH2V1, GAMUT_JPGD_YH1V2, GAMUT_JPGD_YH2V2 }
    enum { GAMUT_HIP_INFLATE_E_BLOCK_TYPE = 1, GAMUT_HIP_INFLATE_E_STORED, GAMUT_HIP_INFLATE_E_LENGTHS, GAMUT_HIP_INFLATE_E_CODE,
           GAMUT_HIP_INFLATE_E_DISTANCE, GAMUT_HIP_INFLATE_E_INPUT }
    enum { GAMUT_HIP_FORMAT_UNKNOWN = -1, GAMUT_HIP_FORMAT_JPEG = 0, GAMUT_HIP_FORMAT_PNG = 1, GAMUT_HIP_FORMAT_QOI = 2 }
    enum GAMUT_HIP_QOI_SLACK = 160;
    enum GAMUT_HIP_COMM_ID_BYTES = 128;

    // ---- runtime ----------------------------------------------------------------------------------------------
    const(char)* gamut_hip_version();
    int   gamut_hip_device_count();
    int   gamut_hip_init(int device);
    void  gamut_hip_shutdown();
    const(char)* gamut_hip_last_error();
    void* gamut_hip_device_malloc(size_t bytes);
    void  gamut_hip_device_free(void* p);
    void* gamut_hip_host_malloc_pinned(size_t bytes);
    void  gamut_hip_host_free_pinned(void* p);
    int   gamut_hip_memcpy_h2d(void* dst, const(void)* src, size_t bytes, void* stream);
    int   gamut_hip_memcpy_d2h(void* dst, const(void)* src, size_t bytes, void* stream);
    void* gamut_hip_stream_create();
    void  gamut_hip_stream_destroy(void* stream);
    int   gamut_hip_stream_synchronize(void* stream);

    // ---- scanline.d: PixelType is passed as int (enum PixelType has base type int, same ordinals, types.d:32-59) ------
    int gamut_hip_pixel_type_size(int type);
    int gamut_hip_scanlines_inter_type(int srcType, int dstType);
    int gamut_hip_scanlines_convert(int srcType, const(ubyte)* src, int srcPitch,
                                    int dstType, ubyte* dst, int dstPitch, int width, int height);
    int gamut_hip_scanlines_copy(int type, const(ubyte)* src, int srcPitch, ubyte* dst, int dstPitch, int width, int height);
    int gamut_hip_scanlines_convert_device(int srcType, const(void)* src, long srcPitch, long srcLayerOffset,
                                           int dstType, void* dst, long dstPitch, long dstLayerOffset,
                                           int width, int height, int layers, void* stream);
    int gamut_hip_flip_device(int type, void* data, long pitch, long layerOffset, int width, int height, int layers, int vertical, void* stream);
    int gamut_hip_flip(int type, ubyte* data, int pitch, int width, int height, int vertical);

    // ---- jpegload.d ---------------------------------------------------------------------------------------------
    struct gamut_hip_jpeg_desc  { const(short)* coeffs; const(ubyte)* max_zag; ubyte* out_; long out_pitch;
                                  int width, height, scan_type, out_comps; }
    struct gamut_hip_jpeg_frame { int width, height, comps, scan_type, mcus_per_row, mcus_per_col, blocks_per_mcu;
                                  short* coeffs; ubyte* max_zag; float pixel_aspect_ratio, dpi_y; }
    int  gamut_hip_jpeg_reconstruct_device(const(gamut_hip_jpeg_desc)* descs, int count, void* stream);
    int  gamut_hip_jpeg_reconstruct_batch_device(const(short)* coeffs, long coeff_stride, const(ubyte)* max_zag, long zag_stride,
                                                 ubyte* out_, long out_pitch, long out_stride,
                                                 int width, int height, int scan_type, int out_comps, int count, void* stream);
    int  gamut_hip_jpeg_decode_coeffs(const(ubyte)* data, size_t len, gamut_hip_jpeg_frame* out_);
    void gamut_hip_jpeg_frame_free(gamut_hip_jpeg_frame* f);
    int  gamut_hip_jpeg_read_header(const(ubyte)* data, size_t len, gamut_hip_jpeg_frame* out_);
    int  gamut_hip_jpeg_scan_layout(const(ubyte)* data, size_t len, gamut_hip_jpeg_frame* info, int* segments, ulong* entropy_bytes);
    int  gamut_hip_jpeg_entropy_decode_device(const(ubyte*)* data, const(size_t)* len, int count,
                                              const(long)* coeff_offset, const(long)* zag_offset,
                                              short* coeffs, ubyte* max_zag, uint* status_dev,
                                              gamut_hip_jpeg_frame* info, int* status_host, void* stream);
    int  gamut_hip_jpeg_decode_batch_device(const(ubyte*)* data, const(size_t)* len, int count, int req_comps,
                                            const(long)* out_offset, ubyte* out_, gamut_hip_jpeg_frame* info, int* status_host,
                                            uint* status_dev, void* stream);
    int  gamut_hip_jpeg_decode_coeffs_batch(const(ubyte*)* data, const(size_t)* len, int count,
                                            gamut_hip_jpeg_frame* out_, int* status, int threads);
    ubyte* gamut_hip_decompress_jpeg_image_from_memory(const(ubyte)* data, size_t len, int* width, int* height, int* actual_comps,
                                                       float* pixelAspectRatio, float* dotsPerInchY, int req_comps);
    // C linkage (this alias sits inside extern(C)); D's bool is one byte = the header's unsigned char
    alias gamut_hip_jpeg_stream_read_func = int function(void* pBuf, int max_bytes_to_read, bool* pEOF_flag, void* userData);
    ubyte* gamut_hip_decompress_jpeg_image_from_stream(gamut_hip_jpeg_stream_read_func rfn, void* userData, int* width, int* height,
                                                       int* actual_comps, float* pixelAspectRatio, float* dotsPerInchY, int req_comps);

    // ---- stbdec.d ------------------------------------------------------------------------------------------------
    struct gamut_hip_png_desc { const(ubyte)* raw; ubyte* out_; uint raw_len, x, y; int img_n, out_n, depth, color; }
    int gamut_hip_png_defilter_device(const(gamut_hip_png_desc)* descs, int count, uint* status, void* stream);
    int gamut_hip_png_defilter_batch_device(const(ubyte)* raw, long raw_stride, uint raw_len, ubyte* out_, long out_stride,
                                            uint x, uint y, int img_n, int out_n, int depth, int color, int count, uint* status, void* stream);
    ubyte*  gamut_hip_stbi_load_from_memory(const(ubyte)* data, size_t len, int* x, int* y, int* comp, int req_comp,
                                            float* ppmX, float* ppmY, float* pixelRatio);
    ushort* gamut_hip_stbi_load_16_from_memory(const(ubyte)* data, size_t len, int* x, int* y, int* comp, int req_comp,
                                               float* ppmX, float* ppmY, float* pixelRatio);
    int gamut_hip_png_is16(const(ubyte)* data, size_t len);
    // same layout as stbi_io_callbacks (stbdec.d:408-419) -- three pointers -- but the members are C-linkage function pointers
    struct gamut_hip_stbi_io_callbacks
    {
        int  function(void* user, char* data, int size) read;
        void function(void* user, int n) skip;
        int  function(void* user) eof;
    }
    ubyte*  gamut_hip_stbi_load_from_callbacks(const(gamut_hip_stbi_io_callbacks)* clbk, void* user, int* x, int* y, int* comp, int req_comp,
                                               float* ppmX, float* ppmY, float* pixelRatio);
    ushort* gamut_hip_stbi_load_16_from_callbacks(const(gamut_hip_stbi_io_callbacks)* clbk, void* user, int* x, int* y, int* comp, int req_comp,
                                                  float* ppmX, float* ppmY, float* pixelRatio);
    int gamut_hip_stbi_png_is16_from_callbacks(const(gamut_hip_stbi_io_callbacks)* clbk, void* user);   // header only; rewind afterwards (png.d:50-62)

    struct gamut_hip_inflate_desc { const(ubyte)* src; ubyte* dst; uint src_len, dst_cap; }
    int gamut_hip_inflate_batch_device(const(gamut_hip_inflate_desc)* descs, int count, uint* out_len_dev, uint* status_dev, void* stream);
    int gamut_hip_inflate_batch_device_sliced(const(gamut_hip_inflate_desc)* descs, int count, uint* out_len_dev, uint* status_dev,
                                              uint slice_bytes, void* stream);
    struct gamut_hip_png_info { uint width, height; int channels_in_file, channels, bits;
                                float pixels_per_meter_x, pixels_per_meter_y, pixel_aspect_ratio; }
    int gamut_hip_png_read_header(const(ubyte)* data, size_t len, gamut_hip_png_info* info);
    int gamut_hip_png_decode_batch_device(const(ubyte*)* data, const(size_t)* len, int count, int req_comp, int bits,
                                          const(long)* out_offset, ubyte* out_, gamut_hip_png_info* info, int* status_host,
                                          int threads, void* stream);

    // ---- qoi.d ---------------------------------------------------------------------------------------------------
    struct gamut_hip_qoi_desc { uint width, height; ubyte channels, colorspace; }
    void* gamut_hip_qoi_decode(const(void)* data, int size, gamut_hip_qoi_desc* desc, int channels);
    int   gamut_hip_qoi_read_header(const(void)* data, int size, gamut_hip_qoi_desc* desc);
    int   gamut_hip_qoi_decode_batch_device(const(ubyte*)* data, const(int)* size, int count, int channels,
                                            const(long)* out_offset, ubyte* out_, gamut_hip_qoi_desc* descs, int* status_host, void* stream);
    int   gamut_hip_qoi_decode_resident_device(const(ubyte)* blob, long blob_len, const(long)* begin, const(int)* size,
                                               const(gamut_hip_qoi_desc)* descs, int count, int channels,
                                               const(long)* out_offset, ubyte* out_, void* stream);

    // ---- any of the three formats, one call (image.d:1045-1061 identifyFormatFromStream + g_plugins[fif].loadProc, batched) ----
    struct gamut_hip_image_info { int format, width, height, channels_in_file, channels; }
    int gamut_hip_identify_format(const(ubyte)* data, size_t len);
    int gamut_hip_decode_batch_device(const(ubyte*)* data, const(size_t)* len, int count, int req_comps,
                                      const(long)* out_offset, ubyte* out_, gamut_hip_image_info* info, int* status_host, void* stream);

    // ---- multi-GPU: image-index round-robin and the gather of decoded outputs over RCCL ----------------------------
    int  gamut_hip_shard_owner(long image_index, int world);
    long gamut_hip_shard_count(int rank, int world, long total_images);
    long gamut_hip_shard_local_index(long image_index, int world);
    long gamut_hip_shard_global_index(long local_index, int rank, int world);
    int  gamut_hip_host_threads();
    struct gamut_hip_comm;
    int  gamut_hip_comm_get_unique_id(void* id128);
    int  gamut_hip_comm_init(gamut_hip_comm** comm, int world, int rank, const(void)* id128);
    void gamut_hip_comm_destroy(gamut_hip_comm* comm);
    int  gamut_hip_comm_rank(const(gamut_hip_comm)* comm);
    int  gamut_hip_comm_world(const(gamut_hip_comm)* comm);
    int  gamut_hip_gather_outputs_device(gamut_hip_comm* comm, const(void)* local, long local_stride, long bytes_per_image,
                                         long total_images, void* dst, long dst_stride, int root, void* stream);

    // ---- struct layouts: the numbers gcc prints for include/gamut_hip.h (tests/c/abi_layout.c); a D compiler checks them the first time
    //      this file is built, tests/test_capi_cpu.py::test_d_binding_struct_layouts checks them against the C compiler every run -------------
    static assert(gamut_hip_jpeg_desc.sizeof == 48 && gamut_hip_jpeg_desc.coeffs.offsetof == 0 && gamut_hip_jpeg_desc.max_zag.offsetof == 8 && gamut_hip_jpeg_desc.out_.offsetof == 16 && gamut_hip_jpeg_desc.out_pitch.offsetof == 24 && gamut_hip_jpeg_desc.width.offsetof == 32 && gamut_hip_jpeg_desc.height.offsetof == 36 && gamut_hip_jpeg_desc.scan_type.offsetof == 40 && gamut_hip_jpeg_desc.out_comps.offsetof == 44);
    static assert(gamut_hip_jpeg_frame.sizeof == 56 && gamut_hip_jpeg_frame.width.offsetof == 0 && gamut_hip_jpeg_frame.height.offsetof == 4 && gamut_hip_jpeg_frame.comps.offsetof == 8 && gamut_hip_jpeg_frame.scan_type.offsetof == 12 && gamut_hip_jpeg_frame.mcus_per_row.offsetof == 16 && gamut_hip_jpeg_frame.mcus_per_col.offsetof == 20 && gamut_hip_jpeg_frame.blocks_per_mcu.offsetof == 24 && gamut_hip_jpeg_frame.coeffs.offsetof == 32 && gamut_hip_jpeg_frame.max_zag.offsetof == 40 && gamut_hip_jpeg_frame.pixel_aspect_ratio.offsetof == 48 && gamut_hip_jpeg_frame.dpi_y.offsetof == 52);
    static assert(gamut_hip_png_desc.sizeof == 48 && gamut_hip_png_desc.raw.offsetof == 0 && gamut_hip_png_desc.out_.offsetof == 8 && gamut_hip_png_desc.raw_len.offsetof == 16 && gamut_hip_png_desc.x.offsetof == 20 && gamut_hip_png_desc.y.offsetof == 24 && gamut_hip_png_desc.img_n.offsetof == 28 && gamut_hip_png_desc.out_n.offsetof == 32 && gamut_hip_png_desc.depth.offsetof == 36 && gamut_hip_png_desc.color.offsetof == 40);
    static assert(gamut_hip_stbi_io_callbacks.sizeof == 24 && gamut_hip_stbi_io_callbacks.read.offsetof == 0 && gamut_hip_stbi_io_callbacks.skip.offsetof == 8 && gamut_hip_stbi_io_callbacks.eof.offsetof == 16);
    static assert(gamut_hip_inflate_desc.sizeof == 24 && gamut_hip_inflate_desc.src.offsetof == 0 && gamut_hip_inflate_desc.dst.offsetof == 8 && gamut_hip_inflate_desc.src_len.offsetof == 16 && gamut_hip_inflate_desc.dst_cap.offsetof == 20);
    static assert(gamut_hip_png_info.sizeof == 32 && gamut_hip_png_info.width.offsetof == 0 && gamut_hip_png_info.height.offsetof == 4 && gamut_hip_png_info.channels_in_file.offsetof == 8 && gamut_hip_png_info.channels.offsetof == 12 && gamut_hip_png_info.bits.offsetof == 16 && gamut_hip_png_info.pixels_per_meter_x.offsetof == 20 && gamut_hip_png_info.pixels_per_meter_y.offsetof == 24 && gamut_hip_png_info.pixel_aspect_ratio.offsetof == 28);
    static assert(gamut_hip_qoi_desc.sizeof == 12 && gamut_hip_qoi_desc.width.offsetof == 0 && gamut_hip_qoi_desc.height.offsetof == 4 && gamut_hip_qoi_desc.channels.offsetof == 8 && gamut_hip_qoi_desc.colorspace.offsetof == 9);
    static assert(gamut_hip_image_info.sizeof == 20 && gamut_hip_image_info.format.offsetof == 0 && gamut_hip_image_info.width.offsetof == 4 && gamut_hip_image_info.height.offsetof == 8 && gamut_hip_image_info.channels_in_file.offsetof == 12 && gamut_hip_image_info.channels.offsetof == 16);
}

// ================================================================================================================
// The callbacks.  Way (2): four trampolines with C linkage that forward to the reference's extern(D) functions.
// They live next to their targets: the JPEG one in plugins/jpeg.d, the three stb ones in codecs/stbdec.d.
// ================================================================================================================

version (GamutHipTrampolines)
{
    import gamut.plugins.jpeg : stream_read_jpeg;          // plugins/jpeg.d:167, extern(D)
    import gamut.codecs.stbdec : stb_read, stb_skip, stb_eof, IOAndHandle;   // codecs/stbdec.d:143-165, extern(D)

    extern(C) int gamut_hip_tramp_read_jpeg(void* pBuf, int max_bytes_to_read, bool* pEOF_flag, void* userData) @system
    {
        return stream_read_jpeg(pBuf, max_bytes_to_read, pEOF_flag, userData);
    }
    extern(C) int  gamut_hip_tramp_stb_read(void* user, char* data, int size) @system { return stb_read(user, data, size); }
    extern(C) void gamut_hip_tramp_stb_skip(void* user, int n) @system               { stb_skip(user, n); }
    extern(C) int  gamut_hip_tramp_stb_eof(void* user) @system                        { return stb_eof(user); }

    /// what `initSTBCallbacks` (stbdec.d:126-133) becomes for the HIP path
    void initHipSTBCallbacks(IOStream* io, IOHandle handle, IOAndHandle* ioh, gamut_hip_stbi_io_callbacks* cb) @system
    {
        ioh.io = io;
        ioh.handle = handle;
        cb.read = &gamut_hip_tramp_stb_read;
        cb.skip = &gamut_hip_tramp_stb_skip;
        cb.eof  = &gamut_hip_tramp_stb_eof;
    }

    // loadJPEG, plugins/jpeg.d:61 -- the call becomes
    //     ubyte* p = gamut_hip_decompress_jpeg_image_from_stream(&gamut_hip_tramp_read_jpeg, &jio, &width, &height, &actualComp,
    //                                                            &pixelAspectRatio, &dotsPerInchY, requestedComp);
    //     ubyte[] decoded = p is null ? null : p[0 .. cast(size_t) width * height * (requestedComp == -1 ? actualComp : requestedComp)];
    // loadPNG, plugins/png.d:45-89 -- `initSTBCallbacks(io, handle, &ioh, &stb_callback)` becomes
    //     gamut_hip_stbi_io_callbacks cb;  initHipSTBCallbacks(io, handle, &ioh, &cb);
    //     bool is16bit = gamut_hip_stbi_png_is16_from_callbacks(&cb, &ioh) != 0;
    //     ... decoded = gamut_hip_stbi_load_from_callbacks(&cb, &ioh, &width, &height, &components, requestedComp, &ppmX, &ppmY, &pixelRatio);
}
