/*
 * s2sr.h -- C ABI of libs2sr.so, the MI355X (gfx950) Real-ESRGAN x4 inference path.
 *
 * The reference (fieldin/sentinel2-super-resolution-poc) has no FFI layer: its seam is the
 * Python class `RealESRGAN` (server/app/cnn_super_resolution.py:161-280) plus the free
 * functions `_enhance_for_crops` (server/app/wow_sr.py:187-209) and the farm variants
 * (server/app/farm_sr.py:61-108).  This header is the native boundary a replacement of that
 * seam binds; the ctypes stub that sits on it is in INTEGRATION.md and
 * in sentinel2-super-resolution-poc_amd/s2sr/native.py.
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative
 * S2SR_E_* code; no exception crosses the boundary; outputs are caller-allocated; the last
 * error text is handle-scoped (s2sr_last_error).  A handle serialises its own calls
 * (internal mutex), so it may be shared by the reference's worker threads
 * (server/app/main.py:247-368 run jobs from a thread pool).
 *
 * Pointers named `d_*` are DEVICE pointers (HIP), everything else is host memory.
 */
#ifndef S2SR_H
#define S2SR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S2SR_OK            0
#define S2SR_E_INVALID    -1   /* bad argument */
#define S2SR_E_HIP        -2   /* HIP runtime error (text in s2sr_last_error) */
#define S2SR_E_NOWEIGHTS  -3   /* forward called before s2sr_load_weights */
#define S2SR_E_BADBLOB    -4   /* weight blob size does not match the configured net */
#define S2SR_E_NODEVICE   -5   /* no gfx950 device visible: there is NO CPU fallback */
#define S2SR_E_CAPACITY   -6   /* caller buffer too small */
#define S2SR_E_IO         -7   /* a file could not be written (errno holds the reason) */

/* arithmetic of the conv stack */
#define S2SR_PREC_F16  0   /* fp16 operands, fp32 accumulate on MFMA; fp32 residual trunk  */
#define S2SR_PREC_F16_HP 1 /* same, plus the six convs outside the RRDB trunk (conv_first, conv_body, up1, up2,
                            * hr, last) computed with split fp16 operands (x_hi,w_hi)+(x_lo,w_hi)+(x_hi,w_lo):
                            * fp32-class head/tail, ~1e-4 of the fp32 reference at ~1.2x the time            */

#define S2SR_PREC_FP8 2    /* the 345 RDB convs on e4m3 operands (block-scaled fp8 MFMA, K = 64; per-output-channel weight
                            * scales, per-tensor-kind activation scales, fp16 trunk); the six head / tail convs in PLAIN fp16 (as S2SR_PREC_F16;
                            * their ~2e-3 is below the trunk's e4m3 error) unless S2SR_FP8_TAIL=hp selects the split-operand forms.
                            * BASELINE.json configs[4] (the /api/sr variant).  NOT within the 1e-3 tolerance: e4m3 keeps
                            * 3 mantissa bits; measured max-abs in tests/test_gpu_net.py (test_fp8_mode_*)             */

typedef struct s2sr_handle s2sr_handle;

/* Mirrors the constructor arguments of the reference net
 * `RRDBNet(num_in_ch=3,num_out_ch=3,num_feat,num_block,num_grow_ch=32,scale)`
 * (cnn_super_resolution.py:113-121,196-203) plus device-side knobs. */
typedef struct s2sr_config {
    int32_t num_block;   /* 23 (realesrgan_x4) or 6 (realesrgan_anime), cnn_super_resolution.py:28-45 */
    int32_t num_feat;    /* must be 64 */
    int32_t num_grow;    /* must be 32 */
    int32_t scale;       /* must be 4 (the only scale in the reference's MODELS table) */
    int32_t precision;   /* S2SR_PREC_* */
    int32_t device;      /* HIP device ordinal */
    int32_t group;       /* images pushed through the trunk together (0 = default) */
    int32_t reserved;
} s2sr_config;

/* One window of RealESRGAN._tile_process (cnn_super_resolution.py:244-278). */
typedef struct s2sr_window {
    int32_t y1, y2, x1, x2;             /* input rectangle, LR pixels                          */
    int32_t crop_top, crop_bottom, crop_left, crop_right;   /* output pixels dropped           */
    int32_t oy1, oy2, ox1, ox2;         /* paste rectangle in the output image                 */
} s2sr_window;

/* Constants of the crop-visibility post-process (wow_sr.py:187-209 / farm_sr.py:61-108,170-178). */
typedef struct s2sr_pp_params {
    float   clahe_clip;      /* cv2.createCLAHE clipLimit: 2.5                     */
    int32_t clahe_grid;      /* tileGridSize (g,g): 8                              */
    float   blur_sigma;      /* GaussianBlur sigma: 1.2 (wow) / 1.5 (farm)         */
    float   w_img;           /* addWeighted alpha: 1.4 (wow) / 2.2 (farm)          */
    float   w_blur;          /* addWeighted beta: -0.4 (wow) / -1.2 (farm)         */
    int32_t hue_lo, hue_hi;  /* exclusive hue bounds of the green mask: 35, 85     */
    float   sat_gain;        /* 1.2 (wow) / 1.3 (farm)                             */
    int32_t stages;          /* bit0 CLAHE, bit1 unsharp, bit2 vegetation; 7 = all */
} s2sr_pp_params;

/* per-kernel-family timing collected with HIP events on the handle's stream */
typedef struct s2sr_kstat {
    char     name[48];
    int64_t  launches;
    double   total_ms;
    double   flops;          /* algorithmic FLOP summed over those launches  */
    double   bytes;          /* algorithmic HBM bytes summed over those launches */
} s2sr_kstat;

const char* s2sr_version(void);
int  s2sr_device_count(void);      /* number of HIP devices (0 when there is no GPU) */

/* replaces RealESRGAN.__init__'s model construction (cnn_super_resolution.py:196-203) */
int  s2sr_create(const s2sr_config* cfg, s2sr_handle** out);
void s2sr_destroy(s2sr_handle* h);
const char* s2sr_last_error(const s2sr_handle* h);   /* h may be NULL: last create() error */

/* Page-locked host memory for images that cross PCIe.  The reference returns `output.cpu().numpy()` (cnn_super_resolution.py:231-233),
 * a pageable array; a destination from s2sr_host_alloc lets s2sr_enhance_u8 / s2sr_forward_batch_u8 land their bands with the DMA
 * engines (no staging copy, no first-touch page faults: 805 MB in 20 ms instead of 130).  Any host pointer stays valid as a
 * destination; the library detects page-locked ones (hipPointerGetAttributes).  Not tied to a handle or device. */
int  s2sr_host_alloc(size_t bytes, void** out);
int  s2sr_host_free(void* p);

/* replaces load_state_dict (cnn_super_resolution.py:205-213).  `blob`: for every conv in
 * registration order (conv_first, body.{b}.rdb{1..3}.conv{1..5}, conv_body, conv_up1,
 * conv_up2, conv_hr, conv_last): weight[Cout][Cin][3][3] then bias[Cout], fp32. */
int  s2sr_load_weights(s2sr_handle* h, const float* blob, size_t n_floats);
size_t s2sr_expected_blob_floats(int32_t num_block);
/* same blob, DEVICE-resident (e.g. the receive buffer of the RCCL weight broadcast); `stream`
 * is the stream the blob was produced on (a hipStream_t; NULL = default stream).  Returns when loaded.  The RDB convs are
 * repacked on the device; only the six head/tail convs' weights (0.9 MB) pass through host memory. */
int  s2sr_load_weights_dev(s2sr_handle* h, const void* d_blob, size_t n_floats, void* stream);

/* S2SR_PREC_FP8 only: choose the two activation scales of the fp8 trunk from data.  Runs one forward of `tiles`
 * ([B,th,tw,3] u8, host) with wide scales, takes the largest |x| of the trunk and |x_k| of the growth features over all
 * RDBs, and sets the exponents so that headroom x those maxima stays below e4m3's 448 (headroom >= 1; 2 is a sane
 * default: e4m3 is a floating format, so a wider scale costs precision only at its subnormal end).  The defaults (3 / 5) come from the synthetic calibration set, profiles/r02_fp8_scale_sweep.txt; a deployment
 * with real checkpoints calls this once after s2sr_load_weights with a few representative tiles. */
int  s2sr_calibrate_fp8(s2sr_handle* h, const uint8_t* tiles, int32_t B, int32_t th, int32_t tw, float headroom,
                        int32_t* x_exp, int32_t* g_exp);

/* pure host function = the index math of _tile_process (cnn_super_resolution.py:244-278) */
int  s2sr_plan_tiles(int32_t H, int32_t W, int32_t tile, int32_t pad, int32_t scale,
                     s2sr_window* out, int32_t cap, int32_t* n);

/* replaces RRDBNet.forward + the u8 quantisation of enhance() on a batch of equal-size tiles
 * (cnn_super_resolution.py:140-158,220-222,231-232): [B,h,w,3] u8 -> [B,4h,4w,3] u8. */
int  s2sr_forward_batch_u8(s2sr_handle* h, const uint8_t* tiles, int32_t B, int32_t th, int32_t tw,
                           uint8_t* out);
/* same with device-resident input/output, asynchronous on `stream` (a hipStream_t; NULL = the
 * default stream, ordered with the caller's other default-stream work) */
int  s2sr_forward_batch_u8_dev(s2sr_handle* h, const void* d_tiles, int32_t B, int32_t th, int32_t tw,
                               void* d_out, void* stream);
/* unquantised net output for parity tests: x [N,3,H,W] fp32 in [0,1] -> y [N,3,4H,4W] fp32 */
int  s2sr_forward_f32(s2sr_handle* h, const float* x, int32_t N, int32_t H, int32_t W, float* y);

/* replaces RealESRGAN.enhance incl. the whole/tiled switch and _tile_process
 * (cnn_super_resolution.py:217-280): HxWx3 u8 -> 4Hx4Wx3 u8, channel order as given. */
int  s2sr_enhance_u8(s2sr_handle* h, const uint8_t* img, int32_t H, int32_t W,
                     int32_t tile, int32_t pad, uint8_t* out);
/* The device work of one /api/wow or /api/sr job in ONE call (apply_wow_sr, wow_sr.py:85-110; apply_farm_sr, farm_sr.py:156-178):
 * RGB image in -> cvtColor RGB2BGR -> RealESRGAN.enhance -> BGR2RGB -> the crop-visibility post-process (prm; NULL: none) ->
 * RGB image out.  Same bytes as s2sr_enhance_u8 on the swapped image followed by s2sr_postprocess_u8; one upload and one
 * download instead of three round trips and two host-side channel flips of the 16x image. */
int  s2sr_enhance_job_u8(s2sr_handle* h, const uint8_t* rgb, int32_t H, int32_t W, int32_t tile, int32_t pad,
                         const s2sr_pp_params* prm, uint8_t* out_rgb);
/* float image before quantisation (HWC fp32), for parity tests of the tiled path */
int  s2sr_enhance_f32(s2sr_handle* h, const uint8_t* img, int32_t H, int32_t W,
                      int32_t tile, int32_t pad, float* out);

/* RealESRGAN._tile_process alone (cnn_super_resolution.py:236-280): always the window plan,
 * whatever the image size; HWC fp32 out (unquantised). */
int  s2sr_tile_process_f32(s2sr_handle* h, const uint8_t* img, int32_t H, int32_t W,
                           int32_t tile, int32_t pad, float* out);

/* Multi-GPU building blocks of _tile_process (cnn_super_resolution.py:244-278), device-resident:
 * cut windows [first, first+count) of the plan into d_tiles [count, wh, ww, 3] (wh/ww = the
 * plan's common window size), and paste ALL T windows' outputs d_tiles [T, 4wh, 4ww, 3] into
 * d_out [4H, 4W, 3] with the reference's crop + overwrite order.  Between the two a rank runs
 * s2sr_forward_batch_u8_dev on its share and the ranks all-gather (RCCL) the outputs. */
int  s2sr_cut_windows_u8_dev(s2sr_handle* h, const void* d_img, int32_t H, int32_t W, int32_t tile, int32_t pad,
                             int32_t first, int32_t count, void* d_tiles, void* stream);
int  s2sr_stitch_windows_u8_dev(s2sr_handle* h, const void* d_tiles, int32_t H, int32_t W, int32_t tile, int32_t pad,
                                void* d_out, void* stream);
/* The same paste for output rows [oy0, oy1) only (d_out is still the whole [4H, 4W, 3] image): a rank that receives the windows
 * chunk by chunk stitches every band as soon as the window rows that own it have arrived, and copies it out under the next
 * chunk's compute.  The plan's paste maps stay on the device between calls with the same (H, W, tile, pad). */
int  s2sr_stitch_rows_u8_dev(s2sr_handle* h, const void* d_tiles, int32_t H, int32_t W, int32_t tile, int32_t pad,
                             int32_t oy0, int32_t oy1, void* d_out, void* stream);
/* s2sr_forward_batch_u8_dev for a PART of a job of `job_windows` (>= B) equal windows: the window mosaic and the workspace are
 * planned for the whole job, so its parts (the chunks a rank's share of an AOI is cut into) share one workspace and their
 * hipGraphs.  Same bytes as any other split. */
int  s2sr_forward_part_u8_dev(s2sr_handle* h, const void* d_tiles, int32_t B, int32_t th, int32_t tw, int32_t job_windows,
                              void* d_out, void* stream);
/* Device -> host: `bytes` from d_src into dst once everything enqueued on `stream` so far has run; returns when they are there.
 * Replaces the reference's `output.cpu()` (cnn_super_resolution.py:231) for callers that hold device buffers: a destination
 * from s2sr_host_alloc takes one DMA, a pageable one goes through pinned staging slices. */
int  s2sr_copy_to_host(s2sr_handle* h, void* dst, const void* d_src, size_t bytes, void* stream);

/* replaces _enhance_for_crops (wow_sr.py:187-209) and enhance_local_contrast /
 * apply_unsharp_mask / enhance_vegetation (farm_sr.py:61-108): HxWx3 u8 RGB -> same. */
int  s2sr_postprocess_u8(s2sr_handle* h, const uint8_t* rgb, int32_t H, int32_t W,
                         const s2sr_pp_params* prm, uint8_t* out);
int  s2sr_postprocess_batch_u8_dev(s2sr_handle* h, const void* d_rgb, int32_t B, int32_t H, int32_t W,
                                   const s2sr_pp_params* prm, void* d_out, void* stream);

/* The same post-process over ONE device-resident image in row bands, for callers whose image becomes complete band by band (an
 * AOI's mosaic: the chunks of s2sr_enhance_u8, the gathers of s2sr/dist.py).  CLAHE's 8x8 grid spans the whole image
 * (wow_sr.py:191-192), so no output row exists before every input row has been counted; the split lets the counting run under
 * the compute of the windows still to come and the finishing overlap the copy out:
 *   begin  geometry, constants and channel order of the image; zeroes the histograms (allocates: call it before queueing work)
 *   hist   counts rows [y0, y1) of d_img ([H, W, 3] u8) -- any order, every row exactly once
 *   lut    clip / redistribute / CDF once all rows are counted
 *   rows   finishes rows [y0, y1) into the same rows of d_out ([H, W, 3]); bands follow each other from row 0; d_out may be d_img
 *          (a band is rewritten only after the CLAHE pass, which runs a blur radius ahead, has read it)
 * Same bytes as s2sr_postprocess_batch_u8_dev on the whole image.  order: S2SR_PP_ORDER_BGR = the bytes are B,G,R (what
 * RealESRGAN.enhance handles, wow_sr.py:85,94; the colour math is always RGB's); S2SR_PP_ORDER_SWAP_OUT = R and B exchanged in
 * the rows written (the job's cvtColor BGR2RGB, wow_sr.py:103, folded into the last pass).  One banded run per handle at a time. */
#define S2SR_PP_ORDER_BGR      1
#define S2SR_PP_ORDER_SWAP_OUT 2
int  s2sr_pp_band_begin_dev(s2sr_handle* h, int32_t H, int32_t W, const s2sr_pp_params* prm, int32_t order, void* stream);
int  s2sr_pp_band_hist_dev(s2sr_handle* h, const void* d_img, int32_t y0, int32_t y1, void* stream);
int  s2sr_pp_band_lut_dev(s2sr_handle* h, void* stream);
int  s2sr_pp_band_rows_dev(s2sr_handle* h, const void* d_img, int32_t y0, int32_t y1, void* d_out, void* stream);

/* ---- XYZ tile pyramid: the step after the path (reference server/app/tiling.py:102-186 shells out to
 * `gdalwarp -t_srs EPSG:3857 -r bilinear` and `gdal2tiles.py --xyz --resampling average`).  The geometry
 * (projection, tile bounds, footprints) is resolved by the caller into tables; tile arrays are
 * [rows north to south][columns][256][256][4] RGBA u8, alpha 0 = no data.
 * warp: grid = float32 [gh][gw][2], the source (column, row) in pixel-centre coordinates at every
 *   `step`-th output pixel (step a power of two, (gh-1)*step >= OH-1); bilinear with edge replication,
 *   alpha = 255 inside the source raster.
 * base: tile pixel = rounded mean of the source pixels with alpha > 0 in columns col_lo..col_hi and
 *   rows row_lo..row_hi (tables of nx*256 and ny*256 entries, lo > hi = empty).  rgba == NULL: the raster is the H x W output the
 *   previous call on this handle -- s2sr_warp_bilinear_u8 -- produced, taken from its device copy (any other call in between
 *   invalidates the copy -> S2SR_E_INVALID).
 * overview: parent pixel = rounded mean of the valid pixels of its 2x2 group in the child array;
 *   (ox, oy) = child-array tile coordinates of the first parent tile's north-west child (may be -1).
 *   child == NULL: the children are the level the previous base / overview call on this handle produced, taken from the
 *   device copy (cnx, cny must match it; any other call on the handle in between invalidates the copy -> S2SR_E_INVALID). */
int  s2sr_warp_bilinear_u8(s2sr_handle* h, const uint8_t* rgb, int32_t H, int32_t W, const float* grid, int32_t gh, int32_t gw,
                           int32_t step, int32_t OH, int32_t OW, uint8_t* out_rgba);
int  s2sr_tiles_base_u8(s2sr_handle* h, const uint8_t* rgba, int32_t H, int32_t W, const int32_t* col_lo, const int32_t* col_hi,
                        const int32_t* row_lo, const int32_t* row_hi, int32_t nx, int32_t ny, uint8_t* out);
int  s2sr_tiles_overview_u8(s2sr_handle* h, const uint8_t* child, int32_t cnx, int32_t cny, int32_t ox, int32_t oy, int32_t pnx,
                            int32_t pny, uint8_t* out);
/* base / overview with out == NULL: the level is computed and stays on the device (for s2sr_tiles_write_png and as the next
 * overview's children).
 * write_png: the PNG files (8-bit RGBA, what s2sr_png_encode writes up to the tokenisation: runs do not cross rows) of the level the
 * previous base / overview call produced, encoded on the device: token statistics and bit emission are kernels, the Huffman codes
 * come from the host between them, only compressed bytes cross PCIe.  paths: nx * ny entries, row-major like the tile array, NULL =
 * skip; flags: S2SR_PNG_SKIP_TRANSPARENT = no file for a tile whose alpha is 0 everywhere, S2SR_PNG_HOST_ENCODER = every tile
 * through the host encoder (the route a tile takes by itself when stored blocks would be smaller; a diagnostic); written
 * (optional): 1 per file written.  Missing parent directories are created. */
#define S2SR_PNG_SKIP_TRANSPARENT 1
#define S2SR_PNG_HOST_ENCODER     2
#define S2SR_PNG_ROW_THREADS      4   /* the first form of the two kernels (one thread walks one row): same bytes, kept as the check */
#define S2SR_PNG_SMALL_GROUPS     8   /* the level goes through in groups of 3 tiles instead of ~2048 (the device phases of a group run under
                                         the host phases of its neighbours): same files; lets a test drive the pipeline on a small level */
int  s2sr_tiles_write_png(s2sr_handle* h, int32_t nx, int32_t ny, const char* const* paths, int32_t flags, int32_t* written);
/* the same for the XYZ layout gdal2tiles writes (tiling.py:138-186): tile (row j, column i) of the level goes to
 * <dir>/<zoom>/<x0 + i>/<y_rows[j]>.png -- the caller hands over ny row numbers instead of nx * ny path strings */
int  s2sr_tiles_write_png_xyz(s2sr_handle* h, int32_t nx, int32_t ny, const char* dir, int32_t zoom, int32_t x0, const int32_t* y_rows,
                              int32_t flags, int32_t* written);

/* measurement: HIP-event timing per kernel family on the launch stream.  on = 0: off;
 * on = N >= 1: every N-th launch of each family is bracketed by a hipEvent pair (N > 1 keeps
 * the event overhead out of a timed region; stats then cover the sampled launches only). */
int  s2sr_set_profiling(s2sr_handle* h, int32_t on);
int  s2sr_get_kernel_stats(s2sr_handle* h, s2sr_kstat* out, int32_t cap, int32_t* n);
int  s2sr_reset_kernel_stats(s2sr_handle* h);
int  s2sr_synchronize(s2sr_handle* h);
/* A group (pack + 351 dependent launches) seen twice with the same shapes, buffers and stream
 * is captured into a hipGraph and replayed afterwards (S2SR_GRAPH=0 disables; the legacy null
 * stream and profiling runs use direct launches).  Counters since s2sr_create. */
int  s2sr_graph_stats(s2sr_handle* h, int64_t* captures, int64_t* replays);

/* host-only codec for the file glue around the path (reference reads LZW GeoTIFFs through rasterio,
 * server/app/wow_sr.py:59-79): TIFF-flavoured LZW (MSB-first 9..12-bit codes, early change).  Decodes
 * at most `cap` bytes into dst, *out_n = bytes produced. */
int  s2sr_tiff_lzw_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n);
/* the encoder for one strip (writes compress="lzw" GeoTIFFs, wow_sr.py:138-151); cap >= n*3/2 + 16 is always enough */
int  s2sr_tiff_lzw_encode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n);

/* host-only PNG encoder for what the path writes (the reference calls cv2.imwrite(path, img) bare, server/app/wow_sr.py:156,163,
 * and hands the tile pyramid to gdal2tiles, server/app/tiling.py:138-186): 8-bit RGB (channels 3) or RGBA (4), filter Sub on
 * every row, deflate with distance-1 matches and dynamic Huffman blocks (cv2's Z_RLE / Z_BEST_SPEED settings, 3-4x zlib's speed).
 * `px`: rows of `width * channels` bytes, `row_stride` bytes apart.  s2sr_png_bound() is always enough for `cap`. */
size_t s2sr_png_bound(int32_t width, int32_t rows, int32_t channels);
/* a complete PNG file (signature, IHDR, one IDAT, IEND) */
int  s2sr_png_encode(const uint8_t* px, int32_t width, int32_t height, int32_t channels, size_t row_stride, uint8_t* out, size_t cap,
                     size_t* out_n);
/* one band of a big image as a complete IDAT chunk, so bands encode on parallel threads: `first` puts the zlib header in front,
 * every band but the `last` ends on a sync flush (byte aligned), the last one on the final block.  The caller writes signature +
 * IHDR, the bands' chunks in order, one more IDAT chunk holding the 4 bytes of the stream's Adler-32 (big endian; combine the
 * per-band values `*adler` over `*raw_n` filtered bytes with adler32_combine), and IEND. */
int  s2sr_png_idat_band(const uint8_t* px, int32_t width, int32_t rows, int32_t channels, size_t row_stride, int32_t first, int32_t last,
                        uint8_t* out, size_t cap, size_t* out_n, uint32_t* adler, size_t* raw_n);
/* `count` square tiles of `size` x `size` pixels, `tile_stride` bytes apart, each written to paths[t] as a PNG file (missing
 * parent directories are created; paths[t] == NULL skips the tile).  skip_transparent: an RGBA tile whose alpha is 0 everywhere is
 * not written (gdal2tiles writes no file for tiles outside the raster, reference server/app/tiling.py:138-186).  written[t]
 * (optional) = 1 for the files written.  One call per tile row keeps the interpreter out of the 12.8k-tile loop of a pyramid. */
int  s2sr_png_write_tiles(const uint8_t* tiles, int32_t count, int32_t size, int32_t channels, size_t tile_stride,
                          const char* const* paths, int32_t skip_transparent, int32_t* written);

/* test hook (host only, no GPU): the OCP e4m3fn encoder the weight packer uses for the fp8
 * correction stages -- round to nearest even, saturating at +-448, NaN -> 0x7f. */
uint8_t s2sr_debug_f32_to_e4m3(float v);
/* test hook (host only): the fp8-trunk weight packer (S2SR_PREC_FP8).  w = [cout][cin][3][3] fp32 -> e4m3 planes of 32 input
 * channels, padded to an even plane count with an all-zero plane: out[plane][tap][ct][16-B half][cout row 0..31][16 bytes] =
 * e4m3(w * 2^k_co); wscale[co] (64 entries) = the E8M0 byte 127 - k_co the MFMA's scale_a operand takes. */
size_t s2sr_debug_pack_f8_bytes(int32_t cin, int32_t cout);
int  s2sr_debug_pack_f8(const float* w, int32_t cin, int32_t cout, uint8_t* out, int32_t* wscale);

/* test hook: one 3x3 conv layer on NCHW fp32 host tensors through the production kernel
 * (upsample != 0 -> nearest-2x on load).  act: 0 none, 1 LeakyReLU(0.2). */
int  s2sr_debug_conv(s2sr_handle* h, const float* x, int32_t N, int32_t Cin, int32_t H, int32_t W,
                     const float* weight, const float* bias, int32_t Cout, int32_t upsample,
                     int32_t act, float* y);

/* test hook: what s2sr_create read from the environment (every kernel-form / scale switch is fixed at creation), so a
 * test that sets S2SR_* can assert the switch took on the handle it then creates. */
typedef struct s2sr_debug_config {
    int32_t precision;      /* S2SR_PREC_* */
    int32_t group;          /* cfg.group as given (0 = default) */
    int32_t trunk_w4;       /* 1: RDB convs on the one-wave-per-SIMD kernels (conv_trunk.hip); 0: 8-wave kernel (S2SR_TRUNK=0) */
    int32_t lo_exp;         /* trunk lo half as e4m3(lo * 2^lo_exp) (S2SR_LO_EXP) */
    int32_t fp8_form;       /* conv_trunk_f8 conv1-4 form bits: 1 no loader wave, 2/4 weight placement, 8 two waves per SIMD */
    int32_t fp8_x_exp, fp8_g_exp;   /* fp8 trunk activation scales (S2SR_FP8_XEXP / _GEXP or s2sr_calibrate_fp8) */
    int32_t fp8_hp_tail;    /* S2SR_FP8_TAIL=hp */
    int32_t graphs_on;      /* S2SR_GRAPH */
    int32_t trunk_wino;     /* 1: fp16 RDB conv1-4 in the row-Winograd F(2,3) form (S2SR_WINO) */
    int32_t reserved[6];    /* [0]: window mosaics on (S2SR_MOSAIC); [1]: fp16 conv1-4 loader-wave form (S2SR_F16_LOADER); [2]: conv_last folded 6-stage form (S2SR_LAST_FOLD); [3]: 4-wave tail convs (S2SR_TAIL_W4); [4]: whole-patch fp16 conv1-4 forms allowed (S2SR_F16_FULL); [5]: workspace allocations since s2sr_create */
} s2sr_debug_config;
int  s2sr_debug_get_config(s2sr_handle* h, s2sr_debug_config* out);

/* How `B` equal windows of th x tw travel through the net: kx x ky per launch image with one zero row / column between neighbours
 * (1 x 1: one window per image -- sizes that are multiples of the 32-pixel patch gain nothing); the windows past the last full
 * mosaic travel as ONE smaller mosaic.  Chosen for the fewest launched patches.  Host arithmetic only. */
int  s2sr_debug_pick_mosaic(int32_t B, int32_t th, int32_t tw, int32_t* kx, int32_t* ky);
/* ... and what that choice LAUNCHES: 32 x 32 patches of floor(B / (kx*ky)) full mosaics plus the remainder's smaller mosaic
 * (`launched`), next to B plain images (`plain`).  The engine only takes a mosaic when launched <= 0.98 plain.  Host arithmetic only. */
int  s2sr_debug_mosaic_patches(int32_t B, int32_t th, int32_t tw, int64_t* launched, int64_t* plain);
/* The chunk plan of a tiled s2sr_enhance_u8 (host arithmetic only, no device needed): `units` row units of `unit_windows` windows
 * each, at most `u_max` units per chunk, `per` windows per launch image (mosaic), `pimg` 32x32 patches per launch image, `ncu`
 * workgroups.  Writes the chunk sizes front to back; *n = their number (cap 0: count only). */
int  s2sr_debug_plan_chunks(int32_t units, int32_t u_max, int32_t unit_windows, int32_t per, int32_t pimg, int32_t ncu,
                            int32_t* sizes, int32_t cap, int32_t* n);

/* test hook: ONE RDB-shaped conv through the TRUNK kernels (conv_trunk.hip: conv_trunk_f16 / conv_trunk_f8), host tensors in
 * NCHW fp32 -- the per-layer parity check of the kernels that carry 84 % of a step (s2sr_debug_conv goes through conv3x3.hip).
 *   kind 0: fp16 conv1-4 form   y = lrelu(conv(x) + b)                        Cin in {64,96,128,160}, Cout 32, y = the fp16 plane written
 *   kind 1: fp16 conv5 form     y = 0.2*(conv(x) + b) + (x[:, :64] + lo)     Cin 192, Cout 64, y = hi + lo of the (fp16, e4m3) pair written
 *   kind 2: fp16 conv5 of rdb3  y = 0.2*(kind 1) + skip
 *   kind 3: fp8 conv1-4 form    y = e4m3(lrelu(conv + b) * 2^g_exp) / 2^g_exp  (x planes at 2^x_exp, growth planes at 2^g_exp)
 *   kind 4: fp8 conv5 form      y = fp16(0.2*(conv + b) + x[:, :64]); y_aux = its e4m3(* 2^x_exp) image
 *   kind 5: fp8 conv5 of rdb3   y = fp16(0.2*(kind 4 value) + skip)
 * x is rounded to the operand format on the way in (fp16, or e4m3 at the handle's scales), so callers pass representable
 * values; `lo` ([N,64,H,W], kinds 1-2, may be NULL) is stored as e4m3(lo * 2^lo_exp); `skip` ([N,64,H,W]) as the
 * (fp16 hi, e4m3 lo) pair (kind 2) or fp16 (kind 5).  form: kind 0: 0 auto, 1 = 16x32 patches, 2 = 32x32 patches,
 * 3 = row-Winograd F(2,3), 4 = 32x32 patches with the load-only fifth wave, 5 = 8x32 patches (single tiles), 10 = 8x32 patches with two
 * planes per pipeline stage; kinds 1-2: 0 auto, 1 = 16x32 patches, 5 = 8x32 patches, 10 = 8x32 patches with two planes per stage; kind 3:
 * the fp8_form bits.  (3, 4, 9 and the long lo-encoding form 2 of kinds 1-2: experimental library only.) */
typedef struct s2sr_debug_trunk_args {
    int32_t kind, form;
    int32_t N, Cin, H, W;
    const float* x;
    const float* weight;    /* [Cout,Cin,3,3] */
    const float* bias;      /* [Cout] */
    const float* lo;
    const float* skip;
    float* y;               /* [N,Cout,H,W] */
    float* y_aux;           /* kinds 4-5: [N,64,H,W], may be NULL */
} s2sr_debug_trunk_args;
int  s2sr_debug_conv_trunk(s2sr_handle* h, const s2sr_debug_trunk_args* a);

/* diagnostic: time one RDB-shaped conv (cin in {64,96,128,160,192}; cout 32 -> conv1..4 form,
 * cout 64 -> conv5 form) over N images of HxW, `iters` launches; avg_us = mean launch time from
 * HIP events.  If trace != NULL, one extra launch of the stamped diagnostic build fills
 * trace[wg*24 + k] with s_memtime ticks for the first trace_wgs workgroups. */
int  s2sr_debug_bench_conv(s2sr_handle* h, int32_t N, int32_t H, int32_t W, int32_t cin, int32_t cout,
                           int32_t iters, float* avg_us, uint64_t* trace, int32_t trace_wgs);

/* diagnostic: what the matrix pipe sustains on THIS part at its power cap, for the roofline claim of the fp16 trunk kernel
 * (csrc/ceiling.hip).  mode 0: a bare v_mfma_f32_32x32x16_f16 loop, operands in registers; 1: the same loop with its operands
 * re-read from LDS at conv_trunk_f16's 0.75 KiB per MFMA; 2: + the LDS ring refilled by LDS-DMA at the kernel's 48 KiB per 288
 * MFMAs from a 336-MB buffer (3-deep ring, counted vmcnt, one barrier per stage); 3: as 2 with half the fill (24 KiB); 4: as 2 from
 * an 8-MB source that stays in L2 / MALL (the fill without the HBM side); 5: the kernel's own mix -- 36 KiB streamed from HBM, 12 KiB
 * from the cached source (the weights), 8 KiB stored per stage (its output); 6: conv5's mix -- 384 MFMAs per stage (64 output channels),
 * 0.44 LDS reads per MFMA, 32 KiB streamed, 16 KiB cached, 12 KiB stored; 7 / 8: as 2 from a 100-MB / 200-MB source (past the L2s, inside
 * the Infinity Cache: the dense tensor of a launch group of 4 / 8 images).  One workgroup per CU, random fp16 operands;
 * `launches` back-to-back launches of `stages` stages per workgroup behind launches / 4 + 1 untimed ones; *ms_total = their
 * time by HIP events, *flop_per_launch / *dma_bytes_per_launch = the work of one (28 stages = one conv1-4 launch of 16 images). */
int  s2sr_debug_mfma_ceiling(s2sr_handle* h, int32_t mode, int32_t stages, int32_t launches, double* flop_per_launch,
                             double* dma_bytes_per_launch, float* ms_total);

/* diagnostic prototype (csrc/persist.hip; nothing of the product calls it): what would a trunk schedule sustain whose workgroups stay across the
 * layers of the RDBs, planes handed to the neighbours through flags, on a launch group small enough for the Infinity Cache?  `grid` workgroups
 * (at most one per CU, on an otherwise idle device: they must all be resident) of `P` = 2..4 patches each run `rdbs` RDB-shaped rounds per launch
 * (per patch 28 stages of 288 MFMAs + 12 of 576, 48 KiB of LDS-DMA per stage, the layer's planes stored behind each patch; working set
 * grid x P x 512 KiB); variant bit 0: plane loads with sc1, plane stores with sc0 sc1; bit 1: the same work on 32-KiB stages in a 4-deep ring
 * (three stages of look-ahead instead of two); variant 4 / 5: variant 0 / 1 with the hand-over CHECKED -- the plane stores carry (layer count,
 * writer) and every landed piece is compared: mismatches[0] = halo pieces (written by a workgroup on another XCD), mismatches[1] = own pieces.
 * *timeouts: dependency waits that ran into their bound (must be 0 for the timing to mean anything). */
int  s2sr_debug_rdb_persistent(s2sr_handle* h, int32_t variant, int32_t grid, int32_t P, int32_t rdbs, int32_t launches, double* flop_per_launch,
                               float* ms_total, int32_t* timeouts, int32_t* mismatches);

#ifdef __cplusplus
}
#endif
#endif /* S2SR_H */
