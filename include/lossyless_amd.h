/*
 * lossyless_amd.h -- C ABI of liblossyless_amd.so (MI355X / gfx950).
 *
 * Drop-in boundary for the compress_dataset hot path of YannDubs/lossyless
 * (hub/compressor.py).  The reference reaches native code for this path through
 * three pybind11 entry points of compressai==1.1.5 and through torch/cuDNN for the
 * CLIP tower; each function below names the reference interface it stands in for.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++ / torch types.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Device
 *     entry points only enqueue work; they never synchronise.
 *   - every function returns LLA_OK (0) or a negative LLA_E* code; nothing throws.
 *   - pointers marked [dev] are device pointers, [host] host pointers.
 *   - no global state; re-entrant; caller owns all buffers.  The only state a call may leave behind lives in
 *     handles the caller creates and destroys (lla_tower_*, lla_profiler_*); a handle is used from one host
 *     thread at a time.  (One process-wide cache exists and is guarded: per-device kernel attributes, set once.
 *     The product library reads no environment variable: csrc/switches.h.)
 *
 * Table layout (same as compressai's EntropyModel buffers after update(),
 * hub/compressor.py:56-63):
 *   cdf      int32 [C][W]   row c holds cdf_len[c] valid entries, 0 .. 65536
 *   cdf_len  int32 [C]      = pmf_length + 2
 *   offset   int32 [C]      = -minima
 * Escape symbol of channel c is index cdf_len[c]-2.
 *
 * Domain: |symbol - offset| < 2^30 (the reference's int32 arithmetic is undefined
 * beyond that, rans_interface.cpp encode_with_indexes).
 */
#ifndef LOSSYLESS_AMD_H
#define LOSSYLESS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LLA_OK 0
#define LLA_EINVAL (-1) /* bad argument (null pointer, size, alignment)        */
#define LLA_ECAP (-2)   /* caller buffer too small                             */
#define LLA_EHIP (-3)   /* HIP runtime reported an error (see lla_last_hip_error) */
#define LLA_EDATA (-4)  /* malformed input (e.g. pmf without a donor frequency) */

/* v4 (round 6): + lla_source_sha(); enum lla_vit_param lost its six LayerNorm-folded weight entries (the blob of
 * lla_vit_b32_weights_bytes() is 99 MB smaller): the algebraic LayerNorm fusion of round 3 they fed was retired. */
#define LLA_ABI_VERSION 4

/* ABI version of the loaded library. */
int lla_abi_version(void);
/* sha256 (first 16 hex digits) over the kernel sources this library was built from (the .hip / .h / .cpp files of csrc + this
 * header; lossyless_amd/csrc/source_sha.py).  The Python host recomputes it from the tree and refuses a stale build. */
const char *lla_source_sha(void);
/* hipError_t of the most recent failing HIP call on this thread (0 if none). */
int lla_last_hip_error(void);

/* ------------------------------------------------------------------------- *
 * Host entry points
 * ------------------------------------------------------------------------- */

/* Replaces compressai._CXX.pmf_to_quantized_cdf(pmf: list[float], precision) ->
 * list[int]  (cpp_exts/ops/ops.cpp), reached from EntropyBottleneck.update(),
 * hub/compressor.py:63; lossyless/rates.py:299.
 * pmf [host] n floats; cdf_out [host] n+1 entries. */
int lla_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf_out);

/* Walks the reference's container (hub/compressor.py:233-237: big-endian u32 N, then
 * N x {big-endian u32 len, len bytes}) held in host memory and writes the byte
 * offset of every record, relative to blob + 4, into off[0..N] (off[N] = body size),
 * i.e. exactly the `off` / record_prefix=1 input of lla_rans_decode_batch for the
 * body blob + 4.  *n_out receives N.  off may be NULL to only query N.
 * Returns LLA_EDATA when a record runs past nbytes or when N records cannot fit in nbytes at all
 * (checked even with off == NULL, so a corrupt count never sizes an allocation), LLA_ECAP when
 * off_cap < N+1. */
int lla_container_index(const uint8_t *blob, size_t nbytes, uint64_t *off, size_t off_cap,
                        uint32_t *n_out);

/* Host coder: the batch forms of
 *   ans.RansEncoder().encode_with_indexes(symbols, indexes, cdfs, cdfs_sizes, offsets) -> bytes
 *   ans.RansDecoder().decode_with_indexes(encoded, indexes, cdfs, cdfs_sizes, offsets) -> list[int]
 * (compressai cpp_exts/rans/rans_interface.cpp) with indexes[i] = i, on HOST pointers, threaded
 * over images.  They serve the reference's default decompress_dataset(is_cpu=True)
 * (hub/compressor.py:227-229,236-238: module moved to the CPU, one decode per image) on a box
 * without a GPU, and give CPU-side tools the same bytes as the device kernels.
 *   encode: symbols [host] B*C; out receives the streams back to back (record_prefix = 1: each
 *           preceded by its big-endian u32 length = the container body, hub/compressor.py:192-196);
 *           out_off [host] B+1 byte offsets, out_off[B] = total.  LLA_ECAP if total > cap (out_off
 *           is filled in that case too, so the caller can size `out` and call again).
 *   decode: payload/off/record_prefix/status exactly as lla_rans_decode_batch, host pointers. */
int lla_rans_encode_batch_host(const int32_t *symbols, int B, int C, const int32_t *cdf, int W,
                               const int32_t *cdf_len, const int32_t *offset, int record_prefix,
                               uint8_t *out, size_t cap, uint64_t *out_off);
int lla_rans_decode_batch_host(const uint8_t *payload, const uint64_t *off, int record_prefix, int B,
                               int C, const int32_t *cdf, int W, const int32_t *cdf_len,
                               const int32_t *offset, int32_t *symbols_out, int32_t *status);
/* Host twin of lla_dequantise (same fp32 operations, same values). */
int lla_dequantise_host(const int32_t *symbols, int B, int C, const float *bias,
                        const float *exp_scale, const float *median, float *z_hat);

/* Upper bound on the bytes one image's stream can occupy (C symbols, all
 * escaped with 8 payload digits).  Use it as the per-image scratch stride. */
size_t lla_rans_max_encoded_bytes(int C);

/* ------------------------------------------------------------------------- *
 * Device entry points: entropy stage (A4, A5, A13, A14, A15)
 * ------------------------------------------------------------------------- */

/* z dtype selector for the fused entry points */
#define LLA_Z_F16 1
#define LLA_Z_F32 2

/* process_z_in + quantise (hub/compressor.py:105-109 then EntropyModel.quantize
 * "symbols"): sym = int(rint((float(z) + bias) * exp_scale - median)), every
 * operation rounded to fp32 on its own.
 * z [dev] B*C of z_dtype; bias/exp_scale/median [dev] C floats; symbols [dev] B*C. */
int lla_quantise(const void *z, int z_dtype, int B, int C, const float *bias,
                 const float *exp_scale, const float *median, int32_t *symbols, void *stream);

/* Replaces ans.RansEncoder().encode_with_indexes(symbols, indexes, cdfs,
 * cdfs_sizes, offsets) -> bytes  (cpp_exts/rans/rans_interface.cpp), called once
 * per image by EntropyModel.compress <- hub/compressor.py:98; lossyless/rates.py:559.
 * Here: B images per call, indexes[i] = i.
 * Image b's stream is written END-ALIGNED into scratch + b*stride, i.e. it
 * occupies [b*stride + stride - lengths[b], b*stride + stride).
 * stride must be >= lla_rans_max_encoded_bytes(C) and a multiple of 4. */
int lla_rans_encode_batch(const int32_t *symbols, int B, int C, const int32_t *cdf, int W,
                          const int32_t *cdf_len, const int32_t *offset, uint8_t *scratch,
                          size_t stride, uint32_t *lengths, void *stream);

/* Fused lla_quantise + lla_rans_encode_batch: z never leaves the device as
 * symbols.  symbols_out [dev] may be NULL. */
int lla_quantise_encode(const void *z, int z_dtype, int B, int C, const float *bias,
                        const float *exp_scale, const float *median, const int32_t *cdf, int W,
                        const int32_t *cdf_len, const int32_t *offset, uint8_t *scratch,
                        size_t stride, uint32_t *lengths, int32_t *symbols_out, void *stream);

/* Workspace bytes lla_rans_compact needs for B images. */
size_t lla_rans_compact_workspace_bytes(int B);

/* Packs the end-aligned streams back to back.
 *   record_prefix = 0: out = s_0 s_1 ... ;            out_off[b] = sum_{k<b} len_k
 *   record_prefix = 1: out = be32(len_0) s_0 be32(len_1) s_1 ... which is the body
 *       of the reference's container after its 4-byte count
 *       (hub/compressor.py:192-196);                  out_off[b] = sum_{k<b} (len_k + 4)
 * out_off [dev] B+1 entries, out_off[B] = total bytes.  The caller sizes `out`
 * for the worst case (B*(stride+4)) or reads lengths first; bytes beyond cap are
 * not written and *out_off[B] still reports the needed size. */
int lla_rans_compact(const uint8_t *scratch, size_t stride, const uint32_t *lengths, int B,
                     int record_prefix, uint8_t *out, size_t cap, uint64_t *out_off,
                     void *workspace, size_t workspace_bytes, void *stream);

/* Replaces ans.RansDecoder().decode_with_indexes(encoded, indexes, cdfs,
 * cdfs_sizes, offsets) -> list[int], called per image by EntropyModel.decompress
 * <- hub/compressor.py:124,238; lossyless/rates.py:563.
 * payload [dev]; image b's stream is payload[off[b] + skip .. off[b+1]) where
 * skip = 4 when record_prefix (the be32 length is stepped over).  Streams must
 * start 4-byte aligned.  status [dev] B ints: 0 ok, 1 stream overrun. */
int lla_rans_decode_batch(const uint8_t *payload, const uint64_t *off, int record_prefix, int B,
                          int C, const int32_t *cdf, int W, const int32_t *cdf_len,
                          const int32_t *offset, int32_t *symbols_out, int32_t *status,
                          void *stream);

/* The same two coder calls with an arbitrary table row per symbol, i.e. the full
 * ans.RansEncoder().encode_with_indexes(symbols, indexes, cdfs, cdfs_sizes, offsets) /
 * ans.RansDecoder().decode_with_indexes(...) signatures as GaussianConditional.compress /
 * .decompress use them (compressai EntropyModel.compress; lossyless/rates.py:704-729, scale
 * table rows selected by build_indexes) -- SURVEY.md 8(f) rank 4.
 * symbols, indexes [dev] B*n int32 (string b = elements [b*n, (b+1)*n)); cdf [dev] T*W int32,
 * cdf_len/offset [dev] T.  Rows are read from global memory (no size limit on T*W); an index
 * outside [0, T) is clamped.  scratch/stride/lengths and payload/off/record_prefix/status as
 * in lla_rans_encode_batch / lla_rans_decode_batch (stride >= lla_rans_max_encoded_bytes(n)). */
int lla_rans_encode_indexed(const int32_t *symbols, const int32_t *indexes, int B, int n,
                            const int32_t *cdf, int T, int W, const int32_t *cdf_len,
                            const int32_t *offset, uint8_t *scratch, size_t stride,
                            uint32_t *lengths, void *stream);
int lla_rans_decode_indexed(const uint8_t *payload, const uint64_t *off, int record_prefix, int B,
                            int n, const int32_t *indexes, const int32_t *cdf, int T, int W,
                            const int32_t *cdf_len, const int32_t *offset, int32_t *symbols_out,
                            int32_t *status, void *stream);

/* EntropyModel.dequantize + process_z_out (hub/compressor.py:111-115):
 * z_hat = (float(sym) + median) / exp_scale - bias, fp32 per operation. */
int lla_dequantise(const int32_t *symbols, int B, int C, const float *bias,
                   const float *exp_scale, const float *median, float *z_hat, void *stream);

/* compressor(X) without coding: process_z_in -> EntropyBottleneck.forward (eval:
 * round(z_in - median) + median) -> process_z_out, hub/compressor.py:95,100-101. */
int lla_represent(const void *z, int z_dtype, int B, int C, const float *bias,
                  const float *exp_scale, const float *median, float *z_hat, void *stream);

/* ------------------------------------------------------------------------- *
 * Device entry point: CLIP preprocessing
 *   stands in for  compressor.preprocess  = clip._transform (clip==1.0): Resize(224,
 *   BICUBIC) -> CenterCrop(224) -> ToTensor -> Normalize, which the reference applies per
 *   image with PIL in DataLoader workers (hub/compressor.py:155,186;
 *   utils/data/images.py:383-411).  Bit-exact against Pillow's 8-bit resampler.
 * ------------------------------------------------------------------------- */

/* images [dev] uint8 [B][H][W][3] (RGB, all the same size).  The tap tables are built on the
 * host exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do (see
 * lossyless_amd/preprocess.py) for the 224 output columns / rows that survive the centre
 * crop: bounds int32 [224][2] = (first tap, tap count), coef int32 [224][ksize] 22-bit fixed
 * point; v_bounds index rows of the RESIZED-width intermediate, i.e. input rows; only input
 * rows [row0, row0+nrows) are touched.  mean3 / std3 [host] floats.  out fp16 [B][224][224][3].
 * workspace [dev] >= lla_preprocess_workspace_bytes(B, nrows). */
size_t lla_preprocess_workspace_bytes(int B, int nrows);
int lla_preprocess_clip(const uint8_t *images, int B, int H, int W, int row0, int nrows,
                        const int32_t *h_bounds, const int32_t *h_coef, int h_ksize,
                        const int32_t *v_bounds, const int32_t *v_coef, int v_ksize,
                        const float *mean3, const float *std3, void *workspace,
                        size_t workspace_bytes, void *out_nhwc_f16, void *stream);

/* Pillow's resampling tap tables on the host -- Resample.c precompute_coeffs (bicubic, a = -0.5, box = the
 * whole axis) followed by normalize_coeffs_8bpc, in the same double arithmetic -- for output positions
 * first .. first+count-1 of a resize in_size -> out_size:
 *   bounds [count][2] = (first tap, number of taps), coef [count][ksize] 22-bit fixed point, rows zero
 *   padded; ksize = lla_pillow_bicubic_ksize(in_size, out_size) = 2 ceil(2 max(in/out, 1)) + 1.
 * A tap TABLE as the ragged entry point below takes it is bounds[224][2] followed by coef[224][ksize]. */
int lla_pillow_bicubic_ksize(int in_size, int out_size);
int lla_pillow_bicubic_taps(int in_size, int out_size, int first, int count, int ksize, int32_t *bounds,
                            int32_t *coef);

/* Ragged batches (every image its own size: ImageNet-style folders, BASELINE configs[2]) -- what the
 * reference's per-image `compressor.preprocess` handles by construction (hub/compressor.py:162-165).
 * One descriptor per image; pointers are DEVICE pointers.  Pixels are RGB, 3 bytes per pixel, rows
 * contiguous; the 3 bytes after an image's last byte must be readable (rows are fetched as aligned
 * dwords).  h_table / v_table: tap tables (see lla_pillow_bicubic_taps) of the 224 cropped output columns /
 * rows; vertical bounds index source rows.  Same bytes out as lla_preprocess_clip on each image alone. */
typedef struct lla_image_desc {
  const uint8_t *pixels;
  const int32_t *h_table;
  const int32_t *v_table;
  int32_t H, W;
  int32_t h_ksize, v_ksize;
} lla_image_desc;
/* LDS bytes one workgroup needs for an image with these (HOST) tables when a band is `band_rows` output rows
 * (a divisor-friendly height such as 28, 14, 7, 4, 2, 1).  The caller passes the maximum over the images of
 * a launch as lds_bytes; LLA_ECAP if that exceeds what the device gives one workgroup (such images go
 * through lla_preprocess_clip one at a time: its two-pass path has no size limit). */
size_t lla_preprocess_ragged_lds_bytes(const int32_t *h_table, int h_ksize, const int32_t *v_table,
                                       int v_ksize, int band_rows);
/* descs [dev] B descriptors; out fp16 [B][224][224][3]. */
int lla_preprocess_clip_ragged(const lla_image_desc *descs, int B, int band_rows, size_t lds_bytes,
                               const float *mean3, const float *std3, void *out_nhwc_f16, void *stream);

/* Synthetic workload (BASELINE.json configs[3]: 1M x 224 x 224 x 3 images sharded across ranks, never
 * materialised): images first_image .. first_image+count-1 of the virtual dataset `seed`, CLIP-normalised fp16
 * NHWC [count][224][224][3].  Element e of the whole virtual tensor is
 *   h = (e ^ (seed * 0x9E3779B97F4A7C15 & (2^63 - 1))) * 0x2545F4914F6CDD1D;  h ^= h >> 29 (int64);
 *   h *= 0x94D049BB133111EB;  u8 = (h >> 40) & 255;  value = fp16((u8 / 255 - mean[c]) / std[c])
 * -- a pure function of (seed, e), so every sharding of the image range generates the same bytes. */
int lla_synthetic_images(uint64_t seed, uint64_t first_image, int count, const float *mean3, const float *std3,
                         void *out_nhwc_f16, void *stream);

/* ------------------------------------------------------------------------- *
 * Device entry points: CLIP ViT-B/32 visual tower (A10)
 *   stands in for  z = self.clip(X)  at hub/compressor.py:93
 *   (clip.model.VisionTransformer.forward, clip==1.0).
 * ------------------------------------------------------------------------- */

#define LLA_LAYOUT_NHWC 0 /* images [B][224][224][3] fp16 */
#define LLA_LAYOUT_NCHW 1 /* images [B][3][224][224] fp16 (what the reference feeds) */

/* Parameter ids of the weight blob.  Matrices are fp16 in torch Linear layout
 * [out][in]; vectors (LayerNorm, biases, embeddings) are fp32. */
enum lla_vit_param {
  LLA_VIT_CONV1_NHWC = 0, /* fp16 [768][32][32][3]   (kh,kw,c) K-order          */
  LLA_VIT_CONV1_NCHW,     /* fp16 [768][3][32][32]   OpenAI order               */
  LLA_VIT_CLASS_EMB,      /* fp32 [768]                                         */
  LLA_VIT_POS_EMB,        /* fp32 [50][768]                                     */
  LLA_VIT_LN_PRE_W,       /* fp32 [768]                                         */
  LLA_VIT_LN_PRE_B,
  LLA_VIT_LN_POST_W,
  LLA_VIT_LN_POST_B,
  LLA_VIT_PROJ_T,         /* fp16 [512][768] = proj.T                           */
  LLA_VIT_GLOBAL_COUNT,
  /* per-layer ids, use with layer = 0..11 */
  LLA_VIT_LN1_W = 16, LLA_VIT_LN1_B,
  LLA_VIT_QKV_W,          /* fp16 [2304][768] attn.in_proj_weight               */
  LLA_VIT_QKV_B,          /* fp32 [2304]                                        */
  LLA_VIT_OUT_W,          /* fp16 [768][768]  attn.out_proj.weight              */
  LLA_VIT_OUT_B,
  LLA_VIT_LN2_W, LLA_VIT_LN2_B,
  LLA_VIT_FC_W,           /* fp16 [3072][768] mlp.c_fc.weight                   */
  LLA_VIT_FC_B,
  LLA_VIT_CPROJ_W,        /* fp16 [768][3072] mlp.c_proj.weight                 */
  LLA_VIT_CPROJ_B,
  LLA_VIT_LAYER_END
};

/* Total bytes of the weight blob. */
size_t lla_vit_b32_weights_bytes(void);
/* Byte offset / byte size of one parameter inside the blob (layer ignored for
 * global ids).  Returns (size_t)-1 for an unknown id. */
size_t lla_vit_b32_param_offset(int param, int layer);
size_t lla_vit_b32_param_bytes(int param);

/* Workspace bytes for a pass that processes `chunk` images at a time (one set of slice buffers: the product library
 * runs the tower on ONE stream; the ablation build doubles it when its two lanes are switched on). */
size_t lla_vit_b32_workspace_bytes(int chunk);

/* images [dev] fp16 in `layout`, CLIP-normalised; weights [dev] blob;
 * z_out [dev] fp16 [B][512].  The batch is walked in slices of at most `chunk` images
 * (chunk <= 0: library default 8704 = 1700 row tiles of 256; capped at 65536; a ragged slice of >= 256 images is cut once more
 * into a multiple of 128 images + the rest: whole 256-row tiles for the residual layers' four-wave GEMM), all on `stream`.  Re-entrant: no
 * state outside the arguments. */
int lla_vit_b32_forward(const void *images, int layout, int B, const void *weights,
                        void *workspace, size_t workspace_bytes, int chunk, void *z_out,
                        void *stream);

/* Tower handle: two HIP streams ("lanes") of the CURRENT device plus the events that fork them from and
 * join them into the caller's stream.  The lanes are the only state the tower entry points keep between
 * calls, and it lives here, in an object the caller owns (one handle per device and driving thread; the
 * entry points taking a handle are not thread-safe with respect to that handle). */
int lla_tower_create(void **tower);
int lla_tower_destroy(void *tower);   /* waits for the lanes to drain */

/* Options of a tower handle (ABI v3).  They select between code paths that give the SAME embeddings bit for bit -- what
 * tests/test_gpu_vit.py asserts with them; nothing here is needed for production use.
 *   LLA_TOWER_OPT_LNX       1 (default): slices of whole 256-row tiles apply the LayerNorm that follows a residual GEMM
 *                           (ln_2 after out-proj, ln_1 of the next block after c_proj: clip VisionTransformer,
 *                           hub/compressor.py:93) in that GEMM's epilogue; 0: layernorm768_kernel after every such GEMM.
 *   LLA_TOWER_OPT_LNX_WAIT  shader cycles a column tile waits there for the two other column tiles of its rows
 *                           (default 24000); < 0: never -- every row tile is normalised by the clean-up kernel. */
#define LLA_TOWER_OPT_LNX 1
#define LLA_TOWER_OPT_LNX_WAIT 2
int lla_tower_set_option(void *tower, int option, int value);

/* The same pass through a tower handle.  PRODUCT LIBRARY (since round 4): everything runs on `stream`, `deferred` only
 * means "the caller joins later", and lla_tower_join is a cheap no-op dependency -- same results, same order.  The
 * two-lane mode described below exists in the tools/ build only (`make ablation`, LLA_VIT_STREAMS=2 there: csrc/switches.h): with two hardware
 * queues active the tower is not bit-reproducible on this stack -- between one embedding in 10^6 and one in 10^8
 * images (box and build dependent) differs by a few fp16 ulps between runs, docs/history/DESIGN_rounds_1-5.md 5.3 -- and bit-identical records
 * for identical inputs are this path's contract.  The entry points stay so that callers written against ABI v2 keep
 * working unchanged.
 *   deferred = 0: batches of >= 640 images (LLA_VIT_SPLIT_MIN) are cut into at least two slices that
 *     alternate between the lanes, forked from and joined back into `stream` with events, when the
 *     workspace holds two slices: one lane's GEMM tails and HBM-bound kernels overlap the other's GEMMs.
 *   deferred = 1: no closing join, for callers that run many batches back to back (RecordStream /
 *     compress_dataset): whole slices alternate between the lanes ACROSS calls, so one batch's last GEMM
 *     rounds overlap the next batch's first kernels.  `z_out` (and the lanes' use of `images`) is complete
 *     on `stream` only after lla_tower_join(tower, stream); a later non-deferred pass on the same handle
 *     joins as well.
 * A workspace of one slice also keeps everything on `stream`. */
int lla_vit_b32_forward_lanes(void *tower, const void *images, int layout, int B, const void *weights,
                              void *workspace, size_t workspace_bytes, int chunk, void *z_out,
                              void *stream, int deferred);
/* `stream` waits for everything queued on the handle's lanes so far. */
int lla_tower_join(void *tower, void *stream);

/* The same pass over a batch that lies in `n_pieces` separate device arrays of `piece_images` images each (the last one
 * may hold fewer): what a caller has who gathers equal batches -- the reference's loop hands over one DataLoader batch
 * at a time, hub/compressor.py:186-197 -- into one chip-filling tower pass, without copying them into one array first
 * (301 KB per image read and written again: 2 % of a pass).  Only the patch-embedding GEMM reads the images; it walks
 * them in 256-row tiles, and 256 images are 49 whole tiles, so with piece_images % 256 == 0 no tile straddles two
 * pieces (a caller whose batches are larger multiples of 256 passes piece_images = 256 and one pointer per 256 images, so
 * that a tower pass may also end in the middle of a batch).  Same embeddings, bit for bit, as lla_vit_b32_forward on the
 * concatenated batch.
 * LLA_EINVAL unless 1 <= n_pieces <= 64, piece_images % 256 == 0, (n_pieces - 1) piece_images < B <= n_pieces piece_images,
 * B % 128 == 0, 256 <= B <= the library's slice size (8704), and the workspace holds one slice of B images: the caller
 * then copies (lla_vit_b32_forward).  Everything runs on `stream`; `pieces` (host array of device pointers) is read
 * before the call returns, the images until the pass has run. */
int lla_vit_b32_forward_gather(void *tower, const void *const *pieces, int n_pieces, int piece_images, int layout, int B,
                               const void *weights, void *workspace, size_t workspace_bytes, void *z_out, void *stream);

/* Optional per-kernel-class timing with HIP events recorded on the launch stream
 * (what bench.py's `roofline` object is computed from).  A profiler owns a pool of
 * event pairs; every kernel launched by lla_vit_b32_forward_profiled is bracketed by
 * one pair.  lla_profiler_collect synchronises the recorded events, accumulates
 * per class  ms[c] += elapsed, work[c] += algorithmic FLOPs (GEMM: 2*M*N*K,
 * attention: 4*50*50*64 per head) or bytes (LayerNorm: bytes read + written),
 * launches[c] += 1, and rewinds the pool. */
#define LLA_PROF_GEMM 0
#define LLA_PROF_LAYERNORM 1
#define LLA_PROF_ATTENTION 2
#define LLA_PROF_CLASSES 3
int lla_profiler_create(void **profiler, int max_launches);
int lla_profiler_destroy(void *profiler);
int lla_profiler_collect(void *profiler, double *ms, double *work, long long *launches);
/* Same as lla_vit_b32_forward (one stream, so that a kernel's duration is its own); profiler may be NULL. */
int lla_vit_b32_forward_profiled(const void *images, int layout, int B, const void *weights,
                                 void *workspace, size_t workspace_bytes, int chunk, void *z_out,
                                 void *stream, void *profiler);

/* Building blocks of the tower, exported for per-kernel parity tests and
 * profiling (same kernels lla_vit_b32_forward launches). */
#define LLA_EPI_F16 0          /* C16 = acc (+bias)                     */
#define LLA_EPI_QUICKGELU_F16 1 /* C16 = quickgelu(acc + bias)           */
#define LLA_EPI_RESID_F32 2    /* C32 += acc + bias                     */
#define LLA_EPI_RELU_F16 4     /* C16 = relu(acc + bias)               */
#define LLA_EPI_ADD_RELU_F16 5 /* C16 = relu(acc + bias + R16)         */
/* C[M][N] (+)= A[M][K] * W[N][K]^T ; A, W fp16 row-major; bias fp32 [N] or NULL.
 * N % 128 == 0, K % 64 == 0. */
int lla_gemm_f16(const void *A, const void *W, const float *bias, void *C, int M, int N, int K,
                 int epilogue, void *stream);
/* The same with explicit row strides (elements; lda % 8 == 0, ldc % 4 == 0; with the ReLU epilogues ldc < N,
 * a multiple of 32, stores only the first ldc columns: narrow convolution outputs keep a narrow pitch) and, for
 * LLA_EPI_ADD_RELU_F16, an fp16 matrix R [M][ldr] added before the ReLU (the identity branch of a
 * ResNet bottleneck).  Used by the RN50-CLIP tower below, where 1x1 convolutions are GEMMs over NHWC
 * activations with a channel pitch. */
int lla_gemm_f16_ex(const void *A, int lda, const void *W, const float *bias, void *C, int ldc,
                    const void *resid, int ldr, int M, int N, int K, int epilogue, void *stream);

/* fp32 Linear layer on the fp32 matrix cores: C[M][N] = A[M][K] * W[N][K]^T (+ bias) (ReLU if `relu`), all
 * fp32 row-major with row strides lda / ldw / ldc (elements, multiples of 4; K % 8 == 0, N % 4 == 0 -- pad with
 * zeros).  Stands in for the `nn.Linear` layers of the reference's hyperprior networks, which run in fp32 under
 * autocast(False) (lossyless/rates.py:104,631-639,687-699; lossyless/architectures.py:94-168): fp32 operands and
 * accumulation, one rounding per product (v_mfma_f32_32x32x2_f32 = an fma chain), K order fixed by the kernel
 * alone, so the values do not depend on the batch size (encoder and decoder may use different ones). */
int lla_gemm_f32(const float *A, int lda, const float *W, int ldw, const float *bias, float *C, int ldc,
                 int M, int N, int K, int relu, void *stream);

/* out[n][H][W][ldc] (first cout channels) = relu(conv3x3(in, stride 1, pad 1) + bias) as an IMPLICIT GEMM:
 * `in` is NHWC fp16 [n][H][W][pitch] (first cin channels used; cin % 64 == 0, or cin == 32), weights fp16
 * [cout][K] with K = 9 cin rounded up to a multiple of 64 (zero padded) in the order (kh, kw, c), bias fp32
 * [cout] (BatchNorm folded in), cout % 128 == 0 (weight rows; ldc < cout, a multiple of 32, stores only the
 * first ldc channels).  The A operand is
 * gathered by the GEMM's LDS-DMA loader (out-of-image taps read a zero line): no im2col matrix.  Stands in
 * for `conv2 -> bn2 -> relu` of clip's Bottleneck (clip/model.py as loaded at lossyless/architectures.py:367-371). */
int lla_conv3x3_relu_f16(const void *in, int n, int H, int W, int pitch, int cin, const void *weights,
                         const float *bias, void *out, int ldc, int cout, void *stream);
/* The same convolution for the tower's NARROW layers -- (cin, cout) = (32, 32), (32, 64), (64, 64); H, W multiples of 8 --
 * as a direct convolution (csrc/conv_direct.hip: one 8 x 8 output tile per wave, the 10 x 10 halo in LDS once, weights
 * resident), bit-identical to lla_conv3x3_relu_f16 on the same operands (`weights` as there, rows of `kpad` halfs; ldc >=
 * cout).  pool != 0 (32 -> 64 only): the 2 x 2 average pool that follows the stem's third convolution is applied in the
 * epilogue and `out` is [n][H/2][W/2][ldc] -- the bytes conv + avgpool write as two kernels.  LLA_EINVAL for any other
 * shape (the caller falls back to the implicit GEMM). */
int lla_conv3x3_direct_relu_f16(const void *in, int n, int H, int W, int pitch, int cin, const void *weights, int kpad,
                                const void *bias, void *out, int ldc, int cout, int pool, void *stream);
/* One WHOLE bottleneck of the RN50 tower's layer1 in one kernel (csrc/bottleneck_fused.hip; clip/model.py Bottleneck as loaded at
 * lossyless/architectures.py:367-371, stride 1, no downsample): out = relu(conv3(relu(conv2(relu(conv1(x))))) + x) for x = NHWC
 * fp16 [n][H][W][pitch] (first cin = 256 channels; or 64, below), H and W multiples of 14; conv1 weights fp16 [64][k1pad] (K = cin), conv2
 * [64][k2pad] (K = 9 * 64 in the order (kh, kw, c)), conv3 [256][k3pad] (K = 64), biases fp32 (BatchNorm folded in); out NHWC
 * fp16 [n][H][W][ldo] (first 256 channels), which must not overlap x.  The 64-channel intermediates are rounded to fp16 as the
 * three-kernel path rounds them, but stay in LDS.  cin = 64 is the stage's FIRST block: no identity, and `w3` [256][k3pad] is the
 * one 1x1 convolution over [conv2's output (64) | x (64)] that conv3 and the downsample convolution fold into (K = 128, bias b3 +
 * bds: lla_rn50_fused_desc(0)).  LLA_EINVAL for any other shape (the caller then runs the three kernels). */
int lla_rn50_bottleneck_f16(const void *x, int n, int H, int W, int pitch, int cin, const void *w1, int k1pad, const void *b1,
                            const void *w2, int k2pad, const void *b2, const void *w3, int k3pad, const void *b3, void *out,
                            int ldo, void *stream);
/* The tower's first convolution (clip/model.py ModifiedResNet.conv1 + bn1 + relu): out[n][H/2][W/2][ldc] (first 32
 * channels) = relu(conv3x3(in, stride 2, pad 1) + bias) for in = NHWC fp16 [n][H][W][3] (pitch 3), H, W multiples of 16;
 * weights fp16 [32][kpad] with K = 27 in the order (kh, kw, c), zero padded; bias fp32 [32].  Direct (csrc/conv_direct.hip),
 * bit-identical to the im2col matrix + lla_gemm_f16_ex it replaces. */
int lla_conv3x3_rgb_s2_relu_f16(const void *in, int n, int H, int W, const void *weights, int kpad, const void *bias,
                                void *out, int ldc, void *stream);
/* Patch embedding alone (conv1 of the tower as a GEMM that gathers 32x32 patches in place, plus the
 * positional embedding): x[b*50 + 1 + t][:] = patch(b, t) . conv_w^T + pos[1 + t] for t < 49; class
 * rows (t = -1) are not written.  images fp16 in `layout`; conv_w fp16 [768][3072] with K ordered
 * (kh,kw,c) for NHWC / (c,kh,kw) for NCHW (LLA_VIT_CONV1_*); pos fp32 [50][768]; x fp32 [B*50][768].
 * Same kernel instantiations lla_vit_b32_forward launches first. */
int lla_patch_embed_f16(const void *images, int layout, int B, const void *conv_w, const float *pos,
                        float *x, void *stream);
/* y16[r][:] = LayerNorm(x32[r*row_stride : +768]) * w + b, eps 1e-5. */
int lla_layernorm768(const float *x, size_t row_stride, const float *w, const float *b,
                     void *y16, int rows, void *stream);
/* qkv fp16 [B*50][2304] -> o fp16 [B*50][768]; 12 heads of 64, softmax(QK^T/8)V. */
int lla_attention50(const void *qkv, void *o, int B, void *stream);

/* ------------------------------------------------------------------------- *
 * Device entry point: CLIP RN50 visual tower
 *   stands in for  clip.load("RN50")[0].visual  as the reference's pretrained featuriser loads it
 *   (lossyless/architectures.py:367-371; clip==1.0 ModifiedResNet + AttentionPool2d, output 1024).
 * The weight blob holds, per convolution in execution order (stem conv1..3; per bottleneck conv1,
 * conv2, conv3 and, in the first block of a stage, the downsample convolution), the BatchNorm-folded
 * weights fp16 [npad][kpad] with K ordered (kh, kw, c) and zero padding, and the folded bias fp32
 * [npad]; then the attention pool's positional embedding fp32 [50][2048], q_proj fp16 [2048][2048] +
 * bias, (k_proj ; v_proj) fp16 [4096][2048] + bias, c_proj fp16 [1024][2048] + bias.
 * lla_rn50_conv_desc(i, out8) -> {cin, cout, ksize, stride, kpad, npad, weight offset, bias offset};
 * lla_rn50_attnpool_offsets(out7) -> byte offsets of {pos, q_w, q_b, kv_w, kv_b, c_w, c_b}.
 * ------------------------------------------------------------------------- */
size_t lla_rn50_weights_bytes(void);
int lla_rn50_conv_count(void);
int lla_rn50_conv_desc(int i, int64_t *out8);
/* First block of stage `stage` (0..3): conv3 and the downsample convolution run as ONE 1x1 convolution over the
 * concatenated inputs [main path | block input] (no identity tensor): out8 = {cin = planes + inplanes, cout = 4 planes,
 * planes, inplanes, kpad, npad, weight offset, bias offset}; weights fp16 [npad][kpad] = [W3 | Wds] (BatchNorm folded,
 * each rounded to fp16 on its own), bias fp32 [npad] = b3 + bds. */
int lla_rn50_fused_desc(int stage, int64_t *out8);
int lla_rn50_attnpool_offsets(int64_t *out7);
size_t lla_rn50_workspace_bytes(int chunk);
/* images [dev] fp16 NHWC [B][224][224][3], CLIP-normalised; z_out [dev] fp16 [B][1024].  With a tower
 * handle (may be NULL: everything on `stream`), from 32 images on the batch is cut in two slices that
 * alternate between the handle's two lanes (see lla_vit_b32_forward_lanes) when the workspace holds two
 * slices (lla_rn50_workspace_bytes returns that size); joined back into `stream` before returning. */
int lla_rn50_forward(const void *images_nhwc_f16, int B, const void *weights, void *workspace,
                     size_t workspace_bytes, int chunk, void *z_out, void *stream, void *tower);

#ifdef __cplusplus
}
#endif
#endif /* LOSSYLESS_AMD_H */
