// CLIP ViT-B/32 visual tower for gfx950 (MI355X): orchestration.  fp16 storage, fp32 accumulate, fp32 residual stream,
// fp32 LayerNorm / softmax statistics.
//
// Stands in for `z = self.clip(X)` at hub/compressor.py:93 (clip==1.0 VisionTransformer.forward; recipe: SURVEY.md 8(a) row
// A10, section 9.3): weight blob layout, slice buffers, the two tower lanes of a tower handle, the layer loop, and the C-ABI
// entry points of include/lossyless_amd.h for the tower and its building blocks.  The kernels live in their own translation
// units (round 6, VERDICT r5 #6: vit.hip split): gemm_pp.hip / gemm_q4.hip / gemm_w8.hip (GEMMs, through gemm_launch.h),
// layernorm_attention.hip (tower_kernels.h).  No A/B switch is read here: switches.h.
#include "gemm_launch.h"
#include "switches.h"
#include "tower_kernels.h"

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>

namespace lla {
namespace {

// ---------------------------------------------------------------------------
// weight blob layout
// ---------------------------------------------------------------------------
constexpr size_t kAlign = 256;
constexpr size_t align_up(size_t v) { return (v + kAlign - 1) & ~(kAlign - 1); }

size_t param_bytes(int id) {
  switch (id) {
    case LLA_VIT_CONV1_NHWC:
    case LLA_VIT_CONV1_NCHW: return (size_t)kWidth * kPatchK * 2;
    case LLA_VIT_CLASS_EMB: return kWidth * 4;
    case LLA_VIT_POS_EMB: return (size_t)kTokens * kWidth * 4;
    case LLA_VIT_LN_PRE_W: case LLA_VIT_LN_PRE_B:
    case LLA_VIT_LN_POST_W: case LLA_VIT_LN_POST_B: return kWidth * 4;
    case LLA_VIT_PROJ_T: return (size_t)kOut * kWidth * 2;
    case LLA_VIT_LN1_W: case LLA_VIT_LN1_B: case LLA_VIT_LN2_W: case LLA_VIT_LN2_B:
    case LLA_VIT_OUT_B: case LLA_VIT_CPROJ_B: return kWidth * 4;
    case LLA_VIT_QKV_W: return (size_t)3 * kWidth * kWidth * 2;
    case LLA_VIT_QKV_B: return 3 * kWidth * 4;
    case LLA_VIT_OUT_W: return (size_t)kWidth * kWidth * 2;
    case LLA_VIT_FC_W: return (size_t)kMlp * kWidth * 2;
    case LLA_VIT_FC_B: return kMlp * 4;
    case LLA_VIT_CPROJ_W: return (size_t)kWidth * kMlp * 2;
    default: return (size_t)-1;
  }
}

size_t globals_bytes() {
  size_t t = 0;
  for (int id = 0; id < LLA_VIT_GLOBAL_COUNT; ++id) t += align_up(param_bytes(id));
  return t;
}
size_t layer_bytes() {
  size_t t = 0;
  for (int id = LLA_VIT_LN1_W; id < LLA_VIT_LAYER_END; ++id) t += align_up(param_bytes(id));
  return t;
}
size_t param_offset(int id, int layer) {
  if (id >= 0 && id < LLA_VIT_GLOBAL_COUNT) {
    size_t t = 0;
    for (int k = 0; k < id; ++k) t += align_up(param_bytes(k));
    return t;
  }
  if (id >= LLA_VIT_LN1_W && id < LLA_VIT_LAYER_END && layer >= 0 && layer < kLayers) {
    size_t t = globals_bytes() + (size_t)layer * layer_bytes();
    for (int k = LLA_VIT_LN1_W; k < id; ++k) t += align_up(param_bytes(k));
    return t;
  }
  return (size_t)-1;
}

struct Workspace {
  float *x;  // [chunk*50][768] fp32 residual stream
  f16 *h;    // [chunk*50][768]  LayerNorm output / attention output
  f16 *big;  // [chunk*50][3072] qkv (2304 wide) or MLP hidden
  f16 *xh;   // [chunk*50][768]  fp16 copy of the residual stream (A operand of the LayerNorm-fused GEMMs)
  float *part;   // [chunk*50][kLnSlots][2] row partial sums
  float *stats;  // [chunk*50][2] (mean, rstd)
};
size_t workspace_bytes(int chunk) {
  const size_t rows = (size_t)chunk * kTokens;
  return align_up(rows * kWidth * 4) + align_up(rows * kWidth * 2) + align_up(rows * kMlp * 2) +
         align_up(rows * kWidth * 2) + align_up(rows * kLnSlots * 2 * 4) + align_up(rows * 2 * 4);
}
// images per tower slice are capped so that chunk * 50 * 768 element offsets fit 32 bits (fp32 epilogues)
constexpr int kMaxChunk = 65536;

}  // namespace

int num_cus() {
  static const int v = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        n = prop.multiProcessorCount;
    }
    return n;
  }();
  return v;
}

// Two tower lanes.  A batch is cut into slices (<= chunk images) and the slices alternate between two
// library-owned HIP streams, each with its own slice buffers: the tail of one lane's persistent GEMM (the
// last, partly filled round of tiles) and its HBM-bound LayerNorm / attention kernels run beside the other
// lane's GEMMs instead of leaving CUs idle.  Images are independent, so the embeddings are bit-identical to
// the one-lane pass (tests/test_gpu_vit.py).  Measured on batch 1024: 93.0k -> 99.5k img/s for the tower
// alone (tools/two_stream_probe.py).  OPT-IN since round 3 (sw::tower_lanes(): tools/ builds, LLA_VIT_STREAMS=2): with two hardware queues active
// the tower is not bit-reproducible on this stack -- between one embedding per 10^6 and one per 10^8 images (box and build dependent) comes out a few fp16 ulps
// (<= 3e-3) different from run to run, i.e. a 1 M-image file differs from its own re-run (docs/history/DESIGN_rounds_1-5.md 5.3; found
// by the 1 M-image sharding test) -- while one stream gave 0 differing embeddings in 15 M images.  Bit-exact
// records are this path's contract, so the default is ONE stream (-4 % img/s); profiled passes always use one.
int tower_lanes() { return sw::tower_lanes(); }
// A tower handle (lla_tower_create) owns the two lane streams of ONE device and their events; nothing about
// the lanes lives in the library itself.
int lanes_create(Lanes **out) {
  Lanes *l = new Lanes();
  hipError_t e = hipGetDevice(&l->device);
  for (int i = 0; i < 2 && e == hipSuccess; ++i) {
    e = hipStreamCreateWithFlags(&l->st[i], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&l->join[i], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&l->fork, hipEventDisableTiming);
  if (e != hipSuccess) { lanes_destroy(l); return hip_fail(e); }
  *out = l;
  return LLA_OK;
}

void lanes_destroy(Lanes *l) {
  if (!l) return;
  for (int i = 0; i < 2; ++i) {
    if (l->st[i]) { (void)hipStreamSynchronize(l->st[i]); (void)hipStreamDestroy(l->st[i]); }
    if (l->join[i]) (void)hipEventDestroy(l->join[i]);
  }
  if (l->fork) (void)hipEventDestroy(l->fork);
  delete l;
}

unsigned dynamic_lds_limit(const void *kernel) {
  static std::mutex mu;
  static std::map<std::pair<int, const void *>, unsigned> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 64u * 1024u;
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_pair(dev, kernel);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int v = 64 * 1024;
  (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
  const unsigned cap = (unsigned)v > 160u * 1024u ? 160u * 1024u : (unsigned)v;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);
  cache.emplace(key, cap);
  return cap;
}

int lanes_fork(Lanes *ln, hipStream_t caller) {
  hipError_t e = hipEventRecord(ln->fork, caller);
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipStreamWaitEvent(ln->st[i], ln->fork, 0);
  return e == hipSuccess ? LLA_OK : hip_fail(e);
}

int lanes_join(Lanes *ln, hipStream_t caller) {
  for (int i = 0; i < 2; ++i) {
    hipError_t e = hipEventRecord(ln->join[i], ln->st[i]);
    if (e == hipSuccess) e = hipStreamWaitEvent(caller, ln->join[i], 0);
    if (e != hipSuccess) return hip_fail(e);
  }
  ln->dirty = false;
  return LLA_OK;
}

}  // namespace lla

using namespace lla;

extern "C" {

size_t lla_vit_b32_weights_bytes(void) { return globals_bytes() + (size_t)kLayers * layer_bytes(); }
size_t lla_vit_b32_param_offset(int param, int layer) { return param_offset(param, layer); }
size_t lla_vit_b32_param_bytes(int param) { return param_bytes(param); }
size_t lla_vit_b32_workspace_bytes(int chunk) {
  if (chunk <= 0) chunk = sw::default_chunk();
  return (size_t)tower_lanes() * workspace_bytes(chunk > kMaxChunk ? kMaxChunk : chunk);   // one slice buffer per lane
}

int lla_patch_embed_f16(const void *images, int layout, int B, const void *conv_w, const float *pos,
                        float *x, void *stream) {
  if (B < 0 || (layout != LLA_LAYOUT_NHWC && layout != LLA_LAYOUT_NCHW)) return LLA_EINVAL;
  if (B == 0) return LLA_OK;
  if (!images || !conv_w || !pos || !x) return LLA_EINVAL;
  GemmParams pe{};
  pe.A = reinterpret_cast<const f16 *>(images);
  pe.W = reinterpret_cast<const f16 *>(conv_w);
  pe.C = x;
  pe.pos = pos;
  pe.M = B * kPatches; pe.N = kWidth; pe.K = kPatchK; pe.lda = 0; pe.ldc = kWidth;
  hipStream_t st = as_stream(stream);
  if (layout == LLA_LAYOUT_NHWC) return launch_gemm_rt(EPI_PATCH, A_PATCH_NHWC, pe, st);
  return launch_gemm_rt(EPI_PATCH, A_PATCH_NCHW, pe, st);
}

int lla_gemm_f16(const void *A, const void *W, const float *bias, void *C, int M, int N, int K,
                 int epilogue, void *stream) {
  GemmParams p{};
  p.A = reinterpret_cast<const f16 *>(A);
  p.W = reinterpret_cast<const f16 *>(W);
  p.bias = bias;
  p.C = C;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.ldc = N;
  hipStream_t st = as_stream(stream);
  switch (epilogue) {
    case LLA_EPI_F16: return launch_gemm_rt(EPI_F16, A_PLAIN, p, st);
    case LLA_EPI_QUICKGELU_F16: return launch_gemm_rt(EPI_QGELU, A_PLAIN, p, st);
    case LLA_EPI_RESID_F32: return launch_gemm_rt(EPI_RESID, A_PLAIN, p, st);
    default: return LLA_EINVAL;
  }
}

int lla_gemm_f16_ex(const void *A, int lda, const void *W, const float *bias, void *C, int ldc,
                    const void *resid, int ldr, int M, int N, int K, int epilogue, void *stream) {
  GemmParams p{};
  p.A = reinterpret_cast<const f16 *>(A);
  p.W = reinterpret_cast<const f16 *>(W);
  p.bias = bias;
  p.C = C;
  p.resid = resid;
  p.ldr = ldr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc;
  if (lda < K || (lda & 7) || (ldc & 3)) return LLA_EINVAL;
  if (ldc < N) {   // narrow output: only the first ldc columns (a multiple of 32) are stored
    if ((ldc & 31) || ldc <= 0 || (epilogue != LLA_EPI_RELU_F16 && epilogue != LLA_EPI_ADD_RELU_F16))
      return LLA_EINVAL;   // (the ReLU kinds run on the kernels whose epilogue knows about n_store)
    p.n_store = ldc;
  }
  hipStream_t st = as_stream(stream);
  switch (epilogue) {
    case LLA_EPI_F16: return launch_gemm_rt(EPI_F16, A_PLAIN, p, st);
    case LLA_EPI_QUICKGELU_F16: return launch_gemm_rt(EPI_QGELU, A_PLAIN, p, st);
    case LLA_EPI_RESID_F32: return launch_gemm_rt(EPI_RESID, A_PLAIN, p, st);
    case LLA_EPI_RELU_F16: return launch_gemm_rt(EPI_RELU, A_PLAIN, p, st);
    case LLA_EPI_ADD_RELU_F16:
      if (!resid || ldr < (ldc < N ? ldc : N) || (ldr & 3)) return LLA_EINVAL;
      return launch_gemm_rt(EPI_ADDRELU, A_PLAIN, p, st);
    default: return LLA_EINVAL;
  }
}

int lla_conv3x3_relu_f16(const void *in, int n, int H, int W, int pitch, int cin, const void *weights,
                         const float *bias, void *out, int ldc, int cout, void *stream) {
  if (n < 0 || H <= 0 || W <= 0 || cin <= 0 || ((cin % BK) && cin != 32) || pitch < cin || (pitch & 7) ||
      cout <= 0 || (cout % BN2) || ldc <= 0 || (ldc & 3) || (ldc < cout && (ldc & 31)))
    return LLA_EINVAL;
  if (n == 0) return LLA_OK;
  if (!in || !weights || !out) return LLA_EINVAL;
  if ((size_t)n * H * W >= (1ull << 31)) return LLA_EINVAL;
  GemmParams p{};
  p.A = reinterpret_cast<const f16 *>(in);
  p.W = reinterpret_cast<const f16 *>(weights);
  p.bias = bias;
  p.C = out;
  p.M = n * H * W; p.N = cout; p.K = (9 * cin + BK - 1) / BK * BK; p.lda = pitch; p.ldc = ldc;
  p.conv_h = H; p.conv_w = W; p.conv_cin = cin;
  if (ldc < cout) p.n_store = ldc;
  return launch_gemm_rt(EPI_RELU, A_CONV3, p, as_stream(stream));
}

int lla_profiler_create(void **profiler, int max_launches) {
  if (!profiler || max_launches <= 0) return LLA_EINVAL;
  Profiler *p = new Profiler();
  p->pool.resize((size_t)max_launches);
  for (auto &r : p->pool) {
    hipError_t e = hipEventCreate(&r.a);
    if (e == hipSuccess) e = hipEventCreate(&r.b);
    if (e != hipSuccess) { delete p; return hip_fail(e); }
  }
  *profiler = p;
  return LLA_OK;
}

int lla_profiler_destroy(void *profiler) {
  Profiler *p = reinterpret_cast<Profiler *>(profiler);
  if (!p) return LLA_EINVAL;
  for (auto &r : p->pool) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  delete p;
  return LLA_OK;
}

int lla_profiler_collect(void *profiler, double *ms, double *work, long long *launches) {
  Profiler *p = reinterpret_cast<Profiler *>(profiler);
  if (!p || !ms || !work || !launches) return LLA_EINVAL;
  for (size_t i = 0; i < p->used; ++i) {
    auto &r = p->pool[i];
    hipError_t e = hipEventSynchronize(r.b);
    float t = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, r.a, r.b);
    if (e != hipSuccess) return hip_fail(e);
    ms[r.cls] += t; work[r.cls] += r.work; launches[r.cls] += 1;
  }
  p->used = 0;
  return LLA_OK;
}

static int vit_forward_impl(const void *images, int layout, int B, const void *weights, void *workspace,
                            size_t ws_bytes, int chunk, void *z_out, void *stream, void *profiler,
                            Lanes *tower, bool deferred, const void *const *pieces = nullptr, int n_pieces = 0,
                            int piece_images = 0);

int lla_tower_create(void **tower) {
  if (!tower) return LLA_EINVAL;
  Lanes *l = nullptr;
  const int rc = lanes_create(&l);
  if (rc == LLA_OK) *tower = l;
  return rc;
}

int lla_tower_destroy(void *tower) {
  if (!tower) return LLA_EINVAL;
  lanes_destroy(reinterpret_cast<Lanes *>(tower));
  return LLA_OK;
}

// Epoch of an EPI_RESID_LNX launch: what its workgroups tag their exchange words with.  Unique per launch within the
// process and never 0 (a counter, not state any result depends on: the words are compared for equality only).
static unsigned next_lnx_epoch() {
  static std::atomic<unsigned> counter{0x5EED0000u};
  unsigned e = counter.fetch_add(1u, std::memory_order_relaxed) + 1u;
  return e ? e : counter.fetch_add(1u, std::memory_order_relaxed) + 1u;
}

int lla_tower_set_option(void *tower, int option, int value) {
  Lanes *l = reinterpret_cast<Lanes *>(tower);
  if (!l) return LLA_EINVAL;
  switch (option) {
    case LLA_TOWER_OPT_LNX: l->lnx = value != 0; return LLA_OK;
    case LLA_TOWER_OPT_LNX_WAIT: l->lnx_wait = value; return LLA_OK;
    default: return LLA_EINVAL;
  }
}

int lla_tower_join(void *tower, void *stream) {
  if (!tower) return LLA_EINVAL;
  return lanes_join(reinterpret_cast<Lanes *>(tower), as_stream(stream));
}

int lla_vit_b32_forward(const void *images, int layout, int B, const void *weights,
                        void *workspace, size_t ws_bytes, int chunk, void *z_out, void *stream) {
  return vit_forward_impl(images, layout, B, weights, workspace, ws_bytes, chunk, z_out, stream, nullptr, nullptr,
                          false);
}

int lla_vit_b32_forward_profiled(const void *images, int layout, int B, const void *weights,
                                 void *workspace, size_t ws_bytes, int chunk, void *z_out,
                                 void *stream, void *profiler) {
  return vit_forward_impl(images, layout, B, weights, workspace, ws_bytes, chunk, z_out, stream, profiler, nullptr,
                          false);
}

int lla_vit_b32_forward_lanes(void *tower, const void *images, int layout, int B, const void *weights,
                              void *workspace, size_t ws_bytes, int chunk, void *z_out, void *stream,
                              int deferred) {
  if (!tower) return LLA_EINVAL;
  return vit_forward_impl(images, layout, B, weights, workspace, ws_bytes, chunk, z_out, stream, nullptr,
                          reinterpret_cast<Lanes *>(tower), deferred != 0);
}

int lla_vit_b32_forward_gather(void *tower, const void *const *pieces, int n_pieces, int piece_images, int layout, int B,
                               const void *weights, void *workspace, size_t ws_bytes, void *z_out, void *stream) {
  if (!tower || !pieces) return LLA_EINVAL;
  return vit_forward_impl(nullptr, layout, B, weights, workspace, ws_bytes, 0, z_out, stream, nullptr,
                          reinterpret_cast<Lanes *>(tower), false, pieces, n_pieces, piece_images);
}

static int vit_forward_impl(const void *images, int layout, int B, const void *weights, void *workspace,
                            size_t ws_bytes, int chunk, void *z_out, void *stream, void *profiler,
                            Lanes *tower, bool deferred, const void *const *pieces, int n_pieces, int piece_images) {
  Profiler *prof = reinterpret_cast<Profiler *>(profiler);
  if (pieces) {
    // the batch in pieces of piece_images images (the last one may be shorter): one slice, whole 256-row tiles
    if (n_pieces < 1 || n_pieces > 64 || piece_images <= 0 || (piece_images & 255) || B <= (n_pieces - 1) * piece_images ||
        B > n_pieces * piece_images || (B & 127) || B < 256)
      return LLA_EINVAL;
    for (int i = 0; i < n_pieces; ++i)
      if (!pieces[i]) return LLA_EINVAL;
    images = pieces[0];
    if (B > (chunk > 0 ? chunk : sw::default_chunk())) return LLA_EINVAL;
  }
  if (!images || !weights || !workspace || !z_out || B < 0) return LLA_EINVAL;
  if (layout != LLA_LAYOUT_NHWC && layout != LLA_LAYOUT_NCHW) return LLA_EINVAL;
  if (B == 0) return LLA_OK;
  if (chunk <= 0) chunk = sw::default_chunk();
  if (chunk > kMaxChunk) chunk = kMaxChunk;
  if (chunk > B) chunk = B;
  hipStream_t st_caller = as_stream(stream);
  // two lanes when the batch is large enough, the caller's buffer holds two slices and nobody is timing launches
  // Lane i's slice buffers live in the i-th half of the caller's workspace (fixed offsets: a deferred pass
  // may still be running on the other lane when the next call arrives with a different slice size).
  const size_t lane_bytes = (ws_bytes / 2) & ~(size_t)255;
  int lanes = 1;
  if (tower) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != tower->device) return LLA_EINVAL;   // a tower belongs to one device
  }
  if (!prof && tower && tower_lanes() == 2) {
    if (deferred) {
      // whole slices alternate between the lanes ACROSS calls; nothing is joined until lla_tower_join
      if (workspace_bytes(chunk) <= lane_bytes) lanes = 2;
    } else if (B >= sw::lane_split_min()) {
      const int half = (B + 1) / 2;
      const int sub = chunk < half ? chunk : half;
      if (workspace_bytes(sub) <= lane_bytes) { lanes = 2; chunk = sub; }
    }
  }
  if (ws_bytes < workspace_bytes(chunk)) return LLA_ECAP;
  Lanes *ln = nullptr;
  int slice = 0;
  if (lanes == 1 && tower && tower->dirty) {
    // a pass on the caller's stream uses lane 0's slice buffers: deferred passes still in flight must finish first
    const int jrc = lanes_join(tower, st_caller);
    if (jrc != LLA_OK) return jrc;
  }
  if (lanes == 2) {
    ln = tower;
    const int frc = lanes_fork(ln, st_caller);
    if (frc != LLA_OK) return frc;
    if (deferred) slice = ln->next;
  }

  const uint8_t *wb = reinterpret_cast<const uint8_t *>(weights);
  auto P16 = [&](int id, int l) { return reinterpret_cast<const f16 *>(wb + param_offset(id, l)); };
  auto P32 = [&](int id, int l) { return reinterpret_cast<const float *>(wb + param_offset(id, l)); };

  Workspace wss[2];
  const size_t rows_cap = (size_t)chunk * kTokens;
  for (int i = 0; i < lanes; ++i) {
    uint8_t *w8 = reinterpret_cast<uint8_t *>(workspace) + (size_t)i * lane_bytes;
    wss[i].x = reinterpret_cast<float *>(w8);
    w8 += align_up(rows_cap * kWidth * 4);
    wss[i].h = reinterpret_cast<f16 *>(w8);
    w8 += align_up(rows_cap * kWidth * 2);
    wss[i].big = reinterpret_cast<f16 *>(w8);
    w8 += align_up(rows_cap * kMlp * 2);
    wss[i].xh = reinterpret_cast<f16 *>(w8);
    w8 += align_up(rows_cap * kWidth * 2);
    wss[i].part = reinterpret_cast<float *>(w8);
    w8 += align_up(rows_cap * kLnSlots * 2 * 4);
    wss[i].stats = reinterpret_cast<float *>(w8);
  }

  int rc = LLA_OK;
#define LLA_TRY(expr) do { rc = (expr); if (rc != LLA_OK) return rc; } while (0)

  // Slices of `chunk` images; a ragged last slice of >= 256 images is cut once more so that its main part is a multiple of
  // 128 images = a whole number of 256-row tiles (50 x 128 = 25 x 256): that part runs on the four-wave GEMM, the
  // < 128 images left over on the small-M kernels.  Images are independent: same embeddings for every cut.
  for (int c0 = 0, bc = 0; c0 < B; c0 += bc, ++slice) {
    bc = (B - c0) < chunk ? (B - c0) : chunk;
    if (bc >= 256 && (bc & 127)) bc -= bc & 127;
    const int M = bc * kTokens;
    const Workspace &ws = wss[lanes == 2 ? (slice & 1) : 0];
    hipStream_t st = lanes == 2 ? ln->st[slice & 1] : st_caller;

    // patch embedding: conv1 as a GEMM that reads patches in place, + pos, into token rows
    GemmParams pe{};
    pe.A = reinterpret_cast<const f16 *>(images) + (size_t)c0 * kImgElems;
    pe.W = P16(layout == LLA_LAYOUT_NHWC ? LLA_VIT_CONV1_NHWC : LLA_VIT_CONV1_NCHW, 0);
    pe.bias = nullptr;
    pe.C = ws.x;
    pe.pos = P32(LLA_VIT_POS_EMB, 0);
    pe.M = bc * kPatches; pe.N = kWidth; pe.K = kPatchK; pe.lda = 0; pe.ldc = kWidth;
    if (pieces) {
      if (lanes != 1 || c0 != 0 || bc != B) return LLA_EINVAL;   // (one slice on the caller's stream)
      pe.a_chunk_images = piece_images;
      for (int i = 0; i < n_pieces; ++i) pe.a_chunk[i] = reinterpret_cast<const f16 *>(pieces[i]);
    }
    LLA_TRY(launch_gemm_rt(EPI_PATCH, layout == LLA_LAYOUT_NHWC ? A_PATCH_NHWC : A_PATCH_NCHW, pe, st, prof));
    // token assembly + ln_pre (fp32, in place) + ln_1 of block 0 (fp16 out); patch rows already hold conv + pos
    LLA_TRY(ln_pre_ln1_impl(ws.x, P32(LLA_VIT_CLASS_EMB, 0), P32(LLA_VIT_POS_EMB, 0), P32(LLA_VIT_LN_PRE_W, 0),
                            P32(LLA_VIT_LN_PRE_B, 0), P32(LLA_VIT_LN1_W, 0), P32(LLA_VIT_LN1_B, 0), ws.h, M, st, prof));

    int dir = 0;                       // direction of the kernel being launched (0: first rows first)
    const int zig = sw::zigzag();
    // LayerNorm in the residual GEMMs' epilogues (EPI_RESID_LNX, gemm_q4.hip) for slices of whole 256-row tiles that the
    // four-wave kernel takes: ln_2 of every block in out-proj's epilogue, ln_1 of the next block in c_proj's; behind each
    // such GEMM lnx_cleanup_kernel redoes the row tiles whose column tiles missed each other.  Output in ws.xh (the
    // consumers' A operand then).  The exchange words of the slice's launches are zeroed here, once.
    // (... and inside the four-wave kernel's 32-bit panel offsets for BOTH residual GEMMs -- c_proj's A operand has the
    // longest rows, lda = 3072: slices of 14 080+ images, ADVICE r5 -- so that launch_q4(EPI_RESID_LNX) cannot answer
    // LLA_EINVAL here; such slices take the LayerNorm kernels, with the residual GEMMs on the ping-pong kernel)
    const bool lnx = (M & 255) == 0 && M >= 9000 && (size_t)M * kMlp * 2 < (1ull << 32) && (!tower || tower->lnx);
    const int tiles_m = M / 256;
    float *const lnx_part = ws.part;                                                       // [tiles_m][3][256] granules of 16 bytes
    unsigned *const lnx_words = reinterpret_cast<unsigned *>(ws.part + (size_t)tiles_m * 3 * 256 * 4);   // [launch]{flag [tiles_m][3], done [tiles_m][3]}
    int lnx_launch = 0;
    if (lnx && hipMemsetAsync(lnx_words, 0, (size_t)(2 * kLayers - 1) * tiles_m * 6 * sizeof(unsigned), st) != hipSuccess)
      return hip_fail(hipGetLastError());
    auto lnx_gemm = [&](GemmParams g, const float *gamma, const float *beta, int &d) -> int {
      g.lnx_g = gamma; g.lnx_b = beta; g.lnx_h = ws.xh; g.lnx_part = lnx_part;
      g.lnx_flag = lnx_words + (size_t)lnx_launch * tiles_m * 6;
      g.lnx_done = g.lnx_flag + (size_t)tiles_m * 3;
      g.lnx_wait = tower ? tower->lnx_wait : kLnxWaitDefault;
      g.lnx_epoch = next_lnx_epoch();
      ++lnx_launch;
      d ^= zig; g.rev = d;
      {
        ProfScope scope(prof, st, LLA_PROF_GEMM, 2.0 * g.M * g.N * g.K);
        const int rc2 = launch_q4(EPI_RESID_LNX, g, st);
        if (rc2 != LLA_OK) return rc2;
      }
#if LLA_LNX_SYNC
      if (hipStreamSynchronize(st) != hipSuccess) return hip_fail(hipGetLastError());   // (A/B only: common.h)
#endif
      d ^= zig;
      return lnx_cleanup_impl(ws.x, g.lnx_done, gamma, beta, ws.xh, tiles_m, d, g.lnx_epoch, st, prof);
    };
    bool ln1_by_gemm = false;          // ln_1 of this block was written to ws.xh by the c_proj GEMM of the block before
    for (int l = 0; l < kLayers; ++l) {
      if (l > 0 && !ln1_by_gemm) {
        dir ^= zig;
        LLA_TRY(layernorm_impl(ws.x, kWidth, P32(LLA_VIT_LN1_W, l), P32(LLA_VIT_LN1_B, l), ws.h, M, st, prof, dir));
      }
      GemmParams g{};
      g.M = M;
      // Only the class token leaves the tower (ln_post(x[:, 0]) @ proj), so after the last
      // block's attention every remaining per-row op runs on the B class rows alone: row
      // stride 50*768 selects them in place, results are bit-identical to the full pass.
      const bool cls_only = (l == kLayers - 1) && sw::prune_last_block();
      // qkv = h @ in_proj^T + b
      g.A = ln1_by_gemm ? ws.xh : ws.h; g.W = P16(LLA_VIT_QKV_W, l); g.bias = P32(LLA_VIT_QKV_B, l); g.C = ws.big;
      g.N = 3 * kWidth; g.K = kWidth; g.lda = kWidth; g.ldc = 3 * kWidth;
      if (cls_only) {
        // ... and of the last block's queries only the class token's is used: K and V for every token
        // (columns 768 .. 2303), Q for the B class rows.  The other query rows keep stale (finite) bytes;
        // attention rows are independent, and only row 0 of every image is read afterwards.
        GemmParams kv = g;
        kv.W = g.W + (size_t)kWidth * kWidth; kv.bias = g.bias + kWidth;
        kv.C = ws.big + kWidth; kv.N = 2 * kWidth;
        GemmParams q = g;
        q.M = bc; q.N = kWidth; q.lda = kTokens * kWidth; q.ldc = kTokens * 3 * kWidth;
        LLA_TRY(launch_gemm_rt(EPI_F16, A_PLAIN, kv, st, prof));
        LLA_TRY(launch_gemm_rt(EPI_F16, A_PLAIN, q, st, prof));
      } else {
        dir ^= zig; g.rev = dir;
        LLA_TRY(launch_gemm_rt(EPI_F16, A_PLAIN, g, st, prof));
        g.rev = 0;
      }
      // o = softmax(q k^T / 8) v   (h is dead, reuse it)
      dir ^= zig;
      LLA_TRY(attention_impl(ws.big, ws.h, bc, st, prof, dir));
      const int rows = cls_only ? bc : M;
      const int xs = cls_only ? kTokens * kWidth : kWidth;  // row stride of x / o for this pass
      // x += o @ out_proj^T + b
      g.M = rows;
      g.A = ws.h; g.W = P16(LLA_VIT_OUT_W, l); g.bias = P32(LLA_VIT_OUT_B, l); g.C = ws.x;
      g.N = kWidth; g.K = kWidth; g.lda = xs; g.ldc = xs;
      const bool ln2_by_gemm = lnx && !cls_only;
      if (ln2_by_gemm) {
        LLA_TRY(lnx_gemm(g, P32(LLA_VIT_LN2_W, l), P32(LLA_VIT_LN2_B, l), dir));
      } else {
        dir ^= zig; g.rev = dir;
        LLA_TRY(launch_gemm_rt(EPI_RESID, A_PLAIN, g, st, prof));
        g.rev = 0;
        dir ^= zig;
        LLA_TRY(layernorm_impl(ws.x, (size_t)xs, P32(LLA_VIT_LN2_W, l), P32(LLA_VIT_LN2_B, l), ws.h, rows, st, prof, dir));
      }
      // g = quickgelu(h @ c_fc^T + b)
      g.A = ln2_by_gemm ? ws.xh : ws.h; g.W = P16(LLA_VIT_FC_W, l); g.bias = P32(LLA_VIT_FC_B, l); g.C = ws.big;
      g.N = kMlp; g.K = kWidth; g.lda = kWidth; g.ldc = kMlp;
      dir ^= zig; g.rev = dir;
      LLA_TRY(launch_gemm_rt(EPI_QGELU, A_PLAIN, g, st, prof));
      g.rev = 0;
      // x += g @ c_proj^T + b
      g.A = ws.big; g.W = P16(LLA_VIT_CPROJ_W, l); g.bias = P32(LLA_VIT_CPROJ_B, l); g.C = ws.x;
      g.N = kWidth; g.K = kMlp; g.lda = kMlp; g.ldc = xs;
      ln1_by_gemm = lnx && !cls_only && l + 1 < kLayers;      // ln_1 of the next block rides in this GEMM's epilogue
      if (ln1_by_gemm) {
        LLA_TRY(lnx_gemm(g, P32(LLA_VIT_LN1_W, l + 1), P32(LLA_VIT_LN1_B, l + 1), dir));
      } else {
        dir ^= zig; g.rev = dir;
        LLA_TRY(launch_gemm_rt(EPI_RESID, A_PLAIN, g, st, prof));
        g.rev = 0;
      }
    }

    // ln_post on class tokens only, then @ proj
    LLA_TRY(layernorm_impl(ws.x, (size_t)kTokens * kWidth, P32(LLA_VIT_LN_POST_W, 0),
                           P32(LLA_VIT_LN_POST_B, 0), ws.h, bc, st, prof));
    GemmParams g{};
    g.A = ws.h; g.W = P16(LLA_VIT_PROJ_T, 0); g.bias = nullptr;
    g.C = reinterpret_cast<f16 *>(z_out) + (size_t)c0 * kOut;
    g.M = bc; g.N = kOut; g.K = kWidth; g.lda = kWidth; g.ldc = kOut;
    LLA_TRY(launch_gemm_rt(EPI_F16, A_PLAIN, g, st, prof));
  }
#undef LLA_TRY
  if (lanes == 2 && deferred) { ln->next = slice & 1; ln->dirty = true; return LLA_OK; }
  if (lanes == 2) return lanes_join(ln, st_caller);   // the caller's stream continues when both lanes are done
  return LLA_OK;
}

}  // extern "C"

