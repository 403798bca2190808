// One NeSVoR training iteration behind ONE entry point.
//
// Replaces the loop body of the reference's train() (nesvor/nesvor/train.py:179-197: NeSVoR.forward
// models.py:260-327, loss.backward(), optimizer.step(), zero_grad()) as a fixed sequence of the native launches of
// this library.  nesvor_amd/direct.py issues the same sequence from Python - ~27 ctypes calls and ~20 tensor
// allocations per iteration, 0.42 ms of host time - which is what bounds the iteration once a GPU's share of the batch
// is small (BASELINE C2: 2^18 points; C3 read literally: 2^17 points per GPU on 8 GPUs need 0.2 ms of GPU time).
// Here the host enqueues everything in one call, into workspace buffers the caller allocated once:
//
//   side stream : pose regulariser (serial chain per slice)           ...            owner pass of the hash-grid backward
//   main stream : prologue | sampler | hash grid | MLPs | loss | MLP backwards | aggregation pass | sampler backward
//                 | per-slice gradients | epilogue | [AdamW]
// (overlap_owner bit 1: the table's AdamW step is taken by the owner pass itself, chunk by chunk while a chunk's gradient is
// in LDS - nesvor_hashgrid_backward_adamw - and the closing AdamW launch covers the small parameters only.)
//
// The configuration switches are those of the reference's args (no_transformation_optimization, no_pixel_variance,
// no_slice_scale, no_slice_variance, n_levels_bias).  Data-parallel runs call the step in two phases so that the host
// can start the all-reduce of the fine levels' gradient in between (nesvor_amd/ddp.py).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <new>
#include "../../include/nesvor_hip.h"

namespace {

struct StepCtx {
  nesvor_step_t d;
  hipEvent_t ev_fork, ev_pose, ev_agg, ev_owner, ev_bwd, ev_sums, ev_norms, ev_sg0, ev_sg1, ev_owner_early;
  // Pipelined table update (round 6): the owner pass of the previous run went out as two launches - levels [0, pipe_level), event
  // ev_owner_early, then the rest, event ev_owner - and this run's hash-grid forward follows it level range by level range
  int pending_pipe_level = 0;
  hipEvent_t tu0[2] = {nullptr, nullptr};  // timing: start of a run's owner pass (double-buffered: read by the NEXT run's union span)
  unsigned run_no = 0;
  // Split operand images of the three networks' weights (nesvor_mlp_t.weight_images), rebuilt once per iteration by the launch that
  // takes the weight norms and copied - not rebuilt - by the workgroups of the four MLP launches.  One allocation, owned here.
  void* wimg = nullptr;
  size_t wimg_stride = 0;
  bool head_on_side = false;  // (NESVOR_STEP_HEAD=side: phase 2 of a split run must join what phase 1 forked)
  bool sums_on_side = false;  // the networks' parameter-gradient sums of the current iteration were left on the side stream
  bool pending_join = false;  // a table update of the previous run is still on the side stream (NESVOR_STEP_DEFER_JOIN)
  // nesvor_step_timing: HIP-event brackets around the PRODUCT launches of a run, each pair on the stream its launch goes to
  bool timing = false, timed_run = false;
  hipEvent_t t0[NESVOR_STEP_TIMED_SPANS], t1[NESVOR_STEP_TIMED_SPANS];
  bool span_used[NESVOR_STEP_TIMED_SPANS];
};
struct Span {  // records t0 now and t1 when it goes out of scope (timing on), on the stream of the launch it brackets
  StepCtx* c; int k; hipStream_t st;
  Span(StepCtx* c_, int k_, hipStream_t st_) : c(c_->timing ? c_ : nullptr), k(k_), st(st_) {
    if (c != nullptr) { (void)hipEventRecord(c->t0[k], st); c->span_used[k] = true; }
  }
  ~Span() { if (c != nullptr) (void)hipEventRecord(c->t1[k], st); }
};

// out[0] = mean(x[0..n)) in two launches (deterministic order): partial sums of 256 workgroups, then one wave
__global__ __launch_bounds__(256) void mean_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ partial) {
  __shared__ float red[256];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void mean_final_kernel(const float* __restrict__ partial, int n_partial, float scale, float* __restrict__ out) {
  __shared__ float red[256];
  red[threadIdx.x] = (int)threadIdx.x < n_partial ? partial[threadIdx.x] : 0.f;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}

__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 a = *reinterpret_cast<float4*>(dst + i);
    const float4 b = *reinterpret_cast<const float4*>(src + i);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    *reinterpret_cast<float4*>(dst + i) = a;
  } else {
    for (int64_t k = i; k < n; ++k) dst[k] += src[k];
  }
}

// biasReg = (mean log_bias)^2 (models.py:322-323) into losses[5]
__global__ void square_kernel(const float* __restrict__ m, float* __restrict__ out) { out[0] = m[0] * m[0]; }

#define NESVOR_TRY(expr)            \
  do {                              \
    const int e_ = (int)(expr);     \
    if (e_ != 0) return e_;         \
  } while (0)

int mlp_backward_into(const nesvor_mlp_t& net, int group_sums, const float* xa, const float* xb, const float* dy,
                      float* const* saved, float* dxa, float* dxb, float* partial, float* grad_segment, int n_params,
                      int64_t N, hipStream_t st, float* const* dpre_scratch, float* dxb_absmax = nullptr) {
  nesvor_mlp_t d = net;
  d.dxa_group_sums = group_sums;
  // fused dX + dW + db kernel: no dpre scratch; shapes it refuses (nesvor_mlp_backward_fused_ok) run as a dX launch + a dW launch
  // through the step's shared scratch
  float* no_scratch[NESVOR_MAX_MLP_LAYERS] = {nullptr, nullptr, nullptr, nullptr};
  if (!nesvor_mlp_backward_fused_ok(&d, N)) {
    for (int l = 0; l < d.n_hidden; ++l) {
      if (dpre_scratch == nullptr || dpre_scratch[l] == nullptr) return (int)hipErrorInvalidValue;
      no_scratch[l] = dpre_scratch[l];
    }
  }
  // per-workgroup partial sums (columns W0,b0,W1,b1,...); the caller sums them into the network's segment of the flat gradient
  // (ONE launch for all networks of the step, after the last backward: nesvor_sum_rows_multi)
  (void)grad_segment; (void)n_params;
  return nesvor_mlp_backward_bounded(&d, xa, xb, dy, saved, no_scratch, dxa, dxb, partial, NESVOR_STEP_MLP_PARTIALS, N, dxb_absmax, st);
}

}  // namespace

extern "C" void* nesvor_step_create(const nesvor_step_t* desc) {
  if (desc == nullptr) return nullptr;
  StepCtx* c = new (std::nothrow) StepCtx;
  if (c == nullptr) return nullptr;
  c->d = *desc;
  hipEvent_t* evs[10] = {&c->ev_fork, &c->ev_pose, &c->ev_agg, &c->ev_owner, &c->ev_bwd, &c->ev_sums, &c->ev_norms, &c->ev_sg0, &c->ev_sg1,
                         &c->ev_owner_early};
  for (hipEvent_t* e : evs) {
    if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) { delete c; return nullptr; }
  }
  for (int k = 0; k < NESVOR_STEP_TIMED_SPANS; ++k) { c->t0[k] = nullptr; c->t1[k] = nullptr; c->span_used[k] = false; }
  {
    // NESVOR_STEP_WEIGHT_IMAGES=0: every MLP launch builds its own images (rounds 2-5; A/B switch)
    static const bool on = []() { const char* e = getenv("NESVOR_STEP_WEIGHT_IMAGES"); return e == nullptr || atoi(e) != 0; }();
    int64_t need = nesvor_mlp_weight_images_bytes(&desc->density);
    if (desc->has_lv && nesvor_mlp_weight_images_bytes(&desc->sigma) > need) need = nesvor_mlp_weight_images_bytes(&desc->sigma);
    if (desc->has_b && nesvor_mlp_weight_images_bytes(&desc->bias_net) > need) need = nesvor_mlp_weight_images_bytes(&desc->bias_net);
    c->wimg_stride = ((size_t)need + 255) / 256 * 256;
    if (on && need > 0 && hipMalloc(&c->wimg, 3 * c->wimg_stride) != hipSuccess) { c->wimg = nullptr; (void)hipGetLastError(); }  // (without it: the in-kernel builds)
  }
  return c;
}

// Per-launch HIP-event timing of the product step (bench.py's roofline leg): on = 1 creates the event pairs and makes every
// following nesvor_step_run bracket its launches - on the stream each goes to, so the owner pass (+ the table's AdamW) is timed
// where it runs, on the side stream; nesvor_step_timing_read waits for the last run and returns the NESVOR_STEP_TIMED_SPANS
// durations in ms (-1: the span did not occur).  An event bracket reads a few microseconds more than the kernel it brackets.
extern "C" int nesvor_step_timing(void* handle, int on) {
  if (handle == nullptr) return (int)hipErrorInvalidValue;
  StepCtx* c = static_cast<StepCtx*>(handle);
  if (on && c->t0[0] == nullptr) {
    for (int k = 0; k < NESVOR_STEP_TIMED_SPANS; ++k)
      if (hipEventCreate(&c->t0[k]) != hipSuccess || hipEventCreate(&c->t1[k]) != hipSuccess) return (int)hipGetLastError();
    if (hipEventCreate(&c->tu0[0]) != hipSuccess || hipEventCreate(&c->tu0[1]) != hipSuccess) return (int)hipGetLastError();
  }
  c->timing = on != 0;
  c->timed_run = false;
  return 0;
}
extern "C" int nesvor_step_timing_read(void* handle, float* ms) {
  if (handle == nullptr || ms == nullptr) return (int)hipErrorInvalidValue;
  StepCtx* c = static_cast<StepCtx*>(handle);
  for (int k = 0; k < NESVOR_STEP_TIMED_SPANS; ++k) ms[k] = -1.f;
  if (!c->timed_run) return 0;
  for (int k = 0; k < NESVOR_STEP_TIMED_SPANS; ++k) {
    if (!c->span_used[k]) continue;
    // (the union span starts at the PREVIOUS run's owner pass: its first event is that run's tu0)
    hipEvent_t e0 = k == NESVOR_STEP_SPAN_HASHGRID_UNION ? c->tu0[(c->run_no - 1u) & 1u] : c->t0[k];
    if (hipEventSynchronize(c->t1[k]) != hipSuccess || hipEventElapsedTime(&ms[k], e0, c->t1[k]) != hipSuccess) return (int)hipGetLastError();
  }
  return 0;
}

extern "C" int nesvor_step_update(void* handle, const nesvor_step_t* desc) {
  if (handle == nullptr || desc == nullptr) return (int)hipErrorInvalidValue;
  static_cast<StepCtx*>(handle)->d = *desc;
  return 0;
}

extern "C" void nesvor_step_destroy(void* handle) {
  if (handle == nullptr) return;
  StepCtx* c = static_cast<StepCtx*>(handle);
  (void)hipEventDestroy(c->ev_fork); (void)hipEventDestroy(c->ev_pose); (void)hipEventDestroy(c->ev_agg); (void)hipEventDestroy(c->ev_owner);
  (void)hipEventDestroy(c->ev_bwd); (void)hipEventDestroy(c->ev_sums); (void)hipEventDestroy(c->ev_norms);
  (void)hipEventDestroy(c->ev_sg0); (void)hipEventDestroy(c->ev_sg1); (void)hipEventDestroy(c->ev_owner_early);
  if (c->tu0[0] != nullptr) { (void)hipEventDestroy(c->tu0[0]); (void)hipEventDestroy(c->tu0[1]); }
  if (c->wimg != nullptr) (void)hipFree(c->wimg);
  for (int k = 0; k < NESVOR_STEP_TIMED_SPANS; ++k) {
    if (c->t0[k] != nullptr) (void)hipEventDestroy(c->t0[k]);
    if (c->t1[k] != nullptr) (void)hipEventDestroy(c->t1[k]);
  }
  delete c;
}

extern "C" int nesvor_step_run(void* handle, const float* xyz, const float* v, const int64_t* slice_idx, uint64_t seed,
                               uint64_t offset, float* losses, int phase, int split_level, const nesvor_adamw_t* adam,
                               void* stream) {
  if (handle == nullptr || xyz == nullptr || v == nullptr || slice_idx == nullptr || losses == nullptr) return (int)hipErrorInvalidValue;
  StepCtx* ctx = static_cast<StepCtx*>(handle);
  const nesvor_step_t& d = ctx->d;
  hipStream_t main = (hipStream_t)stream, side = (hipStream_t)d.side_stream;
  const int B = d.B, S = d.S, n = d.n_slices, L = d.grid.n_levels;
  const int64_t N = (int64_t)B * S;
  const bool defer_join = (phase & NESVOR_STEP_DEFER_JOIN) != 0;
  phase &= ~NESVOR_STEP_DEFER_JOIN;
  if (phase < 0 || phase > 2 || (phase != 0) != (split_level > 0 && split_level < L)) return (int)hipErrorInvalidValue;
  if (d.has_b && phase != 0) return (int)hipErrorInvalidValue;  // the bias field's global mean needs the host's all-reduce: Python path
  float* c = d.has_c ? d.small : nullptr;       // slice scale n softmax(logit_coef)
  float* mat = d.small + n;                      // (n,3,4) pose matrices
  float* acc = d.small + 13 * n;                 // [dc (n) | dmat (n,12)], zero-filled by the prologue
  float* dc = acc;
  float* dmat = acc + n;
  // max |dpe|, raised by the density network's backward and read by the hash-grid backward (which then skips its own pass
  // over dpe); zero-filled with the accumulators.  With a bias field dpe is a sum of two networks' gradients: no bound.
  float* dpe_bound = d.has_b ? nullptr : d.small + 26 * n;
  // Operand bounds and weight norms of the three networks (nesvor_mlp_t.prep, split-operand mode): 3 x NESVOR_MLP_PREP_FLOATS floats
  // behind the accumulators, zero-filled with them by the prologue.  Nobody passes over an operand to find its bound: the
  // producing kernels publish them - the hash-grid forward max |pe|, the density network max |z|, the loss kernel max |d z_0|,
  // max |d log_var|, max |d log_bias|, sigma_net's backward max |d z_1..| - and one launch per network takes the weight norms on
  // the side stream, under the sampler and the hash-grid forward.
  float* prep_d = d.small + 26 * n + 1;
  float* prep_s = prep_d + NESVOR_MLP_PREP_FLOATS;
  float* prep_b = prep_s + NESVOR_MLP_PREP_FLOATS;
  nesvor_mlp_t net_d = d.density, net_s = d.sigma, net_b = d.bias_net;
  net_d.prep = prep_d; net_s.prep = prep_s; net_b.prep = prep_b;
  net_d.y_absmax = d.has_lv ? prep_s + NESVOR_MLP_PREP_XB : nullptr;  // z rows 1.. are sigma_net's matrix input
  void* wimg_d = ctx->wimg;
  void* wimg_s = ctx->wimg != nullptr ? static_cast<char*>(ctx->wimg) + ctx->wimg_stride : nullptr;
  void* wimg_b = ctx->wimg != nullptr ? static_cast<char*>(ctx->wimg) + 2 * ctx->wimg_stride : nullptr;
  // (an image set is valid for the launches of THIS call and - data-parallel runs - of the phase-2 call that follows a phase-1
  //  call: the weights only change in between runs; networks whose shape has no images - 0 bytes - keep building their own)
  net_d.weight_images = nesvor_mlp_weight_images_bytes(&net_d) > 0 ? wimg_d : nullptr;
  net_s.weight_images = (d.has_lv && nesvor_mlp_weight_images_bytes(&net_s) > 0) ? wimg_s : nullptr;
  net_b.weight_images = (d.has_b && nesvor_mlp_weight_images_bytes(&net_b) > 0) ? wimg_b : nullptr;
  // (modes 2 and 4 - the split and its leading term alone - share scales, bounds and images)
  auto scaled = [](const nesvor_mlp_t& n_) { return n_.bf16_operands == 2 || n_.bf16_operands == 4; };
  const bool split_d = scaled(net_d), split_s = d.has_lv && scaled(net_s), split_b = d.has_b && scaled(net_b);
  const int layout = NESVOR_LAYOUT_FEATURE_MAJOR;
  const bool overlap_owner = (d.overlap_owner & 1) != 0;
  // AdamW on the table inside the owner pass (single call covers gradient and update, nothing to exchange in between);
  // the table is the LAST segment of the flat buffers
  const int64_t table_off = d.table - d.flat_param;
  const bool fuse_adamw = adam != nullptr && phase == 0 && (d.overlap_owner & 2) != 0 && d.table != nullptr && d.flat_param != nullptr &&
                          d.g_table == d.flat_grad + table_off && table_off >= 0 && table_off < d.flat_numel;

  // Pixel-feature gradients arrive as one row per 16-sample group (summed in the kernel) only from the wave-specialised fused
  // backward: per NETWORK, divisibility AND nesvor_mlp_backward_fused_ok (round-5 advisor: a network that kernel refuses - e.g. a
  // sigma_net with more than 32 inputs at two hidden layers, or N beyond 32-bit row offsets - runs as a dX launch + a dW launch
  // through dpre_scratch and writes one row per SAMPLE; nesvor_amd/direct.py sizes dxa / dxa_b by the same rule)
  const int group_div = (N % 16 == 0 && S % 16 == 0 && d.ks % 16 == 0) ? 1 : 0;
  const int group_sums_s = (group_div && d.has_lv && nesvor_mlp_backward_fused_ok(&net_s, N)) ? 1 : 0;
  const int group_sums_b = (group_div && d.has_b && nesvor_mlp_backward_fused_ok(&net_b, N)) ? 1 : 0;
  const int rows_pp_s = group_sums_s ? S / 16 : S, rows_pp_b = group_sums_b ? S / 16 : S;
  const bool first_is_sigma = d.has_lv && d.ks;
  const float* dxa_first = first_is_sigma ? d.dxa : ((d.has_b && d.ks) ? d.dxa_b : nullptr);
  const int rows_pp_first = first_is_sigma ? rows_pp_s : rows_pp_b;
  // NESVOR_SLICE_GRADS=by_slice: one workgroup per slice, no atomics, reproducible sums (where the batch fits its pixel list).
  // Default: the per-pixel atomic kernel - measured 1.136-1.140 ms per iteration against 1.152-1.160: a few hundred workgroups
  // with serial sums start behind the owner pass's workgroups and finish later (61 us) than 4096 x 30 contended atomics (46 us)
  auto slice_grads = [&](const float* dc_pix, const float* dlvs_pix, const float* dxa, int rows_per_pixel, const float* dpix, float* dc_,
                         float* dlvs_, float* dse_, float* dmat_, hipStream_t st_) -> int {
    static const bool by_slice = []() { const char* e = getenv("NESVOR_SLICE_GRADS"); return e != nullptr && strcmp(e, "by_slice") == 0; }();
    if (by_slice) {
      const int e = nesvor_slice_grads_by_slice(slice_idx, dc_pix, dlvs_pix, dxa, dpix, dc_, dlvs_, dse_, dmat_, B, rows_per_pixel, d.ks, n, st_);
      if (e != (int)hipErrorInvalidValue) return e;
    }
    return nesvor_slice_grads(slice_idx, dc_pix, dlvs_pix, dxa, dpix, dc_, dlvs_, dse_, dmat_, B, rows_per_pixel, d.ks, st_);
  };
  // (the bias field's second consumer of the slice embedding keeps round 4's single launch behind the sampler backward)
  const bool early_sg = overlap_owner && !d.has_b;
  if (ctx->timing) { for (int k = 0; k < NESVOR_STEP_TIMED_SPANS; ++k) ctx->span_used[k] = false; ctx->timed_run = true; }
  if (phase != 2) {
    // ---- forward
    // ONE launch: slice scales, pose matrices, zero-fill of the per-slice accumulators and operand bounds, and the pose regulariser
    // with its gradient (a function of the parameters alone).  Behind it, on the SAME stream: the networks' weight norms and the
    // slice embedding's bound.  NESVOR_STEP_HEAD=side (A/B switch) restores rounds 4-5's arrangement - both on the side stream,
    // forked behind the prologue, joined in front of the first network / at the epilogue: the fork's event record and the joins'
    // waits each hold the main stream for 6-8 us (kernel -> marker -> kernel), more than the launches they hide.
    static const bool head_on_side = []() { const char* e = getenv("NESVOR_STEP_HEAD"); return e != nullptr && e[0] == 's'; }();
    const bool pose_in_prologue = d.opt_T && !head_on_side;
    NESVOR_TRY(nesvor_step_prologue_pose(d.has_c ? d.logit_coef : nullptr, c, d.axisangle, mat, acc, 13 * n + 1 + 3 * NESVOR_MLP_PREP_FLOATS, n,
                                         pose_in_prologue ? d.axisangle_init : nullptr, pose_in_prologue ? d.trans_terms : nullptr,
                                         pose_in_prologue ? d.g_trans : nullptr, main));
    const bool any_split = split_d || split_s || split_b;
    hipStream_t head = main;
    if (head_on_side && (d.opt_T || any_split)) {
      if (hipEventRecord(ctx->ev_fork, main) != hipSuccess || hipStreamWaitEvent(side, ctx->ev_fork, 0) != hipSuccess) return (int)hipGetLastError();
      head = side;
    }
    if (any_split) {
      const nesvor_mlp_t* nets[3]; float* preps[3]; void* imgs[3]; int nn = 0;
      if (split_d) { nets[nn] = &net_d; imgs[nn] = const_cast<void*>(net_d.weight_images); preps[nn++] = prep_d; }
      if (split_s) { nets[nn] = &net_s; imgs[nn] = const_cast<void*>(net_s.weight_images); preps[nn++] = prep_s; }
      if (split_b) { nets[nn] = &net_b; imgs[nn] = const_cast<void*>(net_b.weight_images); preps[nn++] = prep_b; }
      // ONE launch: the weight norms of all networks and - the pixel features of sigma_net are rows of the slice embedding -
      // the table's maximum as their bound
      const bool se_bound = split_s && d.ks > 0;
      NESVOR_TRY(nesvor_mlp_prepare_weights_images(nets, preps, imgs, nn, se_bound ? d.slice_embedding : nullptr, (int64_t)n * d.ks,
                                                   se_bound ? prep_s + NESVOR_MLP_PREP_XA : nullptr, head));
      if (head_on_side && hipEventRecord(ctx->ev_norms, side) != hipSuccess) return (int)hipGetLastError();
    }
    if (head_on_side && d.opt_T) {
      NESVOR_TRY(nesvor_trans_loss(d.axisangle, d.axisangle_init, d.trans_terms, d.g_trans, n, side));
      if (hipEventRecord(ctx->ev_pose, side) != hipSuccess) return (int)hipGetLastError();
    }
    ctx->head_on_side = head_on_side;
    {
      Span t(ctx, NESVOR_STEP_SPAN_PSF_FWD, main);
      NESVOR_TRY(nesvor_psf_transform_forward_rng_gather(mat, slice_idx, xyz, d.psf_sigma, seed, offset, d.bounding_box, d.x, d.u, B, S,
                                                         d.ks > 0 ? d.slice_embedding : nullptr, d.ks > 0 ? d.se : nullptr, d.ks, main));
    }
    const int pipe = ctx->pending_join ? ctx->pending_pipe_level : 0;
    if (pipe > 0 && pipe < L && S >= 128) {
      // the previous run's table update arrives level range by level range: the forward of levels [0, pipe) runs while the owner
      // pass still updates the finer ones
      if (hipStreamWaitEvent(main, ctx->ev_owner_early, 0) != hipSuccess) return (int)hipGetLastError();
      {
        Span t(ctx, NESVOR_STEP_SPAN_HASHGRID_FWD, main);
        NESVOR_TRY(nesvor_hashgrid_forward_levels(&d.grid, d.u, d.table, d.pe, N, layout, split_d ? prep_d + NESVOR_MLP_PREP_XB : nullptr, 0, pipe, main));
      }
      if (hipStreamWaitEvent(main, ctx->ev_owner, 0) != hipSuccess) return (int)hipGetLastError();
      {
        Span t(ctx, NESVOR_STEP_SPAN_HASHGRID_FWD_LATE, main);
        NESVOR_TRY(nesvor_hashgrid_forward_levels(&d.grid, d.u, d.table, d.pe, N, layout, split_d ? prep_d + NESVOR_MLP_PREP_XB : nullptr, pipe, L, main));
      }
      if (ctx->timing) { (void)hipEventRecord(ctx->t1[NESVOR_STEP_SPAN_HASHGRID_UNION], main); ctx->span_used[NESVOR_STEP_SPAN_HASHGRID_UNION] = true; }
      ctx->pending_join = false;
    } else {
    if (ctx->pending_join) {  // the previous run left its table update on the side stream
      if (hipStreamWaitEvent(main, ctx->ev_owner, 0) != hipSuccess) return (int)hipGetLastError();
      ctx->pending_join = false;
    }
    {
      Span t(ctx, NESVOR_STEP_SPAN_HASHGRID_FWD, main);
      NESVOR_TRY(nesvor_hashgrid_forward_bounded(&d.grid, d.u, d.table, d.pe, N, layout | (S >= 128 ? NESVOR_LAYOUT_CLUSTERED : 0),
                                                 split_d ? prep_d + NESVOR_MLP_PREP_XB : nullptr, main));
    }
    }
    if (head_on_side && any_split && hipStreamWaitEvent(main, ctx->ev_norms, 0) != hipSuccess) return (int)hipGetLastError();
    {
      Span t(ctx, NESVOR_STEP_SPAN_MLP_FWD_DENSITY, main);
      NESVOR_TRY(nesvor_mlp_forward(&net_d, nullptr, d.pe, d.z, d.saved_d, N, main));
    }
    if (d.has_b) {
      // (the bias field is off BASELINE's headline configuration: its input bounds by a pass of their own)
      if (split_b) NESVOR_TRY(nesvor_mlp_prepare(&net_b, d.se, d.pe, nullptr, N, prep_b, NESVOR_MLP_WHAT_INPUT, main));
      NESVOR_TRY(nesvor_mlp_forward(&net_b, d.se, d.pe, d.log_bias, d.saved_b, N, main));
      hipLaunchKernelGGL(mean_partial_kernel, dim3(256), dim3(256), 0, main, d.log_bias, N, d.mean_scratch);
      hipLaunchKernelGGL(mean_final_kernel, dim3(1), dim3(256), 0, main, d.mean_scratch, 256, 1.f / (float)N, d.lb_mean);
    }
    if (d.has_lv) {
      Span t(ctx, NESVOR_STEP_SPAN_MLP_FWD_SIGMA, main);
      NESVOR_TRY(nesvor_mlp_forward(&net_s, d.se, d.z, d.log_var, d.saved_s, N, main));
    }
    // ---- losses: values and gradients in one launch
    const int z_rows = 1 + d.n_features_z, written = 1 + (d.has_lv ? d.n_features_z : 0);
    if (written < z_rows) {
      if (hipMemsetAsync(d.dz + (size_t)written * N, 0, sizeof(float) * (size_t)(z_rows - written) * N, main) != hipSuccess) return (int)hipGetLastError();
    }
    nesvor_loss_t la;
    std::memset(&la, 0, sizeof(la));
    la.z0 = d.z; la.log_var = d.has_lv ? d.log_var : nullptr; la.log_bias = d.has_b ? d.log_bias : nullptr;
    la.x = d.x; la.v = v; la.slice_idx = slice_idx; la.c = c; la.log_var_slice = d.has_lvs ? d.log_var_slice : nullptr;
    la.log_bias_mean = d.has_b ? d.lb_mean : nullptr;
    la.gw = d.gw; la.loss_pix = d.loss_pix; la.dz0 = d.dz; la.dlog_var = d.has_lv ? d.dlv : nullptr;
    la.dlog_bias = d.has_b ? d.dlb : nullptr; la.dx = d.opt_T ? d.dxl : nullptr;
    la.dc_pix = d.has_c ? d.pix : nullptr; la.dlvs_pix = d.has_lvs ? d.pix + B : nullptr;
    la.B = B; la.S = S; la.reg_type = d.reg_type; la.delta = d.delta;
    la.dz0_absmax = prep_d + NESVOR_MLP_PREP_DY; la.dlog_var_absmax = prep_s + NESVOR_MLP_PREP_DY; la.dlog_bias_absmax = prep_b + NESVOR_MLP_PREP_DY;
    {
      Span t(ctx, NESVOR_STEP_SPAN_LOSS, main);
      NESVOR_TRY(nesvor_imaging_loss(&la, main));
    }
    // ---- backward through the networks
    // every network writes its per-workgroup partial parameter gradients into its own third of `partial` (the host allocates
    // 3 x NESVOR_STEP_MLP_PARTIALS rows of the widest network)
    int widest = d.n_density_params;
    if (d.has_lv && d.n_sigma_params > widest) widest = d.n_sigma_params;
    if (d.has_b && d.n_bias_params > widest) widest = d.n_bias_params;
    float* part_d = d.partial;
    float* part_s = d.partial + (size_t)NESVOR_STEP_MLP_PARTIALS * widest;
    float* part_b = d.partial + 2 * (size_t)NESVOR_STEP_MLP_PARTIALS * widest;
    if (d.has_lv) {  // (its input gradient = rows 1.. of dz: raises the density network's upstream bound next to the loss kernel's row 0)
      Span t(ctx, NESVOR_STEP_SPAN_MLP_BWD_SIGMA, main);
      NESVOR_TRY(mlp_backward_into(net_s, group_sums_s, d.se, d.z, d.dlv, d.saved_s, d.ks ? d.dxa : nullptr, d.dz + N, part_s,
                                   d.g_sigma, d.n_sigma_params, N, main, d.dpre_scratch, prep_d + NESVOR_MLP_PREP_DY));  // (a scalar publish: slot 0)
    }
    // Per-slice sums that do not depend on the hash-grid backward - d slice scale, d slice variance and the slice embedding's
    // gradient (sigma_net's input gradient) - go to the side stream NOW, under the density network's backward and the aggregation
    // pass; the pose matrices' share is added by the sampler backward itself (nesvor_psf_transform_backward_rng_slices).  Round 4
    // ran one slice_grads launch behind the sampler backward: 46 us of the step's serial tail next to the owner pass.
    if (early_sg) {
      if (hipEventRecord(ctx->ev_sg0, main) != hipSuccess || hipStreamWaitEvent(side, ctx->ev_sg0, 0) != hipSuccess) return (int)hipGetLastError();
      NESVOR_TRY(slice_grads(d.has_c ? d.pix : nullptr, d.has_lvs ? d.pix + B : nullptr, (d.has_lv && d.ks) ? d.dxa : nullptr, rows_pp_s, nullptr, dc,
                             d.has_lvs ? d.g_log_var_slice : nullptr, d.ks ? d.g_slice_embedding : nullptr, nullptr, side));
      if (hipEventRecord(ctx->ev_sg1, side) != hipSuccess) return (int)hipGetLastError();
    }
    {
      Span t(ctx, NESVOR_STEP_SPAN_MLP_BWD_DENSITY, main);
      NESVOR_TRY(mlp_backward_into(net_d, 0, nullptr, d.pe, d.dz, d.saved_d, nullptr, d.dpe, part_d, d.g_density,
                                   d.n_density_params, N, main, d.dpre_scratch, dpe_bound));
    }
    if (d.has_b) {
      NESVOR_TRY(mlp_backward_into(net_b, group_sums_b, d.se, d.pe, d.dlb, d.saved_b, d.ks ? d.dxa_b : nullptr, d.dpe_b, part_b,
                                   d.g_bias_net, d.n_bias_params, N, main, d.dpre_scratch));
      const int64_t nb = (int64_t)d.kb_bias * N;
      hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((nb / 4 + 255) / 256 + 1)), dim3(256), 0, main, d.dpe, d.dpe_b, nb);
    }
    {
      const float* in[3]; float* out[3]; int cols[3], n_jobs = 0;
      in[n_jobs] = part_d; out[n_jobs] = d.g_density; cols[n_jobs++] = d.n_density_params;
      if (d.has_lv) { in[n_jobs] = part_s; out[n_jobs] = d.g_sigma; cols[n_jobs++] = d.n_sigma_params; }
      if (d.has_b) { in[n_jobs] = part_b; out[n_jobs] = d.g_bias_net; cols[n_jobs++] = d.n_bias_params; }
      // Only the closing AdamW (or the caller's optimizer / gradient exchange, which joins the side stream) reads these sums.
      // NESVOR_STEP_SUMS_SIDE=1 runs them on the side stream, under the aggregation pass, instead of in front of it (9 us of
      // the critical path) - measured in one job, alternating (gpurun_out/s2j2): 895 / 900 it/s on the main stream, 893 / 896
      // on the side stream: the aggregation pass starts later behind the cross-stream hand-over than it gains.  Off.
      hipStream_t sums_stream = main;
      ctx->sums_on_side = false;
      static const bool sums_side = []() { const char* e = getenv("NESVOR_STEP_SUMS_SIDE"); return e != nullptr && strcmp(e, "1") == 0; }();  // A/B switch
      const bool on_side = overlap_owner && sums_side;
      if (on_side) {
        if (hipEventRecord(ctx->ev_bwd, main) != hipSuccess || hipStreamWaitEvent(side, ctx->ev_bwd, 0) != hipSuccess) return (int)hipGetLastError();
        sums_stream = side;
      }
      NESVOR_TRY(nesvor_sum_rows_multi(in, out, cols, cols, n_jobs, NESVOR_STEP_MLP_PARTIALS, sums_stream));
      if (on_side) {
        if (hipEventRecord(ctx->ev_sums, side) != hipSuccess) return (int)hipGetLastError();
        ctx->sums_on_side = true;
      }
    }
  }
  // ---- hash-grid backward (+ input gradient when the poses are optimised)
  float* du = d.opt_T ? d.du : nullptr;
  if (phase == 0) {
    {
      // (Two workgroups per cloud for batches that do not fill the chip - levels [0, 12) on the main stream, [12, 16) concurrently
      //  on the side stream, each with its own input-gradient buffer - were tried in round 5: the sibling launch repeats sort and
      //  set-up, 36 + 50 us side by side against 63 us, and the cross-stream hand-overs ate the rest: 0.285 -> 0.294 ms at 2^17
      //  points.  Not kept.)
      Span t(ctx, NESVOR_STEP_SPAN_HASHGRID_BWD_AGGREGATE, main);
      NESVOR_TRY(nesvor_hashgrid_backward_bounded(&d.grid, d.u, d.table, d.dpe, d.g_table, du, N, layout, d.hg_workspace, 1, 0, L,
                                                  d.queue_scale, dpe_bound, main));
    }
    // the owner pass only finishes grad_table (or, fused, takes the table's AdamW step): it runs under the sampler backward
    // and the per-slice bookkeeping
    hipStream_t owner_stream = main;
    if (overlap_owner) {
      if (hipEventRecord(ctx->ev_agg, main) != hipSuccess || hipStreamWaitEvent(side, ctx->ev_agg, 0) != hipSuccess) return (int)hipGetLastError();
      owner_stream = side;
    }
    // Pipelined table update (round 6; NESVOR_STEP_PIPE_LEVEL=<level>, default 0 = OFF): when this run leaves the table's update on the
    // side stream for the next run to join (NESVOR_STEP_DEFER_JOIN), the owner pass goes out as TWO launches - levels [0, pipe), then
    // the rest - and the next run's forward starts on the first range while the second is still being updated.  Owner pass (84 us
    // with the table's AdamW) and forward (70 us) are strictly one after the other on the step's critical path, and both are
    // latency-bound - yet MEASURED (tools/ab_pipe_level.sh, profiles/r06_ab_pipe_level.log, two alternating rounds): 0.944 ms ->
    // 0.960-0.970 ms at pipe levels 11 / 12 / 14.  The overlapped owner launch stretches by 24 us for 44 us of forward (the two
    // kernels contend: ~45 % overlap efficiency), the second forward launch repeats the set-up (44 + 44 us against 72), and the
    // extra event hand-overs take the rest.  Kept as a switch; off.
    static const int pipe_level = []() { const char* e = getenv("NESVOR_STEP_PIPE_LEVEL"); return e != nullptr ? atoi(e) : 0; }();
    const int pipe = (fuse_adamw && overlap_owner && defer_join && S >= 128 && pipe_level > 0 && pipe_level < L) ? pipe_level : 0;
    ctx->pending_pipe_level = pipe;
    if (ctx->timing) (void)hipEventRecord(ctx->tu0[ctx->run_no & 1u], owner_stream);
    {
      Span t(ctx, NESVOR_STEP_SPAN_HASHGRID_BWD_OWNER, owner_stream);  // (with fuse_adamw: the owner pass AND the table's AdamW step)
      if (pipe > 0) {
        NESVOR_TRY(nesvor_hashgrid_backward_adamw_levels(&d.grid, d.u, d.flat_param + table_off, d.dpe, d.g_table, du, N, layout, d.hg_workspace,
                                                         2, 0, pipe, d.queue_scale, dpe_bound, d.flat_exp_avg + table_off,
                                                         d.flat_exp_avg_sq + table_off, adam, owner_stream));
        if (hipEventRecord(ctx->ev_owner_early, side) != hipSuccess) return (int)hipGetLastError();
        NESVOR_TRY(nesvor_hashgrid_backward_adamw_levels(&d.grid, d.u, d.flat_param + table_off, d.dpe, d.g_table, du, N, layout, d.hg_workspace,
                                                         2, pipe, L, d.queue_scale, dpe_bound, d.flat_exp_avg + table_off,
                                                         d.flat_exp_avg_sq + table_off, adam, owner_stream));
      } else if (fuse_adamw)
        NESVOR_TRY(nesvor_hashgrid_backward_adamw(&d.grid, d.u, d.flat_param + table_off, d.dpe, d.g_table, du, N, layout, d.hg_workspace, 2,
                                                  d.queue_scale, dpe_bound, d.flat_exp_avg + table_off, d.flat_exp_avg_sq + table_off, adam,
                                                  owner_stream));
      else
        NESVOR_TRY(nesvor_hashgrid_backward_levels(&d.grid, d.u, d.table, d.dpe, d.g_table, du, N, layout, d.hg_workspace, 2, 0, L,
                                                   d.queue_scale, owner_stream));
    }
    if (overlap_owner && hipEventRecord(ctx->ev_owner, side) != hipSuccess) return (int)hipGetLastError();
  } else if (phase == 1) {
    // fine levels first (the end of the flat gradient): the host starts their all-reduce when this call returns
    return nesvor_hashgrid_backward_bounded(&d.grid, d.u, d.table, d.dpe, d.g_table, du, N, layout, d.hg_workspace, 3, split_level, L,
                                            d.queue_scale, dpe_bound, main);
  } else {
    // the coarse levels: aggregation on the main stream, their owner pass again under the rest of the step (the caller joins
    // the side stream before it reduces / applies the rest of the gradient)
    NESVOR_TRY(nesvor_hashgrid_backward_bounded(&d.grid, d.u, d.table, d.dpe, d.g_table, du, N, layout, d.hg_workspace, 1 | 4 | 8, 0,
                                                split_level, d.queue_scale, dpe_bound, main));
    hipStream_t owner_stream = main;
    if (overlap_owner) {
      if (hipEventRecord(ctx->ev_agg, main) != hipSuccess || hipStreamWaitEvent(side, ctx->ev_agg, 0) != hipSuccess) return (int)hipGetLastError();
      owner_stream = side;
    }
    NESVOR_TRY(nesvor_hashgrid_backward_bounded(&d.grid, d.u, d.table, d.dpe, d.g_table, du, N, layout, d.hg_workspace, 2 | 4 | 8, 0,
                                                split_level, d.queue_scale, dpe_bound, owner_stream));
    if (overlap_owner && hipEventRecord(ctx->ev_owner, side) != hipSuccess) return (int)hipGetLastError();
  }
  if (d.opt_T) {
    Span t(ctx, NESVOR_STEP_SPAN_PSF_BWD, main);
    if (early_sg)  // the pixels' pose gradients straight into their slices' rows
      NESVOR_TRY(nesvor_psf_transform_backward_rng_slices(mat, slice_idx, xyz, d.psf_sigma, seed, offset, d.bounding_box, d.dxl, du, nullptr, dmat, B, S, main));
    else
      NESVOR_TRY(nesvor_psf_transform_backward_rng(mat, slice_idx, xyz, d.psf_sigma, seed, offset, d.bounding_box, d.dxl, du, d.dpix, B, S, main));
  }
  if (early_sg) {
    if (hipStreamWaitEvent(main, ctx->ev_sg1, 0) != hipSuccess) return (int)hipGetLastError();
  } else {
    NESVOR_TRY(slice_grads(d.has_c ? d.pix : nullptr, d.has_lvs ? d.pix + B : nullptr, dxa_first, rows_pp_first, d.opt_T ? d.dpix : nullptr, dc,
                           d.has_lvs ? d.g_log_var_slice : nullptr, d.ks ? d.g_slice_embedding : nullptr, dmat, main));
    if (d.has_lv && d.has_b && d.ks)  // second consumer of the slice embedding
      NESVOR_TRY(slice_grads(nullptr, nullptr, d.dxa_b, rows_pp_b, nullptr, nullptr, nullptr, d.g_slice_embedding, nullptr, main));
  }
  if (ctx->head_on_side && d.opt_T && hipStreamWaitEvent(main, ctx->ev_pose, 0) != hipSuccess) return (int)hipGetLastError();
  const float img_scale = (d.reg_type == 0 ? d.delta : 1.f) / (float)N, img_off = d.reg_type == 0 ? -d.delta : 0.f;
  NESVOR_TRY(nesvor_step_epilogue(d.has_c ? dc : nullptr, c, d.has_c ? d.g_logit_coef : nullptr, d.opt_T ? dmat : nullptr, d.axisangle,
                                  d.opt_T ? d.g_trans : nullptr, d.w_T, d.opt_T ? d.g_axisangle : nullptr, d.loss_pix,
                                  d.opt_T ? d.trans_terms : nullptr, losses, n, B, img_scale, img_off, main));
  if (d.has_b) hipLaunchKernelGGL(square_kernel, dim3(1), dim3(1), 0, main, d.lb_mean, losses + 5);
  if (adam != nullptr) {
    if (ctx->sums_on_side) {
      if (hipStreamWaitEvent(main, ctx->ev_sums, 0) != hipSuccess) return (int)hipGetLastError();
      ctx->sums_on_side = false;
    }
    if (fuse_adamw) {
      // everything but the table; the table's update is the owner pass, which the next forward must wait for - here, or
      // (NESVOR_STEP_DEFER_JOIN) in the next run right before its hash-grid forward, so that the next iteration's prologue and
      // sampler run under the end of this one's table update
      NESVOR_TRY(nesvor_adamw_step(d.flat_param, d.flat_grad, d.flat_exp_avg, d.flat_exp_avg_sq, table_off, adam->lr, adam->beta1, adam->beta2,
                                   adam->eps, adam->weight_decay, adam->bias_correction1, adam->bias_correction2, adam->grad_scale, 1, main));
      // ... and whatever the caller laid out BEHIND the table (FlatParams puts the table last; a parameter tied with it for
      // the largest size would follow it): segments are 16-byte aligned, the owner pass covered exactly the table's entries
      const int64_t table_end = (table_off + (int64_t)(d.grid.offset[L - 1] + d.grid.size[L - 1]) * d.grid.n_features + 3) / 4 * 4;
      if (table_end < d.flat_numel)
        NESVOR_TRY(nesvor_adamw_step(d.flat_param + table_end, d.flat_grad + table_end, d.flat_exp_avg + table_end, d.flat_exp_avg_sq + table_end,
                                     d.flat_numel - table_end, adam->lr, adam->beta1, adam->beta2, adam->eps, adam->weight_decay,
                                     adam->bias_correction1, adam->bias_correction2, adam->grad_scale, 1, main));
      if (overlap_owner) {
        if (defer_join) ctx->pending_join = true;
        else if (hipStreamWaitEvent(main, ctx->ev_owner, 0) != hipSuccess) return (int)hipGetLastError();
      }
    } else {
      if (phase != 1 && overlap_owner && hipStreamWaitEvent(main, ctx->ev_owner, 0) != hipSuccess) return (int)hipGetLastError();
      NESVOR_TRY(nesvor_adamw_step(d.flat_param, d.flat_grad, d.flat_exp_avg, d.flat_exp_avg_sq, d.flat_numel, adam->lr, adam->beta1, adam->beta2,
                                   adam->eps, adam->weight_decay, adam->bias_correction1, adam->bias_correction2, adam->grad_scale, 1, main));
    }
  }
  ++ctx->run_no;
  return (int)hipGetLastError();
}

extern "C" int nesvor_step_join(void* handle, void* stream) {
  if (handle == nullptr) return (int)hipErrorInvalidValue;
  StepCtx* ctx = static_cast<StepCtx*>(handle);
  if (ctx->pending_join) {
    if (hipStreamWaitEvent((hipStream_t)stream, ctx->ev_owner, 0) != hipSuccess) return (int)hipGetLastError();
    ctx->pending_join = false;
  }
  return 0;
}
