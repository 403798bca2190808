/*
 * nesvor_hip.h — C ABI of libnesvor_hip.so, the MI355X (gfx950) native backend
 * for the NeSVoR INR-training hot path.
 *
 * Plain C: raw device pointers, sizes and an opaque hipStream_t (void*).  No
 * torch types.  Every entry point enqueues work on `stream` and returns the
 * hipError_t of the launch (0 == hipSuccess); nothing synchronises.
 * All tensors are dense, contiguous, fp32 unless stated otherwise.
 *
 * Each group cites the reference interface it replaces (paths relative to the
 * reference tree, daviddmc/NeSVoR v0.1.0).
 */
#ifndef NESVOR_HIP_H
#define NESVOR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NESVOR_MAX_LEVELS 32
#define NESVOR_MAX_MLP_LAYERS 4

/* ABI version; bumped on any signature change. */
int nesvor_hip_abi_version(void);

/* ------------------------------------------------------------------------
 * Rigid-transform conversion.  Replaces the pybind module
 * `nesvor.transform_convert_cuda`
 * (nesvor/transform/transform_convert_cuda.cpp:64-69; kernels
 * transform_convert_cuda_kernel.cu:14-440).
 *   ax  : (n,6)   [rotation vector | translation]
 *   mat : (n,3,4) [R | t]
 * ---------------------------------------------------------------------- */
int nesvor_axisangle2mat_forward(const float* ax, float* mat, int n, void* stream);
int nesvor_axisangle2mat_backward(const float* grad_mat, const float* ax, float* grad_ax, int n, void* stream);
int nesvor_mat2axisangle_forward(const float* mat, float* ax, int n, void* stream);
int nesvor_mat2axisangle_backward(const float* mat, const float* grad_ax, float* grad_mat, int n, void* stream);
/* double-precision variants (the reference dispatches float and double) */
int nesvor_axisangle2mat_forward_f64(const double* ax, double* mat, int n, void* stream);
int nesvor_axisangle2mat_backward_f64(const double* grad_mat, const double* ax, double* grad_ax, int n, void* stream);
int nesvor_mat2axisangle_forward_f64(const double* mat, double* ax, int n, void* stream);
int nesvor_mat2axisangle_backward_f64(const double* mat, const double* grad_ax, double* grad_mat, int n, void* stream);

/* Pose regulariser, forward + gradient in one launch.  Replaces NeSVoR.trans_loss
 * (nesvor/nesvor/models.py:357-363: axisangle2mat x2, RigidTransform.inv/.compose
 * transform.py:44-63, mat2axisangle, and their backwards):
 *   err_k = axisangle(inv(T_init,k) o T_k);  loss = mean(err_R^2) + 1e-3 mean(err_T^2)
 * loss_per_slice (n): each slice's share of the loss (sum == loss);
 * grad_ax (n,6): d loss / d ax. */
int nesvor_trans_loss(const float* ax, const float* ax_init, float* loss_per_slice, float* grad_ax, int n, void* stream);

/* ------------------------------------------------------------------------
 * Slice acquisition forward operator A.  Replaces
 * `nesvor.slice_acq_cuda.forward`
 * (nesvor/slice_acquisition/slice_acq_cuda.cpp:156-161; kernel
 * slice_acq_cuda_kernel.cu:17-171, host :954-991).
 *   transforms (n,3,4) translation in voxel units; vol (D,H,W);
 *   vol_mask (D,H,W) uint8/bool or NULL; slices_mask (n,h,w) uint8/bool or
 *   NULL; psf (d_p,h_p,w_p); slices (n,h,w) out — must be zero-filled by the
 *   caller; slices_weight (n,h,w) out (zero-filled) or NULL.
 * ---------------------------------------------------------------------- */
int nesvor_slice_acq_forward(const float* transforms, const float* vol, const uint8_t* vol_mask,
                             const uint8_t* slices_mask, const float* psf, float* slices, float* slices_weight,
                             int D, int H, int W, int d_p, int h_p, int w_p, int n, int h, int w,
                             float res_slice, int interp_psf, void* stream);

/* Coverage of the slice-acquisition group against the reference's kernels:
 *   * interp_psf != 0 (nearest-voxel sampling with a re-interpolated PSF) exists in all four reference kernels
 *     (slice_acq_cuda_kernel.cu:72-109 forward, :229-279 and :316-366 backward, :526-572 adjoint, :754-800 adjoint
 *     backward) and is built for all four: the forward through its `interp_psf` argument, the other three through the
 *     `*_interp` entry points further down (per-pixel scatter as the reference formulates it).  No caller in the reference
 *     tree ever switches the mode on (slice_acq.py:40-160, svort/srr.py:37-128 and svort/models.py read it from
 *     params["interp_psf"], which every configuration sets to False).
 *   * double precision is built (the *_f64 entry points below; the reference dispatches float and double).
 *   * PSF size: any (the forward kernel keeps PSFs of up to 1024 taps as an LDS list of their non-zero taps and walks
 *     larger ones in global memory; the others read the dense PSF array).
 *
 * Adjoint operator A^T (+ optional equalisation) and backward of A, linear-interpolation mode.
 * Replace `nesvor.slice_acq_cuda.adjoint_forward` / `.backward`
 * (slice_acq_cuda.cpp:156-161; kernels slice_acq_cuda_kernel.cu:472-693 and :173-470).
 * Evaluated as a gather over voxels (no atomics, see csrc/slice_acq.hip).
 *   adjoint_forward: slices (n,h,w) -> vol (D,H,W) overwritten; vol_weight (D,H,W) out or NULL;
 *                    equalize != 0 divides vol by the accumulated weight where it is > 0.
 *   backward       : grad_slices (n,h,w) -> grad_vol (D,H,W) overwritten (or NULL) and
 *                    grad_transforms (n,3,4) overwritten (or NULL).
 *   scratch        : 2*n*h*w floats (adjoint) / n*h*w floats (backward) of device memory. */
int nesvor_slice_acq_adjoint_forward(const float* transforms, const float* psf, const float* slices,
                                     const uint8_t* slices_mask, const uint8_t* vol_mask, float* vol, float* vol_weight,
                                     float* scratch, int D, int H, int W, int d_p, int h_p, int w_p, int n, int h, int w,
                                     float res_slice, int equalize, void* stream);
/* Backward of A^T: replaces `nesvor.slice_acq_cuda.adjoint_backward` (slice_acq_cuda.cpp:156-161; kernels
 * slice_acq_cuda_kernel.cu:672-950, host :1079-1131), linear mode.  grad_vol (D,H,W) is divided IN PLACE by
 * max(vol_weight, 1e-3) where vol_weight > 0 when equalize != 0, exactly like the reference; vol is the
 * equalised adjoint output (read only when equalize).  grad_slices (n,h,w) must be zero-filled by the caller
 * (pixels with no PSF weight are left untouched), grad_transforms (n,3,4) is overwritten; either may be NULL. */
int nesvor_slice_acq_adjoint_backward(const float* transforms, float* grad_vol, const float* vol_weight,
                                      const uint8_t* vol_mask, const float* psf, const float* slices,
                                      const uint8_t* slices_mask, const float* vol, float* grad_slices,
                                      float* grad_transforms, int D, int H, int W, int d_p, int h_p, int w_p, int n, int h,
                                      int w, float res_slice, int equalize, void* stream);
int nesvor_slice_acq_backward(const float* transforms, const float* vol, const uint8_t* vol_mask, const float* psf,
                              const float* grad_slices, const uint8_t* slices_mask, float* grad_vol,
                              float* grad_transforms, float* scratch, int D, int H, int W, int d_p, int h_p, int w_p,
                              int n, int h, int w, float res_slice, void* stream);
/* interp_psf = true for the three operators above (slice_acq_cuda_kernel.cu:229-370, :526-606, :754-836): same argument
 * meaning, no scratch.  The kernels ACCUMULATE (atomics): vol / vol_weight, grad_vol / grad_transforms, grad_slices /
 * grad_transforms must be zero-filled by the caller (as the reference's host functions allocate them); any may be NULL where
 * the operator allows it.  adjoint_forward: equalize != 0 needs vol_weight. */
int nesvor_slice_acq_adjoint_forward_interp(const float* transforms, const float* psf, const float* slices,
                                            const uint8_t* slices_mask, const uint8_t* vol_mask, float* vol, float* vol_weight,
                                            int D, int H, int W, int d_p, int h_p, int w_p, int n, int h, int w,
                                            float res_slice, int equalize, void* stream);
int nesvor_slice_acq_backward_interp(const float* transforms, const float* vol, const uint8_t* vol_mask, const float* psf,
                                     const float* grad_slices, const uint8_t* slices_mask, float* grad_vol,
                                     float* grad_transforms, int D, int H, int W, int d_p, int h_p, int w_p, int n, int h, int w,
                                     float res_slice, void* stream);
int nesvor_slice_acq_adjoint_backward_interp(const float* transforms, float* grad_vol, const float* vol_weight,
                                             const uint8_t* vol_mask, const float* psf, const float* slices,
                                             const uint8_t* slices_mask, const float* vol, float* grad_slices,
                                             float* grad_transforms, int D, int H, int W, int d_p, int h_p, int w_p, int n,
                                             int h, int w, float res_slice, int equalize, void* stream);
int nesvor_slice_acq_adjoint_forward_interp_f64(const double* transforms, const double* psf, const double* slices,
                                                const uint8_t* slices_mask, const uint8_t* vol_mask, double* vol,
                                                double* vol_weight, int D, int H, int W, int d_p, int h_p, int w_p, int n, int h,
                                                int w, double res_slice, int equalize, void* stream);
int nesvor_slice_acq_backward_interp_f64(const double* transforms, const double* vol, const uint8_t* vol_mask,
                                         const double* psf, const double* grad_slices, const uint8_t* slices_mask,
                                         double* grad_vol, double* grad_transforms, int D, int H, int W, int d_p, int h_p,
                                         int w_p, int n, int h, int w, double res_slice, void* stream);
int nesvor_slice_acq_adjoint_backward_interp_f64(const double* transforms, double* grad_vol, const double* vol_weight,
                                                 const uint8_t* vol_mask, const double* psf, const double* slices,
                                                 const uint8_t* slices_mask, const double* vol, double* grad_slices,
                                                 double* grad_transforms, int D, int H, int W, int d_p, int h_p, int w_p,
                                                 int n, int h, int w, double res_slice, int equalize, void* stream);
/* double-precision variants of the four entry points (the reference dispatches float and double,
 * slice_acq_cuda_kernel.cu:970, :1010, :1046, :1114): same contracts, every float* a double*, res_slice a double. */
int nesvor_slice_acq_forward_f64(const double* transforms, const double* vol, const uint8_t* vol_mask,
                                 const uint8_t* slices_mask, const double* psf, double* slices, double* slices_weight,
                                 int D, int H, int W, int d_p, int h_p, int w_p, int n, int h, int w,
                                 double res_slice, int interp_psf, void* stream);
int nesvor_slice_acq_adjoint_forward_f64(const double* transforms, const double* psf, const double* slices,
                                         const uint8_t* slices_mask, const uint8_t* vol_mask, double* vol, double* vol_weight,
                                         double* scratch, int D, int H, int W, int d_p, int h_p, int w_p, int n, int h, int w,
                                         double res_slice, int equalize, void* stream);
int nesvor_slice_acq_backward_f64(const double* transforms, const double* vol, const uint8_t* vol_mask, const double* psf,
                                  const double* grad_slices, const uint8_t* slices_mask, double* grad_vol,
                                  double* grad_transforms, double* scratch, int D, int H, int W, int d_p, int h_p, int w_p,
                                  int n, int h, int w, double res_slice, void* stream);
int nesvor_slice_acq_adjoint_backward_f64(const double* transforms, double* grad_vol, const double* vol_weight,
                                          const uint8_t* vol_mask, const double* psf, const double* slices,
                                          const uint8_t* slices_mask, const double* vol, double* grad_slices,
                                          double* grad_transforms, int D, int H, int W, int d_p, int h_p, int w_p, int n, int h,
                                          int w, double res_slice, int equalize, void* stream);

/* ------------------------------------------------------------------------
 * Multi-resolution hash-grid encoding.  Replaces `tinycudann.Encoding`
 * (external module; call site nesvor/nesvor/models.py:22-25, config
 * models.py:102-111).  See oracle/hashgrid.py for the algorithm statement.
 * ---------------------------------------------------------------------- */
typedef struct {
  int32_t n_levels;
  int32_t n_features;                 /* features per level: 1, 2, 4 or 8 */
  float scale[NESVOR_MAX_LEVELS];     /* fp32 grid scale per level */
  uint32_t res[NESVOR_MAX_LEVELS];    /* vertices per axis */
  uint32_t size[NESVOR_MAX_LEVELS];   /* entries in the level */
  uint32_t offset[NESVOR_MAX_LEVELS]; /* entry offset into the flat table */
  uint32_t hashed[NESVOR_MAX_LEVELS]; /* 1: spatial hash, 0: dense index */
} nesvor_grid_t;

/* layout of the encoded matrix pe / dpe */
#define NESVOR_LAYOUT_ROW_MAJOR 0     /* (N, L*F): what tcnn returns to PyTorch */
#define NESVOR_LAYOUT_FEATURE_MAJOR 1 /* (L*F, N): coalesced producer/consumer layout */
#define NESVOR_LAYOUT_CLUSTERED 4     /* forward only, OR-ed into `layout`: every 256 consecutive points are spatially
                                         clustered (the S PSF samples of a slice pixel are contiguous) - selects the
                                         one-workgroup-per-cloud forward kernel; results do not depend on the hint */
#define NESVOR_LAYOUT_UNCLUSTERED 8   /* backward (owner-computes variants) only, OR-ed into `layout`: consecutive points are NOT
                                         spatially clustered (uniform points, a shuffled batch).  The backward first orders the
                                         points by the cells of a coarse lattice (counting sort, three small launches, scratch
                                         inside `workspace`) and hands every workgroup of the aggregation pass 256 points of
                                         neighbouring cells; the gradients do not depend on the hint beyond the owner pass's
                                         summation order.  Without it the backward assumes what the training step produces:
                                         the S PSF samples of a pixel are contiguous.  tcnn's scatter, which this replaces, is
                                         distribution-agnostic (reference call site nesvor/nesvor/models.py:25). */
#define NESVOR_LAYOUT_DY_SCRATCH 16   /* with NESVOR_LAYOUT_UNCLUSTERED | NESVOR_LAYOUT_FEATURE_MAJOR: the workspace was sized by
                                         nesvor_hashgrid_backward_workspace_bytes_ex(.., the same layout) and has room for dpe
                                         re-ordered into rows (4 L F bytes per point): the aggregation pass then reads whole rows
                                         instead of 4-byte words at scattered columns (1.80 -> 0.9 ms at N = 2^20 uniform points) */
#define NESVOR_LAYOUT_MASK 3          /* the layout proper; the other bits are hints */

/* u (N,3) in [0,1]; table flat fp32; pe out. */
int nesvor_hashgrid_forward(const nesvor_grid_t* grid, const float* u, const float* table, float* pe,
                            int64_t N, int layout, void* stream);
/* The same forward; additionally raises the slotted bound pe_absmax (NESVOR_ABSMAX_FLOATS floats, zero-filled by the caller; may
 * be NULL) to max |pe| - the input bound of the network that consumes pe (nesvor_mlp_t.prep + NESVOR_MLP_PREP_XB). */
int nesvor_hashgrid_forward_bounded(const nesvor_grid_t* grid, const float* u, const float* table, float* pe,
                                    int64_t N, int layout, float* pe_absmax, void* stream);
/* The forward for a batch whose consecutive points are NOT spatially clustered (uniform points, a shuffled batch, one point per
 * voxel: the reference's `--no-output-psf` inference, nesvor/nesvor/sample.py:17-33 -> models.py:154-174, and any tcnn-style
 * caller, models.py:25).  `layout` carries NESVOR_LAYOUT_UNCLUSTERED; the points are first put into the order of a coarse
 * lattice's cells (the counting sort of the unclustered backward: three small launches), the one-workgroup-per-256-points kernel
 * runs on workgroups of neighbouring points (lattice boxes in LDS up to the middle levels instead of 8 gathers per point and
 * level), and - feature-major output - the encoded rows are re-ordered into columns by one more launch.  Results are identical to
 * nesvor_hashgrid_forward's (same arithmetic per point).  `workspace`: nesvor_hashgrid_forward_workspace_bytes(grid, N, layout)
 * bytes of scratch (0 without the hint); N < 2^32. */
int64_t nesvor_hashgrid_forward_workspace_bytes(const nesvor_grid_t* grid, int64_t N, int layout);
int nesvor_hashgrid_forward_unclustered(const nesvor_grid_t* grid, const float* u, const float* table, float* pe, int64_t N,
                                        int layout, float* pe_absmax, void* workspace, int64_t workspace_bytes, void* stream);
/* grad_table is ACCUMULATED into (caller zero-fills when needed);
 * grad_u (N,3) is OVERWRITTEN, or NULL to skip the input gradient.
 * Owner-computes scatter (LDS aggregation per 256 samples -> per-chunk queues ->
 * one owner workgroup per table chunk); `workspace` is scratch device memory of
 * nesvor_hashgrid_backward_workspace_bytes(grid, N, queue_scale) bytes (-1: grid outside the
 * plan's limits, use the _atomic variant).  `stages`: 3 = whole backward;
 * 1 = aggregation launch only, 2 = owner launch only (so a caller can bracket
 * each launch with its own events; 1 then 2 on one stream == 3).
 * `queue_scale`: HOST array of one float per level in (0, 1], or NULL (= 1 everywhere): the fraction of the worst-case
 * queue capacity (8 records per point and level) to provide for each level.  A record that finds its queue full is added
 * with a global atomic instead - the result is exact for any scale - and counted: 32 uint32 counters (one per level) of
 * the latest backward sit at byte nesvor_hashgrid_backward_overflow_offset(workspace) of the workspace, so that a caller
 * can start small (a PSF-cloud batch fills a few percent of the worst case at most levels) and grow the levels that
 * overflow.  The same `queue_scale` must be passed to the size query and to every launch on that workspace. */
int64_t nesvor_hashgrid_backward_workspace_bytes(const nesvor_grid_t* grid, int64_t N, const float* queue_scale);
/* The same for a backward that will be called with this `layout` (hints included): larger than the above only for
 * NESVOR_LAYOUT_FEATURE_MAJOR | NESVOR_LAYOUT_UNCLUSTERED | NESVOR_LAYOUT_DY_SCRATCH. */
int64_t nesvor_hashgrid_backward_workspace_bytes_ex(const nesvor_grid_t* grid, int64_t N, const float* queue_scale, int layout);
/* The first nesvor_hashgrid_backward_workspace_zero_bytes() bytes of a workspace (its two queue-tail regions) must be
 * zero-filled ONCE after the workspace is allocated; the backward keeps them consistent afterwards (every aggregation
 * pass zero-fills the region the next backward will use, so no memset launch is needed per call). */
int64_t nesvor_hashgrid_backward_workspace_zero_bytes(void);
int64_t nesvor_hashgrid_backward_overflow_offset(void* workspace);
int nesvor_hashgrid_backward(const nesvor_grid_t* grid, const float* u, const float* table, const float* dpe,
                             float* grad_table, float* grad_u, int64_t N, int layout, void* workspace, int stages,
                             const float* queue_scale, void* stream);

/* The same backward restricted to levels [level_begin, level_end): a data-parallel step runs the fine levels first,
 * starts their all-reduce and overlaps it with the remaining levels.  Extra `stages` bits: 4 = do not reset the
 * queue tails and re-use the sample order the first launch sorted (every launch of one backward after the first: same
 * u, same N), 8 = add the input gradient of these levels to grad_u instead of overwriting it.  The launches of one
 * backward share `workspace`. */
int nesvor_hashgrid_backward_levels(const nesvor_grid_t* grid, const float* u, const float* table, const float* dpe,
                                    float* grad_table, float* grad_u, int64_t N, int layout, void* workspace, int stages,
                                    int level_begin, int level_end, const float* queue_scale, void* stream);
/* nesvor_hashgrid_backward_levels with a caller-supplied bound: dy_bound = device scalar >= max |dpe| over the batch (any
 * upper bound within ~2^20 of the true maximum keeps fp32 accuracy: the merge table sums in 64-bit fixed point scaled by
 * it), or NULL = the kernel determines each workgroup's maximum itself by reading dpe once more (134 MB at N = 2^20). */
int nesvor_hashgrid_backward_bounded(const nesvor_grid_t* grid, const float* u, const float* table, const float* dpe,
                                     float* grad_table, float* grad_u, int64_t N, int layout, void* workspace, int stages,
                                     int level_begin, int level_end, const float* queue_scale, const float* dy_bound,
                                     void* stream);
/* Same contract, tcnn-style per-corner global atomics (slow on MI355X: memory-side atomics). */
int nesvor_hashgrid_backward_atomic(const nesvor_grid_t* grid, const float* u, const float* table, const float* dpe,
                                    float* grad_table, float* grad_u, int64_t N, int layout, void* stream);

/* ------------------------------------------------------------------------
 * PSF sampling + rigid transform of a batch of slice pixels.  Replaces the head of
 * NeSVoR.forward (nesvor/nesvor/models.py:267-278), ax/mat_transform_points
 * (nesvor/transform/transform.py:259-280, trans_first = True) and the bounding-box
 * normalisation of INR.forward (models.py:143):
 *   x[b,s] = R_k (xyz[b] + noise[b,s] * sigma_k + t_k),  k = slice_idx[b];  u = (x - bb0)/(bb1 - bb0)
 * mat (n,3,4) per-slice [R|t]; slice_idx (B) int64; xyz (B,3); sigma (n,3); noise (B,S,3)
 * standard normal; bb (2,3); x (B,S,3) out; u (B*S,3) out or NULL.
 * backward: dx (B,S,3) and/or du (B*S,3) (either may be NULL) -> dmat (B,3,4) per PIXEL
 * (gradient w.r.t. the pixel's slice matrix; the caller index-adds pixels into slices).
 * ---------------------------------------------------------------------- */
int nesvor_psf_transform_forward(const float* mat, const int64_t* slice_idx, const float* xyz, const float* sigma,
                                 const float* noise, const float* bb, float* x, float* u, int B, int S, void* stream);
int nesvor_psf_transform_backward(const float* mat, const int64_t* slice_idx, const float* xyz, const float* sigma,
                                  const float* noise, const float* bb, const float* dx, const float* du, float* dmat,
                                  int B, int S, void* stream);
/* The same two operators with the PSF noise drawn inside the kernels: xi[b,s] = the three N(0,1) draws of a counter-based
 * generator (Philox4x32-10) keyed by `seed`, counter = (sample index b*S+s, `offset`) - nothing is stored, the backward
 * evaluates the same function.  The reference draws torch.randn (models.py:270); any N(0,1) stream is the same model.
 * x (and u) may be NULL when not needed.  nesvor_psf_noise writes the draws themselves, (n_samples, 3). */
int nesvor_psf_transform_forward_rng(const float* mat, const int64_t* slice_idx, const float* xyz, const float* sigma,
                                     uint64_t seed, uint64_t offset, const float* bb, float* x, float* u, int B, int S,
                                     void* stream);
/* The same launch also gathers se[b, :] = embedding[slice_idx[b], :] ((B, ks); ks = 0: nothing) - the per-pixel input of
 * sigma_net / b_net of the training step (the `slice_embedding(slice_idx)` lookup of nesvor/nesvor/models.py:281). */
int nesvor_psf_transform_forward_rng_gather(const float* mat, const int64_t* slice_idx, const float* xyz, const float* sigma,
                                            uint64_t seed, uint64_t offset, const float* bb, float* x, float* u, int B, int S,
                                            const float* embedding, float* se, int ks, void* stream);
int nesvor_psf_transform_backward_rng(const float* mat, const int64_t* slice_idx, const float* xyz, const float* sigma,
                                      uint64_t seed, uint64_t offset, const float* bb, const float* dx, const float* du,
                                      float* dmat, int B, int S, void* stream);
/* The same backward; every pixel's gradient is also (dpix non-NULL) or only (dpix NULL) ADDED to row slice_idx[b] of dmat_slice
 * (n,3,4) - the caller zero-fills it -, i.e. the index_add over slice_idx the reference's autograd performs, without a second
 * launch over dpix. */
int nesvor_psf_transform_backward_rng_slices(const float* mat, const int64_t* slice_idx, const float* xyz, const float* sigma,
                                             uint64_t seed, uint64_t offset, const float* bounding_box, const float* dx,
                                             const float* du, float* dpix, float* dmat_slice, int B, int S, void* stream);
int nesvor_psf_noise(uint64_t seed, uint64_t offset, float* out, int64_t n_samples, void* stream);

/* ------------------------------------------------------------------------
 * Fused small MLP (fp32 matrix cores).  Replaces the nn.Linear/ReLU stacks that
 * build_network creates in single-precision mode (nesvor/nesvor/models.py:42-67)
 * for density_net (:113-121), sigma_net (:238-246) and b_net (:249-258), plus the
 * expand/cat glue of NeSVoR.net_forward (:339-353).
 *   input  = [ xa[n / samples_per_pixel][0..k_a) | xb[b_row0 .. b_row0+k_b)[n] ]
 *            xa: (P, k_a) per-pixel features or NULL (k_a = 0); xb: (rows, N) feature-major
 *   layers = Linear(k_a+k_b, 64) ReLU [Linear(64,64) ReLU]*(n_hidden-1) Linear(64, out_dim)
 *            weight[l]: (out,in) row-major as in nn.Linear; bias[l]: (out)
 *   y      : (out_dim, N) feature-major
 * saved_hidden[l] (l < n_hidden): N_pad16*64 floats each, written by forward (pass
 * NULL for inference), read by backward.  dpre_scratch[l]: same size, scratch.
 * backward: dy (out_dim,N) -> dxa (N,k_a) per-sample (or NULL), dxb (k_b,N) (or NULL)
 * and dw_partial (n_partial, sum_l(out*in+out)) partial parameter gradients in
 * order W0,b0,W1,b1,.. that the caller sums over dim 0 (no atomics anywhere).
 * ---------------------------------------------------------------------- */
typedef struct {
  int32_t width;              /* hidden width; 64.  LIMIT: the fused kernels are built around one 64-column tile per layer (four
                                 16-feature blocks, weights resident in LDS as MFMA operand images).  The reference takes any
                                 --width / --depth (nesvor/cli/main.py:68-73, build_network models.py:42-67): widths below 64
                                 run zero-padded on these kernels (exactly the same function; nesvor_amd/mlp.py::kernel_params),
                                 depth up to NESVOR_MAX_MLP_LAYERS - 1 hidden layers is native, and anything wider or deeper is
                                 REFUSED here (hipErrorInvalidValue) and taken by nesvor_mlp_wide_t below (round 6: width <= 128,
                                 up to seven hidden layers, hand-written fp32-MFMA kernels; rounds 3-5 evaluated those shapes
                                 with rocBLAS under autograd).  tests/test_gpu_parity.py::test_other_widths_and_depths_match_oracle_losses
                                 holds 128 x 1, 64 x 4 and 96 x 5 to the oracle.  Only beyond THOSE limits does the host side
                                 (nesvor_amd/mlp.py::library_mlp) fall back to library GEMMs */
  int32_t n_hidden;           /* hidden layers, 1..NESVOR_MAX_MLP_LAYERS-1 */
  int32_t out_dim;            /* 1..16 */
  int32_t k_a, k_b, b_row0;
  int32_t samples_per_pixel;
  int32_t dxa_group_sums;     /* backward only: dxa is (N/16, k_a), one row per 16-sample group = the sum over its samples
                                 (needs N, samples_per_pixel and k_a multiples of 16 and the fused backward) */
  int32_t bf16_operands;      /* how the matrix products are evaluated (storage, accumulation, outputs are fp32 in all modes):
                                 0: fp32 MFMA (v_mfma_f32_16x16x4_f32, an fp32 FMA chain);
                                 1: operands (weights, activations, upstream gradients) rounded to bf16 - an opt-in
                                    mixed-precision mode, not the reference's fp32 semantics; the saved activations
                                    are bf16.  Backward: wave-specialised kernel only (N, S, k_a multiples of 16, at
                                    most two hidden layers, k_a + k_b <= 32 for two hidden layers);
                                 2: every fp32 operand x written as two fp16 numbers of a scaled copy (x s = hi + lo, both
                                    roundings to nearest, s a power of two per operand tensor and launch derived from the
                                    bounds in `prep`), three fp16 MFMAs per product, fp32 accumulation: fp32-equivalent accuracy
                                    (error against fp64 at or below the fp32 FMA chain's: tools/f16_split_probe.hip,
                                    tests/test_gpu_ops.py::test_fused_mlp_split_operands_keep_fp32_accuracy) on the 16x
                                    faster 16-bit matrix pipe.  Pipelined forward and wave-specialised backward (dX chain and dW
                                    products); every other kernel evaluates this mode as mode 0, which is always a valid
                                    evaluation of it.  (Rounds 2-4: three bf16 terms per operand, six MFMAs per product.)
                                    REQUIRES `prep`;
                                 3: operands rounded to fp16 (the reference's default arithmetic; overflows beyond 65504: its
                                    GradScaler's job), same kernels as mode 1;
                                 4: "scaled fp16" (round 6): mode 2's scales, operand images, bits-only save and per-sample
                                    backward scale with the LEADING term of every split alone - operands rounded to fp16 after
                                    scaling (no overflow, no loss scaler), ONE MFMA per product in the pipelined forward and the
                                    compact wave-specialised backward; every other kernel evaluates it as mode 2 or mode 0 (more
                                    accurate, always valid).  REQUIRES `prep`.  Measured: DESIGN.md section 4. */
  int32_t compact_save;       /* 1: training keeps, per hidden unit and sample, ONE BIT (h > 0) of every hidden layer and nothing
                                 else: saved_hidden[0] = one uint32 per (16-sample group, lane) - 16 N bytes instead of 256 N
                                 per hidden layer - with bit 16 l + 4 b + r = [pre-activation sign bit clear] (= [h_l > 0] for every
                                 value but an exact +0: the gate then passes a gradient the reference's ReLU'(0) = 0 drops - the
                                 convention of this mode, see tests/test_gpu_ops.py::test_fused_mlp_compact_save) for the unit the
                                 lane holds in block b, element r of the fragment layout; saved_hidden[l >= 1] are not
                                 touched (any non-NULL pointer).  The backward gates with the bits and RECOMPUTES every hidden
                                 layer from the network input (the forward's own products, 12 + 24 MFMAs per 16 samples at two
                                 hidden layers) instead of streaming it back: 536 MB less written and 670 MB less read per
                                 network and step at N = 2^20.  (Rounds 3-4 kept the values of the layers after the first.)
                                 Needs what nesvor_mlp_compact_save_ok() checks (split operands, the pipelined forward and
                                 the wave-specialised backward); forward and backward of one step must agree on it. */
  const float* weight[NESVOR_MAX_MLP_LAYERS];
  const float* bias[NESVOR_MAX_MLP_LAYERS];
  const float* prep;          /* mode 2: NESVOR_MLP_PREP_FLOATS device floats, valid bounds for THIS launch (a bound may be loose -
                                 it costs resolution below 2^-17 of it - but never too small: fp16 overflows at 65504 s).
                                 Operand bounds are SLOTTED: NESVOR_ABSMAX_FLOATS floats each, the bound is the maximum of every
                                 NESVOR_ABSMAX_STRIDE-th of them
                                 (publishing kernels spread their atomic maxima over the slots by workgroup):
                                   [NESVOR_MLP_PREP_XA ..] >= max |xa|, [NESVOR_MLP_PREP_XB ..] >= max |xb rows b_row0 .. b_row0 + k_b|,
                                   [NESVOR_MLP_PREP_DY ..] >= max |dy| (backward),
                                   [NESVOR_MLP_PREP_LAYER0 + 4 l + {0,1,2,3}] = max |W_l|, max_o sum_k |W_l[o][k]|,
                                   max_k sum_o |W_l[o][k]|, max |b_l|.
                                 nesvor_mlp_prepare() fills any part of it; inside the training step the producing kernels publish
                                 the operand bounds (hash-grid forward: max |pe|; the networks' y_absmax / dxb_absmax; the loss
                                 kernel: max |d z|, max |d log_var|) and one launch per step takes the weight norms.  Forward
                                 and backward of one step must see the same [0], [1] and weights. */
  float* y_absmax;            /* forward, optional: a slotted bound (NESVOR_ABSMAX_FLOATS zero-filled device floats) raised to max |y| */
  const void* weight_images;  /* mode 2, optional (NULL: every launch builds its LDS operand images from `weight`, as rounds 2-5 did):
                                 nesvor_mlp_weight_images_bytes(net) device bytes holding the split fp16 operand images of ALL layers
                                 for the CURRENT weights - written by nesvor_mlp_prepare_weights_images() together with the weight
                                 norms, so that the images' scales are the ones `prep` yields.  The training step (csrc/step.hip)
                                 builds them once per iteration for its four MLP launches (round 6; measured: -3.4 us of a 0.284 ms
                                 step at 2^17 points, within noise at 2^20).  Bit-identical to the in-kernel builds
                                 (tests/test_gpu_ops.py::test_fused_mlp_prebuilt_weight_images_are_bit_identical). */
} nesvor_mlp_t;
#define NESVOR_ABSMAX_SLOTS 16      /* a slotted bound: this many floats, NESVOR_ABSMAX_STRIDE floats (one 256-byte line) apart - */
#define NESVOR_ABSMAX_STRIDE 64     /* atomics on one cache line serialise at the memory side, publishers spread by workgroup */
#define NESVOR_ABSMAX_FLOATS (NESVOR_ABSMAX_SLOTS * NESVOR_ABSMAX_STRIDE)
#define NESVOR_MLP_PREP_XA 0
#define NESVOR_MLP_PREP_XB (1 * NESVOR_ABSMAX_FLOATS)
#define NESVOR_MLP_PREP_DY (2 * NESVOR_ABSMAX_FLOATS)
#define NESVOR_MLP_PREP_LAYER0 (3 * NESVOR_ABSMAX_FLOATS)
#define NESVOR_MLP_PREP_FLOATS (NESVOR_MLP_PREP_LAYER0 + 4 * NESVOR_MAX_MLP_LAYERS)
#define NESVOR_MLP_WHAT_INPUT 1   /* `what` of nesvor_mlp_prepare: the xa and xb bounds */
#define NESVOR_MLP_WHAT_DY 2      /* the dy bound */
#define NESVOR_MLP_WHAT_WEIGHTS 4 /* the per-layer norms from net->weight / net->bias */
/* Fill (parts of) `prep` for `net` on `stream`: absolute maxima by a grid-wide reduction (the selected entries are zero-filled
 * first), weight norms by one workgroup per layer.  xa / xb / dy may be NULL when their bit is not set. */
int nesvor_mlp_prepare(const nesvor_mlp_t* net, const float* xa, const float* xb, const float* dy, int64_t N, float* prep,
                       int what, void* stream);
/* The weight norms of up to three networks in ONE launch (the training step: once per iteration, on its side stream);
 * optionally also max |x[0..n_x)| into the slotted bound `x_slots` (the slice embedding table: the pixel features of
 * sigma_net / b_net are rows of it).  preps[i] as nesvor_mlp_t.prep of nets[i]; nothing is zero-filled. */
int nesvor_mlp_prepare_weights(const nesvor_mlp_t* const* nets, float* const* preps, int n_nets, const float* x, int64_t n_x,
                               float* x_slots, void* stream);

/* ... the same launch also writing each network's split operand images (nesvor_mlp_t.weight_images): images[i] = NULL or
 * nesvor_mlp_weight_images_bytes(nets[i]) device bytes (16-byte aligned); `images` itself may be NULL. */
int nesvor_mlp_prepare_weights_images(const nesvor_mlp_t* const* nets, float* const* preps, void* const* images, int n_nets,
                                      const float* x, int64_t n_x, float* x_slots, void* stream);
/* Size of a network's prebuilt operand images (0: a shape the split-mode kernels do not take). */
int64_t nesvor_mlp_weight_images_bytes(const nesvor_mlp_t* net);

/* 1 if (net, N) can run with compact_save = 1 (the field itself is ignored by this query), else 0. */
int nesvor_mlp_compact_save_ok(const nesvor_mlp_t* net, int64_t N);

int nesvor_mlp_forward(const nesvor_mlp_t* net, const float* xa, const float* xb, float* y,
                       float* const* saved_hidden, int64_t N, void* stream);
/* 1 if (net, N) is a shape the fused backward takes - dX, dW and db in ONE wave-specialised launch, no dpre scratch (pass NULL
 * entries): N, samples_per_pixel and k_a multiples of 16, at most two hidden layers, at most two 16-row input blocks at two
 * hidden layers.  0: pass dpre_scratch[l] (N_pad16 * 64 floats each) and the backward runs as a dX launch + a dW launch. */
int nesvor_mlp_backward_fused_ok(const nesvor_mlp_t* net, int64_t N);
int nesvor_mlp_backward(const nesvor_mlp_t* net, const float* xa, const float* xb, const float* dy,
                        float* const* saved_hidden, float* const* dpre_scratch, float* dxa, float* dxb,
                        float* dw_partial, int n_partial, int64_t N, void* stream);
/* The same backward; additionally raises the device scalar *dxb_absmax (atomic max; the caller zero-fills it) to max |dxb|
 * when dxb_absmax and dxb are non-NULL: the consumer of dxb - the hash-grid backward, nesvor_hashgrid_backward_bounded -
 * then needs no pass of its own over the gradient to scale its fixed-point sums. */
int nesvor_mlp_backward_bounded(const nesvor_mlp_t* net, const float* xa, const float* xb, const float* dy,
                                float* const* saved_hidden, float* const* dpre_scratch, float* dxa, float* dxb,
                                float* dw_partial, int n_partial, int64_t N, float* dxb_absmax, void* stream);

/* ------------------------------------------------------------------------
 * The same networks at ANY width up to 128 and up to NESVOR_MLP_WIDE_MAX_LAYERS - 1 hidden layers (round 6; csrc/mlp_wide.hip):
 * the reference builds its MLPs with any --width / --depth (nesvor/cli/main.py:68-73 -> build_network,
 * nesvor/nesvor/models.py:42-67; bias-free tinycudann.Network in half precision, models.py:28-41), and shapes outside
 * nesvor_mlp_t's (width 64, at most three hidden layers) used to leave the hand-written path for library GEMMs.  fp32 matrix
 * cores (v_mfma_f32_16x16x4_f32: an fp32 FMA chain), activations in registers from input to output, ONE layer's weights in LDS
 * at a time (a 128 x 128 layer is 64 KB: the workgroup swaps the operand image between layers of a 256-sample tile).  Widths
 * that are not multiples of 16 run zero-padded (exactly the same function).  Input composition, layouts and the meaning of
 * xa / xb / b_row0 / k_a / k_b / samples_per_pixel as for nesvor_mlp_t; k_a + k_b <= 64, out_dim <= 16.
 *   forward : y (out_dim, N); saved_hidden = n_hidden device buffers of nesvor_mlp_wide_saved_floats(net, N) floats each (the
 *             post-ReLU activations in MFMA fragment layout) or NULL (inference);
 *   backward: dy (out_dim, N) -> dxa (N, k_a) per SAMPLE (or NULL), dxb (k_b, N) (or NULL), dw_partial (n_partial,
 *             nesvor_mlp_wide_param_count(net)): per-workgroup partial sums in nn.Linear parameter order W0, b0, W1, b1, ...
 *             (no b columns for a NULL bias), to be summed over the rows by the caller; dpre_scratch = n_hidden buffers of the
 *             saved buffers' size.  Two launches (dX chain, then dW / db), no atomics.
 * Errors: hipErrorInvalidValue for shapes outside the limits or missing buffers. */
#define NESVOR_MLP_WIDE_MAX_LAYERS 8
typedef struct {
  int32_t width;              /* hidden width, 1..128 */
  int32_t n_hidden;           /* 1..NESVOR_MLP_WIDE_MAX_LAYERS-1 */
  int32_t out_dim;            /* 1..16 */
  int32_t k_a, k_b, b_row0;
  int32_t samples_per_pixel;
  int32_t reserved;
  const float* weight[NESVOR_MLP_WIDE_MAX_LAYERS];  /* nn.Linear layout (out, in), fp32 */
  const float* bias[NESVOR_MLP_WIDE_MAX_LAYERS];    /* (out) or NULL: bias-free layer */
} nesvor_mlp_wide_t;
int64_t nesvor_mlp_wide_saved_floats(const nesvor_mlp_wide_t* net, int64_t N);
int nesvor_mlp_wide_param_count(const nesvor_mlp_wide_t* net);
int nesvor_mlp_wide_forward(const nesvor_mlp_wide_t* net, const float* xa, const float* xb, float* y, float* const* saved_hidden,
                            int64_t N, void* stream);
int nesvor_mlp_wide_backward(const nesvor_mlp_wide_t* net, const float* xa, const float* xb, const float* dy,
                             float* const* saved_hidden, float* const* dpre_scratch, float* dxa, float* dxb, float* dw_partial,
                             int n_partial, int64_t N, void* stream);
/* ... additionally raising the device scalar *dxb_absmax to max |dxb| (as nesvor_mlp_backward_bounded does; NULL: not wanted).  At
 * width 64 the saved / dpre buffers have the fragment layout of nesvor_mlp_t's full save: nesvor_mlp_backward_bounded hands the
 * shapes its wave-specialised kernel does not take (ragged N, samples per pixel or pixel features not in multiples of 16, three
 * hidden layers) to this entry point - round 6 retired mlp.hip's own dX / dW launch pair. */
int nesvor_mlp_wide_backward_bounded(const nesvor_mlp_wide_t* net, const float* xa, const float* xb, const float* dy,
                                     float* const* saved_hidden, float* const* dpre_scratch, float* dxa, float* dxb,
                                     float* dw_partial, int n_partial, int64_t N, float* dxb_absmax, void* stream);

/* ------------------------------------------------------------------------
 * Imaging model + losses, value and gradient in one launch.  Replaces the tail of
 * NeSVoR.forward (nesvor/nesvor/models.py:286-325), edge_reg/tv_reg/l2_reg
 * (models.py:366-384) and their autograd backward.  See csrc/loss.hip for the math.
 * Inputs (device): z0, log_var, log_bias (B*S) [log_var / log_bias may be NULL],
 * x (B,S,3), v (B), slice_idx (B) int64, c (n) slice scale or NULL, log_var_slice (n) or
 * NULL, log_bias_mean (1) (only read when log_bias != NULL).
 * Forward launch (gw == NULL) writes loss_pix (B,3) = per-pixel [MSE term, logVar term,
 * sum_s regulariser term].  Backward launch (gw = device pointer to the 4 upstream
 * gradients d total / d {MSE, logVar, imageReg, biasReg}) writes dz0, dlog_var, dlog_bias
 * (B*S), dx (B,S,3) or NULL, dc_pix (B) or NULL, dlvs_pix (B) or NULL; if loss_pix is
 * non-NULL as well, the same launch also writes the loss values.
 * reg_type: 0 edge, 1 TV, 2 L2.
 * ---------------------------------------------------------------------- */
typedef struct {
  const float* z0; const float* log_var; const float* log_bias; const float* x; const float* v;
  const int64_t* slice_idx; const float* c; const float* log_var_slice; const float* log_bias_mean;
  const float* gw;
  float* loss_pix; float* dz0; float* dlog_var; float* dlog_bias; float* dx; float* dc_pix; float* dlvs_pix;
  int32_t B, S, reg_type;
  float delta;
  /* optional (backward launch): slotted bounds (NESVOR_ABSMAX_FLOATS floats each, zero-filled by the caller) raised to max |dz0|,
     max |dlog_var|, max |dlog_bias| - the upstream-gradient bounds of the networks that consume them
     (nesvor_mlp_t.prep + NESVOR_MLP_PREP_DY) */
  float* dz0_absmax; float* dlog_var_absmax; float* dlog_bias_absmax;
} nesvor_loss_t;

int nesvor_imaging_loss(const nesvor_loss_t* args, void* stream);

/* Per-pixel -> per-slice gradient accumulation of one training iteration (the index_add /
 * embedding-backward / sum-over-samples steps autograd runs for models.py:267-325):
 *   dc[k] += dc_pix[b]; dlvs[k] += dlvs_pix[b]; dmat[k,:] += dpix[b,:] (12 floats);
 *   dse[k,:] += sum_s dxa[b,s,:]   with k = slice_idx[b].
 * dxa (B*S, ks) per-sample gradient w.r.t. the slice embedding fed to an MLP.  Any of the four
 * source pointers may be NULL (skipped).  Outputs are ACCUMULATED into (caller zero-fills). */
/* Small-tensor bookkeeping of one training iteration (nesvor_amd/direct.py), one launch each:
 *   prologue: c (n) = n softmax(logit_coef) (models.py:296; logit_coef may be NULL), mat (n,3,4) =
 *             axisangle2mat(axisangle) (axisangle may be NULL), zero_buf[0..n_zero) = 0;
 *   epilogue: dlogit (n) = c (dc - <dc,c>/n) (dc may be NULL); daxisangle (n,6) = axisangle2mat_backward(dmat,
 *             axisangle) + w_trans * dtrans and losses[3] = sum(trans_terms) (dmat may be NULL);
 *             losses[0,1,2,4] = {MSE, logVar, MSE+logVar, imageReg} from loss_pix (B,3) (nesvor_imaging_loss):
 *             imageReg = img_scale * sum(loss_pix[:,2]) + img_offset. */
int nesvor_step_prologue(const float* logit_coef, float* c, const float* axisangle, float* mat, float* zero_buf,
                         int n_zero, int n, void* stream);
/* The same launch with one more workgroup: NeSVoR.trans_loss (nesvor_trans_loss's arithmetic) for the n slices -
 * trans_terms (n) and g_trans (n,6) are overwritten.  axisangle_init NULL: exactly nesvor_step_prologue. */
int nesvor_step_prologue_pose(const float* logit_coef, float* c, const float* axisangle, float* mat, float* zero_buf,
                              int n_zero, int n, const float* axisangle_init, float* trans_terms, float* g_trans, void* stream);
int nesvor_step_epilogue(const float* dc, const float* c, float* dlogit, const float* dmat, const float* axisangle,
                         const float* dtrans, float w_trans, float* daxisangle, const float* loss_pix,
                         const float* trans_terms, float* losses, int n, int B, float img_scale, float img_offset,
                         void* stream);
int nesvor_slice_grads(const int64_t* slice_idx, const float* dc_pix, const float* dlvs_pix, const float* dxa,
                       const float* dpix, float* dc, float* dlvs, float* dse, float* dmat, int B, int S, int ks,
                       void* stream);
/* The same sums without atomics: one workgroup per slice lists the batch's pixels of its slice (in batch order: reproducible
 * sums) and adds their totals to the outputs it alone owns.  Needs B <= 4096 and 256 % ks == 0 (or dxa == NULL); returns
 * hipErrorInvalidValue (= 1) otherwise, and the caller falls back to nesvor_slice_grads. */
int nesvor_slice_grads_by_slice(const int64_t* slice_idx, const float* dc_pix, const float* dlvs_pix, const float* dxa,
                                const float* dpix, float* dc, float* dlvs, float* dse, float* dmat, int B, int S, int ks,
                                int n_slices, void* stream);

/* ------------------------------------------------------------------------
 * Fused AdamW over a flat fp32 parameter buffer.  Replaces the
 * torch.optim.AdamW step at nesvor/nesvor/train.py:144-152,195-197
 * (betas (0.9,0.99), eps 1e-15, decoupled weight decay):
 *   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
 *   p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
 * grad is multiplied by grad_scale first (DDP mean) and, if zero_grad != 0,
 * zero-filled afterwards (fuses optimizer.zero_grad()).
 * ---------------------------------------------------------------------- */
int nesvor_adamw_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay,
                      float bias_correction1, float bias_correction2, float grad_scale, int zero_grad,
                      void* stream);

/* The same numbers as one struct (bias_correction{1,2} = 1 - beta{1,2}^t of the step being taken). */
typedef struct {
  float lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2, grad_scale;
} nesvor_adamw_t;

/* Hash-grid backward whose owner pass also takes the AdamW step on the table: the workgroup that completes a chunk's
 * gradient updates table / exp_avg / exp_avg_sq of that chunk while the gradient is still in LDS, so the table gradient
 * never travels through HBM.  Equals nesvor_hashgrid_backward_bounded(stages) over all levels followed by
 * nesvor_adamw_step(table, grad_table, exp_avg, exp_avg_sq, <table numel>, ..., zero_grad = 1): grad_table is read (it may
 * hold an earlier backward's gradient) and is all zero afterwards, but the gradient of THIS backward is never stored in it.
 * stages: 1 = aggregation pass, 2 = owner pass + AdamW, 3 = both (as nesvor_hashgrid_backward; no level ranges).
 * table, exp_avg, exp_avg_sq, grad_table: (sum of level sizes, F), the layout nesvor_hashgrid_forward reads. */
int nesvor_hashgrid_backward_adamw(const nesvor_grid_t* grid, const float* u, float* table, const float* dpe,
                                   float* grad_table, float* grad_u, int64_t N, int layout, void* workspace, int stages,
                                   const float* queue_scale, const float* dy_bound, float* exp_avg, float* exp_avg_sq,
                                   const nesvor_adamw_t* adam, void* stream);
/* The owner stage (stages == 2) of that backward for levels [level_begin, level_end) only, and the per-cloud forward for a level
 * range (round 6).  The table's update is what the NEXT iteration's forward waits for; updated range by range - the coarse
 * levels first - the next forward starts on the finished levels while the owner pass still works on the others
 * (csrc/step.hip: 84 us of owner pass + 70 us of forward, both latency-bound, no longer strictly one after the other).
 * nesvor_hashgrid_forward_levels: the NESVOR_LAYOUT_CLUSTERED kernel; rows of pe outside the range are not touched. */
int nesvor_hashgrid_backward_adamw_levels(const nesvor_grid_t* grid, const float* u, float* table, const float* dpe,
                                          float* grad_table, float* grad_u, int64_t N, int layout, void* workspace, int stages,
                                          int level_begin, int level_end, const float* queue_scale, const float* dy_bound,
                                          float* exp_avg, float* exp_avg_sq, const nesvor_adamw_t* adam, void* stream);
int nesvor_hashgrid_forward_levels(const nesvor_grid_t* grid, const float* u, const float* table, float* pe, int64_t N, int layout,
                                   float* pe_absmax, int level_begin, int level_end, void* stream);

/* out[c] = sum_r in[r * ld + c], c < cols, for a row-major matrix of row pitch ld >= cols floats: reduces the
 * dw_partial of nesvor_mlp_backward (the `partial.sum(0)` of the host side) straight into a gradient segment; with
 * ld > cols, a column range of it (one layer's weights when the model keeps no biases, tinycudann.Network). */
int nesvor_sum_rows(const float* in, float* out, int rows, int cols, int ld, void* stream);
/* n_jobs <= 4 of those sums (same number of rows) in one launch; in / out / cols / ld: host arrays of n_jobs entries. */
int nesvor_sum_rows_multi(const float* const* in, float* const* out, const int* cols, const int* ld, int n_jobs, int rows,
                          void* stream);

/* ------------------------------------------------------------------------
 * One training iteration behind one entry point (csrc/step.hip).  Replaces the loop body of the reference's train()
 * (nesvor/nesvor/train.py:179-197: NeSVoR.forward models.py:260-327, loss.backward(), optimizer.step(), zero_grad()):
 * the host enqueues every launch of the iteration - sampler, hash grid, MLPs, imaging loss, the backwards, per-slice
 * gradients, AdamW - with one call, into workspace buffers the caller allocated once for (B, S).  All pointers are device
 * pointers unless stated; gradients go straight into the caller's (flat) gradient buffer, which the AdamW step zero-fills.
 *   switches   : opt_T = !no_transformation_optimization, has_lv = !no_pixel_variance, has_c = !no_slice_scale,
 *                has_lvs = !no_slice_variance, has_b = n_levels_bias > 0 (cli/main.py:61,86-110)
 *   small      : n (1 + 12 + 13) + 1 + 3 NESVOR_MLP_PREP_FLOATS floats: slice scale c | pose matrices | zeroed accumulators
 *                [dc | dmat] | max |dpe| | the three networks' operand bounds and weight norms (nesvor_mlp_t.prep; the step
 *                points density / sigma / bias_net at them itself)
 *   saved_*    : per hidden layer N_pad16 * 64 floats (nesvor_mlp_forward; slot 0 of a network with compact_save: 4 N_pad16
 *                floats);  partial: 3 x NESVOR_STEP_MLP_PARTIALS x (largest network's parameter count) floats - one third per
 *                network, summed by one nesvor_sum_rows_multi launch;  losses (run argument): 6 floats {MSE, logVar, MSE+logVar, transReg,
 *                imageReg, biasReg} (models.py:14-19)
 *   queue_scale: HOST array (nesvor_hashgrid_backward);  side_stream: a second stream of the same device (pose regulariser,
 *                owner pass of the hash-grid backward); with overlap_owner the table gradient is complete only behind the
 *                side stream - the step joins it itself before its own AdamW, a caller that runs the optimizer makes
 *                its stream wait for side_stream
 * nesvor_step_run(phase = 0): the whole iteration.  Data parallel: phase 1 = everything up to and including the hash-grid
 * backward of levels [split_level, L) (the host then starts the all-reduce of that part of the table gradient), phase 2 =
 * the remaining levels and the rest of the iteration.  adam == NULL: no optimizer step (the caller reduces gradients first).
 * The PSF noise is drawn inside the sampler kernels from (seed, offset) as in nesvor_psf_transform_forward_rng.
 * ---------------------------------------------------------------------- */
#define NESVOR_STEP_MLP_PARTIALS 256
/* nesvor_step_timing: spans of a phase-0 run that are bracketed by HIP events on the stream their launch goes to */
#define NESVOR_STEP_SPAN_PSF_FWD 0
#define NESVOR_STEP_SPAN_HASHGRID_FWD 1
#define NESVOR_STEP_SPAN_MLP_FWD_DENSITY 2
#define NESVOR_STEP_SPAN_MLP_FWD_SIGMA 3
#define NESVOR_STEP_SPAN_LOSS 4
#define NESVOR_STEP_SPAN_MLP_BWD_SIGMA 5
#define NESVOR_STEP_SPAN_MLP_BWD_DENSITY 6
#define NESVOR_STEP_SPAN_HASHGRID_BWD_AGGREGATE 7
#define NESVOR_STEP_SPAN_HASHGRID_BWD_OWNER 8   /* the owner pass; with the fused optimizer: owner pass + the table's AdamW step */
#define NESVOR_STEP_SPAN_PSF_BWD 9
#define NESVOR_STEP_SPAN_HASHGRID_FWD_LATE 10   /* pipelined table update (round 6): the forward of the levels the previous step's owner pass
                                                   finished LAST; NESVOR_STEP_SPAN_HASHGRID_FWD then covers the other levels only */
#define NESVOR_STEP_SPAN_HASHGRID_UNION 11      /* pipelined table update: from the start of the PREVIOUS step's owner pass to the end of this
                                                   step's hash-grid forward - the time the device spends on those overlapped launches */
#define NESVOR_STEP_TIMED_SPANS 12
typedef struct {
  nesvor_grid_t grid;
  nesvor_mlp_t density, sigma, bias_net;   /* weights / biases: the model's parameters; bf16_operands: evaluation mode */
  int32_t B, S, n_slices;
  int32_t opt_T, has_lv, has_c, has_lvs, has_b;
  int32_t n_features_z, ks, kb_bias;       /* ks: slice-embedding width fed to sigma_net / b_net (0: none); kb_bias: rows of pe b_net sees */
  int32_t reg_type;                        /* 0 edge, 1 TV, 2 L2 */
  int32_t overlap_owner;                   /* bit 0: owner pass of the hash-grid backward on side_stream; bit 1: a phase-0 run with
                                              `adam` takes the table's AdamW step inside the owner pass (nesvor_hashgrid_backward_adamw;
                                              the table must be the last segment of the flat buffers) */
  float delta, w_T;
  const float *axisangle, *axisangle_init, *psf_sigma, *bounding_box, *logit_coef, *log_var_slice, *slice_embedding, *table;
  float *g_axisangle, *g_logit_coef, *g_log_var_slice, *g_slice_embedding, *g_table, *g_density, *g_sigma, *g_bias_net;
  int32_t n_density_params, n_sigma_params, n_bias_params;
  const float* gw;                         /* 4 upstream gradients d total / d {MSE, logVar, imageReg, biasReg} */
  float *flat_param, *flat_grad, *flat_exp_avg, *flat_exp_avg_sq;
  int64_t flat_numel;
  float *small, *x, *u, *pe, *z, *log_var, *log_bias, *se, *dz, *dlv, *dlb, *dxl, *loss_pix, *pix, *dpe, *dpe_b, *du, *dpix,
      *dxa, *dxa_b, *trans_terms, *g_trans, *lb_mean, *mean_scratch, *partial;
  float* saved_d[NESVOR_MAX_MLP_LAYERS];
  float* saved_s[NESVOR_MAX_MLP_LAYERS];
  float* saved_b[NESVOR_MAX_MLP_LAYERS];
  float* dpre_scratch[NESVOR_MAX_MLP_LAYERS]; /* N_pad16 * 64 floats per hidden layer, shared by the networks' backwards one after the
                                               other; only read for networks nesvor_mlp_backward_fused_ok() refuses (samples per pixel
                                               or pixel features not in multiples of 16 ...): may be NULL when it takes them all */
  void* hg_workspace;
  const float* queue_scale;
  void* side_stream;
} nesvor_step_t;

void* nesvor_step_create(const nesvor_step_t* desc);                 /* NULL on failure */
int nesvor_step_update(void* step, const nesvor_step_t* desc);       /* pointers / sizes changed (e.g. a re-made workspace) */
void nesvor_step_destroy(void* step);
int nesvor_step_run(void* step, const float* xyz, const float* v, const int64_t* slice_idx, uint64_t seed, uint64_t offset,
                    float* losses, int phase, int split_level, const nesvor_adamw_t* adam, void* stream);
/* OR-ed into `phase` (phase 0 with `adam` and overlap_owner bits 0 and 1): return without making `stream` wait for the table's
 * update on side_stream.  The next nesvor_step_run joins it right before its hash-grid forward (its prologue and sampler
 * then overlap the end of this update); anything else that touches the table, its moments or its gradient first calls
 * nesvor_step_join.  Everything but the table (losses, the other parameters) is complete on `stream` as always. */
#define NESVOR_STEP_DEFER_JOIN 8
int nesvor_step_join(void* step, void* stream);
/* Per-launch timing of the product step (the roofline leg of bench.py): nesvor_step_timing(handle, 1) makes every following
 * nesvor_step_run bracket the launches listed above with HIP events on the stream each goes to; nesvor_step_timing_read waits
 * for the last run and returns NESVOR_STEP_TIMED_SPANS durations in ms (-1: the span did not occur in that run). */
int nesvor_step_timing(void* handle, int on);
int nesvor_step_timing_read(void* handle, float* ms);

/* ----------------------------------------------------------------------
 * Similarity sums of the stack registration.  Replaces, per optimisation step of `VVR`
 * (nesvor/svort/registration.py:143-247), the K = 1 + 2 x 6 successive evaluations of
 *   warped = F.grid_sample(source, (R_k (points + t_k)) * to_unit, align_corners=True)   (:233-247)
 *   loss(warped, target) reduced over the points                                          (:166-170)
 * by one pass over the points: for every pose k
 *   sums[k] = { sum I, sum I^2, sum I J },  I = source sampled under pose k, J = target;
 *   target_sums = { sum J, sum J^2 } (optional)
 * from which NCC and MSE follow.  source (D,H,W) fp32, points (M,3) physical coordinates, target (M),
 * mats (K,3,4) row-major [R | t] with the translation applied first, to_unit_xyz: HOST pointer to the three factors
 * that map a moved point to grid_sample's normalised coordinates (x, y, z).  sums (K,3) / target_sums (2): fp64,
 * zero-filled by the call.
 * ---------------------------------------------------------------------- */
int nesvor_vvr_similarity(const float* source, int D, int H, int W, const float* points, const float* target,
                          const float* mats, const float* to_unit_xyz, int64_t M, int K, double* sums,
                          double* target_sums, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NESVOR_HIP_H */
