/*
 * nsr.h -- C ABI of libnsr.so, the MI355X-native NICE-SLAM render hot path.
 *
 * Every entry point replaces a piece of the reference's *Python* interface (the reference has no
 * native code).  Citations are relative to the reference tree:
 *
 *   nsr_get_samples      <- src/common.py:125-134  get_samples (after the torch.randint draw)
 *   nsr_aabb_keep        <- src/Mapper.py:471-481, src/Tracker.py:95-104  bounding-box pre-filter (mask, no compaction)
 *   nsr_pack_params      <- (none) operand re-layout of src/conv_onet/models/decoder.py parameters
 *   nsr_render_fwd       <- src/utils/Renderer.py:63-198  Renderer.render_batch_ray  (forward)
 *                           incl. eval_points (:23-61), NICE.forward (decoder.py:312-342),
 *                           raw2outputs_nerf_color (src/common.py:204-245)
 *   nsr_render_bwd       <- the autograd backward of the above (src/Mapper.py:503, src/Tracker.py:125)
 *   nsr_eval_points_fwd  <- src/utils/Renderer.py:23-61   Renderer.eval_points (forward only)
 *   nsr_masked_adam      <- src/Mapper.py:368-379,394-401,504,511-519  masked write-back + Adam on one feature grid
 *   nsr_masked_adam_multi <- the same for all grids of a stage, step counts on the device (capturable)
 *   nsr_flat_adam        <- src/Mapper.py:368-387,504, src/Tracker.py:214-222,127  torch.optim.Adam on the dense rest of the
 *                           callers' optimiser (decoder parameter blobs, camera tensors), one launch pair, capturable
 *   nsr_get_samples_window <- src/Mapper.py:437-481  sampling loop over the mapping window + bounding-box pre-filter
 *   nsr_get_samples_window_fused <- the same as the first launch of a fused iteration: + the pixel draw of src/common.py:99 and the
 *                           zero fill that `loss.backward()` (src/Mapper.py:503) relies on, inside the one launch (ABI 7)
 *   nsr_get_samples_window_sharded <- the same for one rank of a ray-sharded iteration: + the batch-global max(gt_depth) of
 *                           src/utils/Renderer.py:109,144 over ALL ranks' draws, without a collective (ABI 8)
 *   nsr_pose_grad        <- autograd of src/common.py:74-88 for that window (local BA, src/Mapper.py:417-419)
 *   nsr_pack_rows        <- (none) gather / scatter of the voxel rows + blobs that travel in the multi-GPU all-reduce
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller (PyTorch); the library never frees or
 *     retains them beyond the call.  No torch types cross this boundary.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises.
 *   - every function returns 0 on success; on failure a non-zero code and nsr_last_error()
 *     (thread-local, valid until the next failing call on that thread) describes it.
 *   - feature grids are fp32, 32 channels, CHANNELS-LAST: element (z,y,x,c) at ((z*Y+y)*X+x)*32+c.
 *     This is the physical layout of a torch tensor of logical shape [1,32,Z,Y,X] in
 *     torch.channels_last_3d memory format (reference logical shape: src/NICE_SLAM.py:218-222).
 *   - decoder parameters are ONE flat fp32 blob per decoder in the order of the reference module's
 *     named_parameters() (decoder.py:124-159 / :235-245), see nsr_param_count().
 */
#ifndef NSR_H_
#define NSR_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSR_VERSION 8

/* stages of NICE.forward (decoder.py:312-342) */
enum { NSR_STAGE_COARSE = 0, NSR_STAGE_MIDDLE = 1, NSR_STAGE_FINE = 2, NSR_STAGE_COLOR = 3 };
/* decoder / grid slots */
enum { NSR_COARSE = 0, NSR_MIDDLE = 1, NSR_FINE = 2, NSR_COLOR = 3 };

#define NSR_MAX_SAMPLES 64

typedef struct nsr_grid {
    const float *feat;   /* [Z][Y][X][32] */
    float *dfeat;        /* gradient accumulator, same layout, caller-zeroed; NULL = not needed */
    int32_t Z, Y, X;     /* Z * Y * X < 2^25 voxels (a grid is addressed with 32-bit byte offsets: 4 GB)   */
    int32_t pad_;
    double lo[3];        /* normalisation box of the decoder reading this grid, xyz order          */
    double hi[3];        /* (decoder .bound, src/NICE_SLAM.py:152-157; coarse = scene bound * 2)   */
} nsr_grid;

typedef struct nsr_decoder {
    const float *params; /* flat blob, nsr_param_count(slot) floats                                */
    const float *packed; /* MFMA operand stream written by nsr_pack_params, nsr_packed_count(slot) */
    float *dparams;      /* flat gradient blob (caller-zeroed) or NULL                             */
} nsr_decoder;

typedef struct nsr_render_args {
    int32_t stage;            /* NSR_STAGE_*                                                        */
    int32_t n_samples;        /* uniform samples per ray  (cfg rendering.N_samples)                 */
    int32_t n_surface;        /* near-surface samples; forced to 0 when gt_depth == NULL or coarse  */
    int32_t pad_;
    int64_t n_rays;
    const float *rays_o;      /* [N][3] */
    const float *rays_d;      /* [N][3] */
    const float *gt_depth;    /* [N] or NULL (Renderer.py:88-96)                                    */
    const float *gt_max;      /* device scalar: max over the WHOLE batch of gt_depth (Renderer.py:109,144);
                                 required when gt_depth != NULL.  Kept on device: no host sync.     */
    double bound_lo[3];       /* Renderer.bound (un-enlarged): far_bb and the in-bound test         */
    double bound_hi[3];
    float t_uniform[NSR_MAX_SAMPLES];   /* torch.linspace(0,1,n_samples) fp32 (Renderer.py:152)     */
    double t_surface[NSR_MAX_SAMPLES];  /* torch.linspace(0,1,n_surface).double() (Renderer.py:132) */
    nsr_grid grid[4];         /* slots NSR_COARSE..NSR_COLOR; unused slots may be zeroed            */
    nsr_decoder dec[4];
    /* outputs */
    double *depth;            /* [N]    */
    double *var;              /* [N]    */
    float *rgb;               /* [N][3] */
    float *raw;               /* [N][S][4] decoder output after the out-of-bound override, S = n_samples+n_surface;
                                 written by fwd, read by bwd.  May be NULL for a forward-only call.  */
    double *zvals;            /* [N][S] sorted sample depths, written by fwd when non-NULL; required (like acts and raw) by a call
                                 that will be differentiated                                                            */
    /* --- fused mapping loss (src/Mapper.py:487-493), optional: all NULL / 0 for the plain renderer ------------------
     * fwd with loss != NULL adds   sum over rays r with keep[r] of  [gt_depth[r] > 0] |gt_depth[r] - depth[r]|   (gt_depth
     *                              as passed in this block, also in the coarse stage whose SAMPLING ignores it)
     *                              + (colour stage) w_color * sum_c |gt_color[r][c] - rgb[r][c]|     to *loss (fp64)
     * and writes that sum's derivative w.r.t. each ray's outputs to dl_depth / dl_rgb -- exactly the arrays
     * nsr_render_bwd takes as d_depth / d_rgb (d_var = NULL), so the caller's loss and its backward cost no launch. */
    const float *gt_color;    /* [N][3] */
    const uint8_t *keep;      /* [N] ray mask of the callers' bounding-box pre-filter (nsr_aabb_keep / nsr_get_samples_window);
                                 NULL = every ray counts.  With skip_masked the kernels read it through the scalar cache one
                                 aligned 32-bit word at a time: the buffer must be readable up to the next 4-byte boundary
                                 behind its last byte (any allocator's padding; do not hand in the tail of a page-exact mapping) */
    double *loss;             /* device scalar, caller-zeroed */
    double *dl_depth;         /* [N]    out, optional */
    float *dl_rgb;            /* [N][3] out, optional */
    float w_color;            /* cfg mapping.w_color_loss */
    int32_t acts_masks_only;  /* with `acts`: 1 = the backward will want no parameter gradients (tracking), the forward only
                                 writes the relu masks (1 of the 13 KB per tile and decoder); 0 = everything.  ABI 8: bits 1..3 say the
                                 same for ONE decoder pass each (2: middle -- or the coarse stage's only pass --, 4: fine, 8: colour): a
                                 decoder nobody will step (src/Mapper.py:335-341 steps the colour decoder only) saves its masks only */
    /* --- saved decoder activations + workspace of the split backward (ABI 3/4; required for a call that will be differentiated since
     * ABI 6).  nsr_acts_floats(stage, n_rays, n_samples + n_surface) floats, device, uninitialised: nsr_render_fwd (given acts, zvals
     * and raw) runs as sample placement -> decoder passes -> compositor and writes, per decoder pass and 16-point tile of the sample list,
     * the five hidden states, the decoder's grid features and the relu masks (13 KB), the sample positions (fp64 and fp32) and, with the
     * fused loss, d raw; nsr_render_bwd, given the SAME pointer, runs dX / dW / finalize over them.  NULL: forward only (one launch,
     * nothing saved); nsr_render_bwd then fails. */
    float *acts;
    void *ev_pass_start;      /* ABI 6, optional hipEvent_t pair recorded on `stream` right before / after the decoder-pass kernel of */
    void *ev_pass_stop;       /* a differentiated forward (the forward's dominant kernel); NULL = no timing                           */
    int32_t skip_masked;      /* ABI 6, with `keep`: 1 = rays with keep[r] == 0 are REMOVED from the batch like the reference's
                                 compaction does (src/Mapper.py:471-481, src/Tracker.py:95-104: `batch_rays_d[inside_mask]`): no
                                 decoder evaluation, outputs depth = var = rgb = 0, no loss term, no gradient.  Their slots of
                                 raw / zvals / acts stay unwritten.  nsr_render_bwd must be given the same keep / skip_masked.
                                 Honoured by calls that will be differentiated (acts + zvals + raw given); a forward-only call
                                 renders every ray.  0 (default): keep only masks the fused loss; every ray is rendered. */
    int32_t pad2_;
    const uint8_t *grad_voxel_mask[4]; /* ABI 8, opt-in, read by nsr_render_bwd only: per grid slot a [Z][Y][X] byte mask (the layout
                                 nsr_frustum_mask writes and nsr_masked_adam takes) of the voxels whose gradient the caller will
                                 CONSUME.  With `frustum_feature_selection` the mapper's optimiser holds only `val[mask]`
                                 (src/Mapper.py:315-333,394-401): the reference's autograd scatters into the whole grid and
                                 `val.grad[~mask]` is never read.  Given a mask, the backward's scatter skips voxels whose byte is 0:
                                 dfeat of voxels inside the mask equals the dense run's (up to the order of the atomic adds),
                                 dfeat of voxels outside stays as the caller left it (zero).  Ray / pose and decoder gradients are
                                 unaffected (they never depended on dfeat).  NULL (default) = the reference's dense gradient. */
} nsr_render_args;

typedef struct nsr_bwd_args {
    const double *d_depth;    /* [N]    dL/d depth   */
    const double *d_var;      /* [N]    dL/d var, may be NULL (== 0)  */
    const float *d_rgb;       /* [N][3] dL/d rgb, may be NULL (== 0)  */
    const double *depth;      /* [N]    forward result (needed for the variance term) */
    float *d_rays_o;          /* [N][3] caller-zeroed, or NULL */
    float *d_rays_d;          /* [N][3] caller-zeroed, or NULL */
    float *workspace;         /* nsr_bwd_workspace_floats() floats, contents undefined */
    int64_t workspace_floats;
    int32_t max_blocks;       /* persistent-grid cap used to size the workspace (0 = library default) */
    int32_t overwrite_dparams; /* 0: every nsr_decoder.dparams is ACCUMULATED into (autograd semantics, caller-zeroed for a
                                 fresh gradient); 1: it is OVERWRITTEN (saves the caller the zero fill) */
    void *ev_start;           /* optional hipEvent_t pair recorded on `stream` right before the first / after the last */
    void *ev_stop;            /* kernel of the backward; NULL = no timing                                               */
    const double *grad_scale; /* optional DEVICE scalar every output gradient (d_depth, d_var, d_rgb) is multiplied by -- the
                                 incoming gradient of a loss node fused around the render (no host sync, no extra launch);
                                 NULL = 1 */
    int32_t loss_grads_from_forward; /* ABI 6, opt-in.  1: d_depth / d_rgb ARE nsr_render_args.dl_depth / dl_rgb exactly as the forward's
                                 loss epilogue wrote them (same pointers, contents unmodified, d_var NULL): the backward then uses
                                 the per-sample `d raw` that epilogue already left in `acts` and skips the compositor backward (one
                                 launch).  0 (default): d_depth / d_var / d_rgb are read as given -- a caller may mask or re-weight
                                 dl_* in place before the backward.  Only grad_scale is applied on top in either case. */
    int32_t pad_;
    void *ev_dx_done;         /* ABI 6, optional hipEvent_t recorded behind the dX kernel / behind the dW kernel: with ev_start / ev_stop */
    void *ev_dw_done;         /* they give the time of each kernel of the backward (dX | dW | finalize)                                 */
} nsr_bwd_args;

int nsr_version(void);
const char *nsr_last_error(void);

/* number of fp32 parameters of decoder slot `slot` (coarse 6 337 / middle 15 800 / fine 20 920 / color 15 899) */
int64_t nsr_param_count(int slot);
/* number of floats of the packed operand stream for decoder slot `slot` */
int64_t nsr_packed_count(int slot);
/* size of nsr_render_args.acts in floats (-1 on bad arguments) */
int64_t nsr_acts_floats(int stage, int64_t n_rays, int n_samples_total);

/* floats of scratch nsr_render_bwd needs for (stage, n_rays, max_blocks) */
int64_t nsr_bwd_workspace_floats(int stage, int64_t n_rays, int n_samples_total, int max_blocks);

int nsr_pack_params(int slot, const float *params, float *packed, void *stream);

int nsr_render_fwd(const nsr_render_args *a, void *stream);
int nsr_render_bwd(const nsr_render_args *a, const nsr_bwd_args *b, void *stream);

/* Renderer.eval_points forward: p [M][3] fp64 world points -> out [M][4] fp32 (rgb, occ) */
int nsr_eval_points_fwd(const nsr_render_args *a, const double *points, int64_t n_points, float *out, void *stream);

/* get_samples after the index draw: flat indices into the crop [H0,H1)x[W0,W1) (row-major) ->
 * rays + gathered depth/colour.  c2w is a row-major 3x4 (or 4x4) fp32 matrix with row stride c2w_stride. */
int nsr_get_samples(const int64_t *indices, int64_t n, int32_t H0, int32_t H1, int32_t W0, int32_t W1,
                    int32_t W_full, float fx, float fy, float cx, float cy,
                    const float *c2w, int32_t c2w_stride, const float *depth, const float *color,
                    float *rays_o, float *rays_d, float *out_depth, float *out_color, void *stream);

/* --- the mapper's sampling loop in one launch -----------------------------------------------------------------------------
 * Replaces, for the K frames of the mapping window, the K get_samples calls + torch.cat of src/Mapper.py:437-468 and the
 * bounding-box pre-filter of :471-481 (as a mask, like nsr_aabb_keep): indices [K][n] (row-major flat indices into the crop,
 * one draw per frame), frames[k] = that frame's depth [H][W], colour [H][W][3] and pose (row-major 3x4 or 4x4, row stride
 * c2w_stride floats), all device pointers; outputs are the concatenation over frames, [K*n] rays.  keep / kept_max optional
 * (kept_max caller-zeroed).  bound_lo / bound_hi: HOST arrays of 3 doubles.  K <= 32. */
typedef struct nsr_frame {
    const float *depth;
    const float *color;
    const float *c2w;
    int32_t c2w_stride;
    int32_t pad_;
} nsr_frame;
int nsr_get_samples_window(const int64_t *indices, int32_t K, int64_t n, int32_t H0, int32_t H1, int32_t W0, int32_t W1,
                           int32_t W_full, float fx, float fy, float cx, float cy, const nsr_frame *frames,
                           float *rays_o, float *rays_d, float *out_depth, float *out_color,
                           const double *bound_lo, const double *bound_hi, uint8_t *keep, float *kept_max, void *stream);
/* The same with the pixel draw of src/common.py:99 (`torch.randint(h * w, (n,))`, one call per keyframe there) inside the
 * kernel: ray t of the window takes one uniform index in [0, crop_h * crop_w) from philox4x32-10 (counter = (t, calls so far),
 * key = seed) and the drawn indices are written to indices_out [K * n] (for nsr_pose_grad and the caller).  rng_state: three
 * uint64 on the device -- {seed, calls so far, 0} -- owned by the caller and advanced by the kernel, so that a captured graph
 * draws afresh on every replay without the host.  Same distribution as the reference's draw, not the same stream. */
int nsr_get_samples_window_draw(int64_t *indices_out, uint64_t *rng_state, int32_t K, int64_t n, int32_t H0, int32_t H1, int32_t W0,
                                int32_t W1, int32_t W_full, float fx, float fy, float cx, float cy, const nsr_frame *frames,
                                float *rays_o, float *rays_d, float *out_depth, float *out_color,
                                const double *bound_lo, const double *bound_hi, uint8_t *keep, float *kept_max, void *stream);
/* The window kernel as the FIRST launch of a fused iteration (ABI 7): beside the sampling it zero-fills the iteration's gradient
 * buffer -- x-blocks of its own that store zeros over zero[0 .. zero_floats) (16-byte aligned) while the others place the rays:
 * the fill is bandwidth, the sampling a chain of latencies, side by side they cost the longer of the two instead of two launches
 * (src/Mapper.py:503: `loss.backward()` accumulates into zeroed .grad tensors; Mapper.py:437-481 for the sampling) -- and its last
 * block to finish writes the iteration's 16-byte header {loss accumulator (fp64) = 0, kept_max, 0.f}: nothing has to be zeroed
 * before the launch.  indices: [K * n] or NULL (then drawn as in nsr_get_samples_window_draw and written to indices_out).
 * state: four uint64 on the device -- {seed, calls so far, 0, 0} -- owned by the caller; words 2 and 3 are the launch's hand-off
 * (blocks done, bit pattern of the running maximum) and are zero again when it ends; two launches that share a state must not
 * overlap.  header must be 16-byte aligned (one 16-byte store), state 8-byte aligned.  K, n >= 1. */
int nsr_get_samples_window_fused(const int64_t *indices, int64_t *indices_out, uint64_t *state, int32_t K, int64_t n, int32_t H0, int32_t H1,
                                 int32_t W0, int32_t W1, int32_t W_full, float fx, float fy, float cx, float cy, const nsr_frame *frames,
                                 float *rays_o, float *rays_d, float *out_depth, float *out_color,
                                 const double *bound_lo, const double *bound_hi, uint8_t *keep, float *header,
                                 float *zero, int64_t zero_floats, void *stream);
/* The fused window launch of ONE RANK of a ray-sharded mapping iteration (ABI 8; one process per GPU): the in-kernel
 * pixel draw of nsr_get_samples_window_fused for this rank's state, and -- instead of an all-reduce of the kept rays' maximum depth,
 * the one scalar of the WHOLE batch the render needs (src/utils/Renderer.py:109,144) -- the same draw REPEATED for the other ranks:
 * peer_seeds (HOST array, n_peers <= 15 entries) are the seeds of their states, every rank has made the same number of calls (the
 * call counter of `state` is used for all of them), keyframes and poses are replicated: extra blocks re-draw peer p's pixels, gather
 * their depths and apply the bounding-box pre-filter (src/Mapper.py:471-481) without writing any ray, and the header's kept maximum
 * becomes the maximum over the union batch -- bit-identical on every rank, no collective between the sampling and the render.
 * header: 16-byte aligned; state: 8-byte aligned (also required by nsr_get_samples_window_fused). */
int nsr_get_samples_window_sharded(int64_t *indices_out, uint64_t *state, const uint64_t *peer_seeds, int32_t n_peers, int32_t K, int64_t n,
                                   int32_t H0, int32_t H1, int32_t W0, int32_t W1, int32_t W_full, float fx, float fy, float cx, float cy,
                                   const nsr_frame *frames, float *rays_o, float *rays_d, float *out_depth, float *out_color,
                                   const double *bound_lo, const double *bound_hi, uint8_t *keep, float *header,
                                   float *zero, int64_t zero_floats, void *stream);
/* gradient of the K poses from the ray gradients of such a window (autograd of src/common.py:74-88; local BA,
 * src/Mapper.py:417-419,441-453): d_c2w + k * out_stride holds rows 0..2 of pose k's gradient, row-major (12 floats;
 * out_stride = 12 for 3x4 poses, 16 for 4x4 ones whose last row the caller zero-fills). */
int nsr_pose_grad(const int64_t *indices, int32_t K, int64_t n, int32_t H0, int32_t H1, int32_t W0, int32_t W1,
                  float fx, float fy, float cx, float cy, const float *d_rays_o, const float *d_rays_d,
                  float *d_c2w, int32_t out_stride, void *stream);

/* --- the per-iteration grid update of Mapper.optimize_map, fused ---------------------------------
 * Replaces, for one feature grid, `val[mask] = val_grad` (src/Mapper.py:394-401), the Adam step on the masked leaf
 * (:368-379,504, torch.optim.Adam defaults) and the write-back `val[mask] = val_grad.detach()` (:511-519) by one in-place
 * pass over the grid: for every voxel whose mask byte is non-zero (mask == NULL: every voxel) and each of its 32 channels
 *     m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= step_size * m / (sqrt(v)/bias2_sqrt + eps)
 * with step_size = lr/(1-b1^t) and bias2_sqrt = sqrt(1-b2^t) evaluated by the caller in double precision, exactly the two
 * host scalars torch.optim.Adam forms (t = number of steps in which this grid received a gradient).
 * p, g, m, v: [Z][Y][X][32] fp32 (channels-last, like every grid of this ABI); voxel_mask: [Z][Y][X] uint8.
 * lr == 0 still updates the moments, exactly like torch.optim.Adam. */
int nsr_masked_adam(float *p, const float *g, float *m, float *v, const uint8_t *voxel_mask, int64_t n_voxels,
                    float step_size, float beta1, float beta2, float eps, float bias2_sqrt, void *stream);

/* The same for up to 4 grids in ONE launch pair with the step counts kept on the device, so that a whole mapping iteration
 * can be captured in a hipGraph (no host scalar changes between replays): steps[i] (device int32, one per grid) is
 * incremented, step_size = lr / (1 - b1^t) and bias2_sqrt = sqrt(1 - b2^t) are formed on the device in fp64, then every
 * grid is updated like nsr_masked_adam does.  zero_grad != 0 also clears the gradient of the voxels it read (so that a
 * persistent, accumulated-into gradient buffer needs no separate fill).  scratch: 8 device floats. */
typedef struct nsr_adam_grid {
    float *p;
    float *g;
    float *m;
    float *v;
    const uint8_t *voxel_mask;
    int64_t n_voxels;
    int32_t *step;            /* device counter of this grid */
    float lr;
    int32_t pad_;
} nsr_adam_grid;
int nsr_masked_adam_multi(const nsr_adam_grid *grids, int32_t n_grids, double beta1, double beta2, double eps,
                          int32_t zero_grad, float *scratch, void *stream);   /* (betas as doubles since ABI 7: the bias corrections
                          1 - beta^t are formed from them in fp64, like torch's python scalars) */

/* The dense rest of the callers' optimiser -- torch.optim.Adam over the decoder parameters and the camera tensors
 * (src/Mapper.py:368-387,504; src/Tracker.py:214-222,127) -- for up to 4 flat fp32 spans (a decoder's parameter blob, an [n,7] pose
 * tensor) in ONE launch pair with the step counts on the device: the same single-tensor Adam arithmetic as nsr_masked_adam_multi
 * (lerp / addcmul / addcdiv order, bias corrections formed on the device in fp64; lr == 0 still updates the moments), every element of
 * every span.  torch's capturable Adam takes ~10 launches per step for the same; together with nsr_masked_adam_multi a whole optimiser
 * step of a mapping iteration is four launches.  zero_grad != 0 clears the gradient it read.  scratch: 8 device floats. */
typedef struct nsr_adam_span {
    float *p;
    float *g;
    float *m;
    float *v;
    int64_t n;
    int32_t *step;            /* device counter of this span */
    float lr;
    int32_t pad_;
} nsr_adam_span;
int nsr_flat_adam(const nsr_adam_span *spans, int32_t n_spans, double beta1, double beta2, double eps, int32_t zero_grad, float *scratch,
                  void *stream);     /* betas as doubles: 1 - beta is formed in fp64 and then rounded, like torch's `value=1 - beta2` */

/* --- packing for the one-collective gradient exchange -------------------------------------------------------
 * Gathers (unpack = 0) the listed voxel rows (32 floats each, rows[] = voxel indices in [Z][Y][X] raster order) of up to 4
 * channels-last grid-gradient tensors and up to 4 flat spans into `packed` (rows of grid 0, grid 1, ..., then the spans), or
 * scatters `packed` back (unpack = 1).  The caller all-reduces `packed` in between (torch.distributed / RCCL): the library
 * itself stays free of communication.  Replaces the nsr_comm_* / nsr_reduce_grid_grads entry points sketched in SURVEY §8(b). */
typedef struct nsr_rows {
    float *grid;
    const int64_t *rows;
    int64_t n_rows;
} nsr_rows;
typedef struct nsr_span {
    float *ptr;
    int64_t n;
} nsr_span;
int nsr_pack_rows(const nsr_rows *grids, int32_t n_grids, const nsr_span *spans, int32_t n_spans, float *packed,
                  int32_t unpack, void *stream);

/* The tracker's loss on rendered outputs -- src/Tracker.py:108-124 (Tracker.optimize_cam_in_batch): with
 * tmp = |gt_depth - depth| / sqrt(var + 1e-10) (fp64, `var` detached), mask = keep & (gt_depth > 0) and, when
 * handle_dynamic, & (tmp < 10 * median(tmp over the kept rays)) (torch.median: the lower middle element; NaN if any
 * element is NaN),
 *     *loss += sum_mask tmp  [+ w_color * sum_mask |gt_color - rgb|   when use_color]
 * and dl_depth [n] / dl_rgb [n][3] receive d loss / d depth and d loss / d rgb -- what autograd would hand to the backward
 * of render_batch_ray (feed them to nsr_render_bwd as d_depth / d_rgb).  `keep` (or NULL) is the bounding-box pre-filter
 * of :92-104 as a byte mask (nsr_aabb_keep / nsr_get_samples_window): the reference compacts the batch instead, so its
 * median runs over exactly the kept rays.  One launch, no host synchronisation, any n. */
int nsr_tracking_loss(int64_t n_rays, const float *gt_depth, const float *gt_color, const uint8_t *keep,
                      const double *depth, const double *var, const float *rgb,
                      int32_t handle_dynamic, int32_t use_color, float w_color,
                      double *loss, double *dl_depth, float *dl_rgb, void *stream);

/* get_camera_from_tensor / quad2rotation -- src/common.py:137-176: n camera tensors [quaternion (w,x,y,z) | translation]
 * (7 floats each) -> n 3x4 matrices [R | T] in `rt` (d_rt == NULL), or -- d_rt != NULL -- the backward: d_cam [n][7] from
 * d_rt [n][3][4].  The pose parametrisation of the tracker and of local BA (Tracker.py:87, Mapper.py:447-451). */
int nsr_camera_from_tensor(const float *cam, int64_t n, float *rt, const float *d_rt, float *d_cam, void *stream);

/* --- frustum feature selection ---------------------------------------------------------------------
 * Replaces Mapper.get_mask_from_c2w (src/Mapper.py:93-164) for one non-coarse feature grid: every voxel centre (xs[ix],
 * ys[iy], zs[iz]: the per-axis torch.linspace over the scene bound, :111-113, device arrays) is projected with
 * w2c = inv(c2w) (HOST array, 12 floats = rows 0..2; :120-131), its depth is looked up in the current depth image with
 * cv2.remap INTER_LINEAR semantics (:133-140), zero depths are replaced by the maximum remapped depth (:147-148), and the
 * voxel is selected when it projects inside the image with 0 <= -z <= depth + 0.5 (:142-151) or lies within 0.5 m of the
 * camera centre cam_center (HOST array, 3 floats; :155-161).
 * depth: [H][W] fp32 device; voxel_mask: [Z][Y][X] uint8 device (the layout nsr_masked_adam takes; the reference's
 * (X,Y,Z) return value is its transpose, which Mapper.py:318 undoes); workspace: nsr_frustum_workspace_floats(nx*ny*nz)
 * floats of device scratch. */
int64_t nsr_frustum_workspace_floats(int64_t n_voxels);
int nsr_frustum_mask(const float *w2c, const float *cam_center, double fx, double fy, double cx, double cy,
                     int32_t H, int32_t W, const float *depth, const float *xs, const float *ys, const float *zs,
                     int32_t nx, int32_t ny, int32_t nz, float *workspace, uint8_t *voxel_mask, void *stream);

/* --- SURVEY §8(f) rank 1 (third item): the callers' bounding-box pre-filter without compaction --------------------------
 * Mapper.optimize_map (src/Mapper.py:471-481) and Tracker.optimize_cam_in_batch (src/Tracker.py:95-104) drop, before
 * rendering, every ray whose depth lies beyond the scene bound:  t = min_axis max((lo-o)/d, (hi-o)/d)  (fp64),
 * keep = t >= gt_depth, then index all four ray tensors with the boolean mask (a `nonzero` = a host sync per iteration).
 * nsr_aabb_keep writes the same mask as bytes and, if kept_max != NULL (device float, caller-set to 0), the maximum
 * gt_depth over the KEPT rays -- the batch-global scalar render_batch_ray needs -- so a caller can render the whole batch
 * and weight its loss by the mask instead of compacting.  bound_lo / bound_hi: HOST arrays of 3 doubles. */
int nsr_aabb_keep(const float *rays_o, const float *rays_d, const float *gt_depth, int64_t n,
                  const double *bound_lo, const double *bound_hi, uint8_t *keep, float *kept_max, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NSR_H_ */
