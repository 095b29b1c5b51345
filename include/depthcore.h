/*
 * depthcore.h -- C ABI of libdepthcore.so: the MI355X (gfx950) native hot path of
 * self-supervised monocular depth training (Monodepth2-family photometric step).
 *
 * Drop-in boundary: the reference is pure Python on PyTorch and
 * every op below is, in the reference, a chain of stock ATen calls made from
 * `layers.py` / `trainer.py`.  Each entry point cites the reference code it replaces.
 * The Python facade (`self-supervised-depth-estimation_amd/layers.py`, `networks/`,
 * `trainer.py`) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - all tensors fp32, NCHW, contiguous, device memory owned by the caller;
 *     the library never allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it;
 *   - return value: 0 = ok, negative = DC_E* (no exceptions cross the boundary);
 *   - "frame 0/1" means source frame_id -1 / +1 (`trainer.py:482`);
 *   - deterministic: no float atomics; partial sums are reduced in fixed order.
 */
#ifndef DEPTHCORE_H
#define DEPTHCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DC_OK 0
#define DC_EINVAL (-1)      /* bad shape / null pointer / unsupported option */
#define DC_ELAUNCH (-2)     /* hipLaunch error */
#define DC_EWORKSPACE (-3)  /* workspace too small */

#define DC_MAX_SCALES 4

/* option flags of the fused photometric loss (`trainer.py:531-622` ablation branches) */
#define DC_OPT_NO_AUTOMASK 1u   /* opt.disable_automasking */
#define DC_OPT_AVG_REPROJ 2u    /* opt.avg_reprojection */
#define DC_OPT_NO_SSIM 4u       /* opt.no_ssim */
#define DC_OPT_ALIGN_CORNERS 8u /* grid_sample(align_corners=True); default False = installed-torch default */
#define DC_OPT_PRED_MASK 32u    /* opt.predictive_mask (trainer.py:571-584): reprojection losses are multiplied by `pred_mask[s]`; needs
                                   DC_OPT_NO_AUTOMASK (trainer.py:116-117).  The BCE weighting term of the masks is the caller's */
#define DC_OPT_PHOTO_SPLIT 64u  /* pin the training forward's mode for THIS desc, whatever dc_set_photo_full says when its backward runs: */
#define DC_OPT_PHOTO_FULL 128u  /* SPLIT = the round-4 split, FULL = all the way (see dc_set_photo_full).  dc_photo_bwd must get the same bit
                                   as its dc_photo_fwd (it re-derives the workspace layout from it); with neither bit the process-wide
                                   setting decides at each call */
#define DC_OPT_NO_GRAD 16u      /* dc_photo_fwd only: evaluation -- the forward does not emit d(loss)/d(warped), the workspace
                                   is smaller and dc_photo_bwd on this descriptor returns DC_EINVAL */

const char* dc_version(void);
/* Compiled-for architecture string, e.g. "gfx950". */
const char* dc_arch(void);
/* Reads and clears HIP's per-thread last-error code (returned as an int, 0 = none).  The dc_* entry points detect a failed
 * launch through that code; call this after an error that was NOT theirs (e.g. a hipGraph capture that was invalidated) so
 * that the next launch does not report it as DC_ELAUNCH. */
int dc_clear_error(void);
/* Ends whatever hipGraph capture `stream` is in (a capture that was invalidated leaves its origin stream in capture mode, and
 * every later launch on it fails), discards the graph and clears the error state.  Returns the capture status it found
 * (0 none, 1 active, 2 invalidated). */
int dc_abort_capture(void* stream);

/* ------------------------------------------------------------------ a5 */
/* layers.py:28-103 transformation_from_parameters (+rot_from_axisangle, get_translation_matrix).
 * axisangle, translation: (B,3); invert: 0/1; out M: (B,4,4). */
int dc_pose_matrix_fwd(const float* axisangle, const float* translation, int invert, float* M, int B,
                       void* stream);
/* dM (B,4,4) -> d_axisangle (B,3), d_translation (B,3). */
int dc_pose_matrix_bwd(const float* axisangle, const float* translation, int invert, const float* dM,
                       float* d_axisangle, float* d_translation, int B, void* stream);

/* The pose networks' tail and the cam_T_cam of their callers as ONE launch each way (networks/pose_decoder.py:50-54,
 * networks/pose_cnn.py:48-52: out.mean(3).mean(2), 0.01 * out.view(-1, nf, 1, 6), [..., :3] / [..., 3:]; trainer.py:416-419,436-440:
 * transformation_from_parameters(axisangle[:, i], translation[:, i], invert)).  y: the last convolution's output (N, 6 nf, P pixels);
 * vec (N, 6 nf) = scale * mean over the pixels, read as (N, nf, [axisangle | translation]); group g: rows [row0, row0 + rows) of
 * frame `slot` -> M[g] (rows,4,4) (`groups`, `M`, `dM` are HOST arrays of ngroups <= 8 entries; nf <= 8).  Backward: dM[g]
 * (rows,4,4) or NULL -> d_y (N, 6 nf, P), zeros in channels no group reads. */
typedef struct dc_pose_group { int row0, rows, slot, invert; } dc_pose_group;
int dc_pose_head_fwd(const float* y, int N, int nf, int P, float scale, const dc_pose_group* groups, int ngroups, float* vec,
                     float* const* M, void* stream);
int dc_pose_head_bwd(const float* vec, int N, int nf, int P, float scale, const dc_pose_group* groups, int ngroups,
                     const float* const* dM, float* d_y, void* stream);

/* ------------------------------------------------------------------ a6 */
/* layers.py:16-25 disp_to_depth: scaled = 1/max + (1/min-1/max)*disp; depth = 1/scaled. n elements. */
int dc_disp_to_depth_fwd(const float* disp, float* scaled, float* depth, size_t n, float min_depth,
                         float max_depth, void* stream);
/* d_disp = g_scaled*(1/min-1/max) - g_depth*depth^2*(1/min-1/max); g_* nullable. */
int dc_disp_to_depth_bwd(const float* disp, const float* g_scaled, const float* g_depth, float* d_disp,
                         size_t n, float min_depth, float max_depth, void* stream);

/* ------------------------------------------------------------------ a7 */
/* layers.py:139-161: fills pix_coords (B,3,H*W) = [x; y; 1] with x = i mod W, y = i div W (exact). */
int dc_pix_coords(float* pix_coords, int B, int H, int W, void* stream);
/* layers.py:163-168 BackprojectDepth.forward: depth (B,1,H,W), inv_K (B,4,4) -> cam (B,4,H*W). */
int dc_backproject_fwd(const float* depth, const float* inv_K, float* cam, int B, int H, int W, void* stream);
/* g_cam (B,4,HW) -> d_depth (B,1,H,W). */
int dc_backproject_bwd(const float* g_cam, const float* inv_K, float* d_depth, int B, int H, int W,
                       void* stream);

/* ------------------------------------------------------------------ a8 */
/* layers.py:171-193 Project3D.forward: points (B,4,HW), K,T (B,4,4) -> grid (B,H,W,2). */
int dc_project3d_fwd(const float* points, const float* K, const float* T, float* grid, int B, int H, int W,
                     float eps, void* stream);
/* g_grid (B,H,W,2) -> d_points (B,4,HW) and d_T (B,4,4) (d_T nullable).  `ws` : workspace of
 * dc_project3d_bwd_workspace(B,H,W) bytes. */
size_t dc_project3d_bwd_workspace(int B, int H, int W);
int dc_project3d_bwd(const float* points, const float* K, const float* T, const float* g_grid,
                     float* d_points, float* d_T, void* ws, int B, int H, int W, float eps, void* stream);

/* ------------------------------------------------------------------ a9 */
/* F.grid_sample(img, grid, mode=bilinear, padding_mode="border") as called at trainer.py:508-511.
 * img (B,C,H,W), grid (B,Ho,Wo,2) -> out (B,C,Ho,Wo). */
int dc_grid_sample_fwd(const float* img, const float* grid, float* out, int B, int C, int H, int W, int Ho,
                       int Wo, int align_corners, void* stream);
/* g_out -> d_grid (B,Ho,Wo,2) (images are leaves without grad in the reference). */
int dc_grid_sample_bwd(const float* img, const float* grid, const float* g_out, float* d_grid, int B, int C,
                       int H, int W, int Ho, int Wo, int align_corners, void* stream);

/* ------------------------------------------------------------------ a10 */
/* F.interpolate(x, [Ho,Wo], mode="bilinear", align_corners=False) (trainer.py:474-475), x (B,C,h,w). */
/* upsample(x) = F.interpolate(x, scale_factor=2, mode="nearest") (layers.py:196-199): (BC,h,w) -> (BC,2h,2w); backward = 2x2 block sums.
 * (The depth decoder does not call it: its nearest-x2 is folded into dc_conv3x3_fwd's loader.) */
int dc_upsample_nearest2x_fwd(const float* x, float* out, int BC, int h, int w, void* stream);
int dc_upsample_nearest2x_bwd(const float* g_out, float* d_x, int BC, int h, int w, void* stream);
int dc_upsample_bilinear_fwd(const float* x, float* out, int BC, int h, int w, int Ho, int Wo, void* stream);
int dc_upsample_bilinear_bwd(const float* g_out, float* d_x, int BC, int h, int w, int Ho, int Wo,
                             void* stream);

/* ------------------------------------------------------------------ a11 */
/* layers.py:218-248 SSIM.forward: x,y (B,C,H,W) -> clamp((1-n/d)/2,0,1) (B,C,H,W). */
int dc_ssim_fwd(const float* x, const float* y, float* out, int BC, int H, int W, void* stream);
/* g_out -> d_x and d_y (each nullable). */
int dc_ssim_bwd(const float* x, const float* y, const float* g_out, float* d_x, float* d_y, int BC, int H,
                int W, void* stream);

/* ------------------------------------------------------------------ a13 */
/* layers.py:202-215 get_smooth_loss: disp (B,1,h,w), img (B,C,h,w) -> scalar out[0].
 * ws: dc_smooth_workspace(B,h,w) bytes. */
size_t dc_smooth_workspace(int B, int h, int w);
int dc_smooth_fwd(const float* disp, const float* img, float* out, void* ws, int B, int C, int h, int w,
                  void* stream);
/* d_disp = g[0] * dL/ddisp. */
int dc_smooth_bwd(const float* disp, const float* img, const float* g, float* d_disp, int B, int C, int h,
                  int w, void* stream);

/* ------------------------------------------------------------------ a12/a14/a15 fused */
/* The fused photometric step: trainer.py:465-515 (generate_images_pred) + 517-529
 * (compute_reprojection_loss) + 531-622 (compute_losses) in three launches forward and three backward.
 *
 * All sizes in one struct so that the ctypes binding stays readable. */
typedef struct dc_photo_desc {
    int32_t B, H, W;              /* per-rank batch, full resolution (opt.height, opt.width) */
    int32_t num_scales;           /* len(opt.scales) <= 4; scale s has size (H>>s, W>>s) */
    uint32_t flags;               /* DC_OPT_* */
    float min_depth, max_depth;   /* options.py:117-124 */
    float smoothness;             /* opt.disparity_smoothness */
    /* inputs */
    const float* target;          /* inputs[("color",0,0)]   (B,3,H,W) */
    const float* source[2];       /* inputs[("color",-1,0)], inputs[("color",+1,0)] */
    const float* color_s[DC_MAX_SCALES]; /* inputs[("color",0,s)]  (B,3,H>>s,W>>s) */
    const float* packed[3];       /* optional: pixel-interleaved RGBx copies (B,H,W,4) of target / source[0] / source[1]
                                     (dc_data_to_rgbx / dc_pack_rgbx, 16-byte aligned), all three or none.  The kernels gather
                                     from this layout; when absent dc_photo_fwd builds it in the workspace at every call */
    const float* K;               /* inputs[("K",0)]      (B,4,4) */
    const float* inv_K;           /* inputs[("inv_K",0)]  (B,4,4) */
    const float* T[2];            /* outputs[("cam_T_cam",0,-1/+1)]  (B,4,4) */
    const float* disp[DC_MAX_SCALES];    /* outputs[("disp",s)]  (B,1,H>>s,W>>s) */
    const float* noise[DC_MAX_SCALES];   /* tie-break randn (B,2,H,W) per scale ((B,1,H,W) with
                                            AVG_REPROJ), trainer.py:594-595; NULL = on-device RNG */
    uint64_t rng_seed;            /* used when noise[s] == NULL: counter-based on-device noise keyed non-linearly by
                                     (all 64 seed bits, scale, frame) and indexed by pixel.  Distribution: Irwin-Hall(4)
                                     of the hash bytes, zero mean / unit variance, 1021 values within +-3.45 sigma -- a
                                     tie-breaker, not torch.randn; pass noise[] for bit-parity with a randn stream */
    const uint64_t* rng_seed_dev; /* optional: when not NULL the seed is READ FROM DEVICE MEMORY when the kernel runs
                                     (rng_seed is ignored) -- a step captured in a hipGraph advances the value with a
                                     device-side add and every replay draws a new noise field */
    /* outputs of forward */
    float* losses;                /* (num_scales+1): loss/0.., loss */
    uint8_t* argmin[DC_MAX_SCALES];      /* (B,H,W) winning channel of torch.min(combined,1) */
    /* optional materialised log tensors (trainer.py:480-515); any may be NULL */
    float* depth[DC_MAX_SCALES];         /* ("depth",0,s)   (B,1,H,W) */
    float* sample[DC_MAX_SCALES][2];     /* ("sample",f,s)  (B,H,W,2) */
    float* color[DC_MAX_SCALES][2];      /* ("color",f,s)   (B,3,H,W) */
    float* identity_selection[DC_MAX_SCALES]; /* "identity_selection/s" (B,H,W) float 0/1 */
    /* backward */
    const float* g_losses;        /* (num_scales+1) upstream gradient of `losses` */
    float* d_disp[DC_MAX_SCALES]; /* (B,1,H>>s,W>>s) */
    float* d_T[2];                /* (B,4,4) */
    /* DC_OPT_PRED_MASK only */
    const float* pred_mask[DC_MAX_SCALES]; /* outputs["predictive_mask"][("disp",s)] at FULL resolution (B,2,H,W): the caller
                                              upsamples (trainer.py:574-577, dc_upsample_bilinear_fwd); channel f = frame -1 / +1 */
    float* d_pred_mask[DC_MAX_SCALES];     /* backward: (B,2,H,W) */
    /* scratch */
    void* workspace;              /* dc_photo_workspace(desc) bytes, same buffer for fwd and bwd */
    size_t workspace_bytes;
    /* optional per-scale poses: `--pose_model_type posecnn` rebuilds T at every scale from the translation scaled by that
     * scale's mean inverse depth (trainer.py:490-499).  T_scale[s][f] != NULL replaces T[f] at scale s (all 2*num_scales or
     * none); the backward then writes d_T_scale[s][f] (B,4,4) per scale and leaves d_T[] untouched (it may be NULL). */
    const float* T_scale[DC_MAX_SCALES][2];
    float* d_T_scale[DC_MAX_SCALES][2];
} dc_photo_desc;

size_t dc_photo_workspace(const dc_photo_desc* d);
int dc_photo_fwd(const dc_photo_desc* d, void* stream);
int dc_photo_bwd(const dc_photo_desc* d, void* stream);
/* Per-launch algorithmic bytes of the fused forward / backward kernel, for bench.py. */
double dc_photo_algorithmic_bytes(const dc_photo_desc* d, int backward);
/* 1 (default; env DC_PHOTO_FULL): the training forward of the default loss configuration contracts all the way to
 * d(loss)/d(upsampled disp) and the pose sums (one float per pixel and scale leaves it; dc_photo_bwd is the transposed upsample
 * alone).  0: the round-4 split -- the forward emits d(loss)/d(source coordinates), a pointwise backward chains them (A/Bs).
 * Returns the previous setting.  Must not change between a dc_photo_fwd and its dc_photo_bwd. */
int dc_set_photo_full(int mode);
/* The current process-wide setting.  Callers that may see the setter between a forward and its backward read it once, at the forward,
 * and pin it in the desc with DC_OPT_PHOTO_FULL / DC_OPT_PHOTO_SPLIT (depthcore/ops.py does). */
int dc_get_photo_full(void);

/* Measurement hook (bench.py `roofline`): when enabled, dc_photo_fwd / dc_photo_bwd bracket their
 * dominant kernel (photo_fwd_kernel / photo_bwd_kernel) AND their whole launch chain (forward: identity + smoothness +
 * photo_fwd + finalize; backward: photo_bwd + disp_grad + pose_grad) with hipEvents on the launch stream.
 * dc_profile_enable(n) allocates n event pairs per bracket (0 disables and frees);
 * dc_profile_collect synchronises on the recorded events and returns summed milliseconds and launch counts since the
 * last enable/collect (any output pointer may be NULL).
 * The compute entry points of this library are stateless and re-entrant; these hooks and dc_conv_profile_* are the
 * only process-global mutable state (event pools guarded by a mutex) and are meant for one measuring thread. */
int dc_profile_enable(int max_launches);
int dc_profile_collect(double* fwd_ms, int* fwd_launches, double* bwd_ms, int* bwd_launches,
                       double* fwd_chain_ms, double* bwd_chain_ms);

/* ------------------------------------------------------------------ a2/a3 decoder blocks */
/* layers.py:106-136 + 196-199 and networks/depth_decoder.py:50-66 (and the 3x3 conv + ReLU pairs of
 * networks/pose_decoder.py:27-29,45-48) as ONE fused convolution (Winograd F(2x2,3x3) on even widths, else direct):
 *   y = act( conv3x3( pad1( cat( up2?(x0), x1 ) ) ) + bias )
 * x0 (B,C0,H>>up0,W>>up0) nearest-upsampled x2 on the fly when up0=1, x1 (B,C1,H,W) nullable skip,
 * weight (Co,C0+C1,3,3), bias nullable; act: 0 none, 1 ELU, 2 sigmoid, 3 ReLU, 4 tanh; pad_mode: 0 ReflectionPad2d(1),
 * 1 ZeroPad2d(1).  Output (B,Co,H,W).  ws: dc_conv3x3_fwd_workspace bytes. */
size_t dc_conv3x3_fwd_workspace(int C0, int C1, int B, int Co, int H, int W);
int dc_conv3x3_fwd(const float* x0, int C0, int up0, const float* x1, int C1, const float* weight,
                   const float* bias, float* y, void* ws, int B, int Co, int H, int W, int act, int pad_mode,
                   void* stream);
/* Backward: gy (B,Co,H,W) is the gradient wrt the *activated* output y (ELU / sigmoid derivatives are
 * taken from y).  Produces dx0 (pre-upsample shape, 2x2-summed when up0), dx1, dweight, dbias (each
 * nullable).  ws: dc_conv3x3_bwd_workspace bytes.  Deterministic (no atomics). */
size_t dc_conv3x3_bwd_workspace(int C0, int C1, int B, int Co, int H, int W);
int dc_conv3x3_bwd(const float* x0, int C0, int up0, const float* x1, int C1, const float* weight,
                   const float* y, const float* gy, float* dx0, float* dx1, float* dweight, float* dbias,
                   void* ws, int B, int Co, int H, int W, int act, int pad_mode, void* stream);
/* dx0 += addend0, dx1 += addend1 (nullable, shapes of dx0 / dx1) inside the pass that writes them: the gradient of another
 * consumer of the same input (the decoder's x feeds dispconv AND the next upconv, networks/depth_decoder.py:55-66). */
int dc_conv3x3_bwd_add(const float* x0, int C0, int up0, const float* x1, int C1, const float* weight,
                       const float* y, const float* gy, float* dx0, float* dx1, const float* addend0, const float* addend1,
                       float* dweight, float* dbias, void* ws, int B, int Co, int H, int W, int act, int pad_mode, void* stream);

/* ------------------------------------------------------------------ a1 BatchNorm + residual + ReLU */
/* Training-mode nn.BatchNorm2d fused with the residual add and ReLU of torchvision's BasicBlock / Bottleneck
 * (reached from networks/resnet_encoder.py:87-98): y = relu?( bn(x) [+ res] ).  x, res, y: (N,C,H,W) with
 * HW = H*W; gamma, beta, save_mean, save_invstd, running_*: (C).  running_* nullable (then not updated);
 * they are updated like torch (momentum, unbiased variance).  `groups` > 1 normalises `groups` equal
 * sub-batches independently (save_mean / save_invstd then hold groups*C values) -- bit-for-bit the statistics
 * and running-stat updates of calling the module once per sub-batch in order; it lets the two pose pairs of
 * trainer.py:404-405 share one encoder pass.  ws: dc_bn_workspace(N,C,HW) bytes. */
size_t dc_bn_workspace(int N, int C, int HW);
/* relu_mask (nullable, dc_bn_mask_bytes(N,C,HW) bytes; 0 = not available for this shape): when relu=1 the forward also
 * stores [y > 0] as one bit per element (wave ballots), which the backward reads instead of y -- 1/32 of the bytes, in
 * each of its two passes. */
size_t dc_bn_mask_bytes(int N, int C, int HW);
int dc_bn_relu_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                   float* save_mean, float* save_invstd, float* running_mean, float* running_var, void* ws,
                   void* relu_mask, int N, int C, int HW, float eps, float momentum, int relu, int groups, void* stream);
/* gy is the gradient wrt y; the ReLU mask is taken from relu_mask (as written by the forward) or, when that is NULL,
 * from y (one of the two is required when relu=1).  dres (nullable) receives the gradient of the residual input;
 * dgamma, dbeta nullable. */
int dc_bn_relu_bwd(const float* x, const float* y, const float* gy, const float* gamma, const float* save_mean,
                   const float* save_invstd, float* dx, float* dres, float* dgamma, float* dbeta, void* ws,
                   const void* relu_mask, int N, int C, int HW, int relu, int groups, void* stream);

/* ------------------------------------------------------------------ a1 BatchNorm folded into the neighbouring convolutions
 * The same training-mode BatchNorm2d (+ReLU, +residual) of torchvision's BasicBlock / Bottleneck (networks/resnet_encoder.py:87-98),
 * without its stand-alone passes over HBM wherever a neighbouring convolution already holds the tensor:
 *   forward   conv A (statistics epilogue: per-channel partial {sum, sum of squares} of its raw output x)
 *             -> dc_bn_finalize (partials -> mean, invstd, running statistics, scale = gamma*invstd, shift = beta - mean*scale)
 *             -> conv B reads relu(scale*x + shift) in its loader            (a BatchNorm + ReLU with ONE consumer), or
 *                dc_bn_apply writes y = relu?(scale*x + shift [+ res])       (block outputs: several consumers);
 *   backward  conv B's data gradient masks its result with the ReLU decision in the store epilogue (g') and emits partial
 *             {sum g', sum g'*(x - mean)} -> dc_bn_bwd_finalize (coefficients, dgamma, dbeta)
 *             -> dc_bn_bwd_apply: dx = a*g' + b*(x - mean) + c0;   conv B's weight gradient re-forms relu(scale*x + shift)
 *             in its loader.
 * Partials: float2 part[channel][p], p < nparts; partial p covers pixels of one BatchNorm group, group(p) = min(p / ppg,
 * groups - 1) (the *_parts queries return nparts and ppg, 0 = this shape has no epilogue: use dc_bn_stats / dc_bn_relu_bwd).
 * Everything is summed in a fixed order: deterministic.  Arithmetic = dc_bn_relu_fwd / _bwd up to the order of the sums. */
typedef struct dc_bn_fold {
    int groups;                 /* BatchNorm groups of the batch (dc_bn_relu_fwd); image b belongs to group b / (B / groups) */
    /* input side (forward, weight gradient): the convolution's input is relu(in_scale[g,c]*x + in_shift[g,c]); NULL = plain x.
     * In the data gradient the same pair re-derives the ReLU decision when bn_mask is NULL. */
    const float* in_scale;      /* (groups, Ci) */
    const float* in_shift;      /* (groups, Ci) */
    /* output side, forward: statistics epilogue.  (Co, nparts, 2) floats, nparts = dc_*_stat_parts(); NULL = none */
    float* stat_part;
    /* input side, data gradient: BatchNorm-backward epilogue.  bn_x = the raw input of the BatchNorm whose ReLU-ed output this
     * convolution read (same shape as dx), bn_mean (groups, Ci), bn_mask = that BatchNorm's ReLU bit mask (dc_bn_apply; for
     * decisions that involved a residual) or NULL, bwd_part (Ci, nparts, 2) floats, nparts = dc_*_bwd_parts(); NULL = none */
    const float* bn_x;
    const float* bn_mean;
    const void* bn_mask;
    float* bwd_part;
} dc_bn_fold;
/* stand-alone statistics pass in the partial layout, for producers without the epilogue: x (N,C,HW), HW % 4 == 0 */
int dc_bn_stat_parts(int N, int C, int HW, int groups, int* ppg);
int dc_bn_stats(const float* x, float* part, int N, int C, int HW, int groups, void* stream);
/* count = elements per (group, channel).  mean, invstd, scale, shift: (groups, C); running_* nullable, updated group by group */
int dc_bn_finalize(const float* part, int nparts, int ppg, double count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float* mean, float* invstd, float* scale, float* shift,
                   int C, int groups, float eps, float momentum, void* stream);
/* y = relu?(scale*x + shift [+ res]); relu_mask (nullable) as in dc_bn_relu_fwd.  HW % 4 == 0 */
int dc_bn_apply(const float* x, const float* res, const float* scale, const float* shift, float* y, void* relu_mask,
                int N, int C, int HW, int relu, int groups, void* stream);
/* coef: (groups, C, 4) floats; dgamma, dbeta (C) nullable */
int dc_bn_bwd_finalize(const float* part, int nparts, int ppg, double count, const float* gamma, const float* mean,
                       const float* invstd, float* coef, float* dgamma, float* dbeta, int C, int groups, void* stream);
/* dx = a*gp + b*(x - mean) + c0 with gp the MASKED upstream gradient (which is also the residual input's gradient) */
int dc_bn_bwd_apply(const float* x, const float* gp, const float* coef, float* dx, int N, int C, int HW, int groups,
                    void* stream);

/* nn.MaxPool2d(3, 2, 1) of the ResNet stem (networks/resnet_encoder.py:93).  x (NC planes of HxW) ->
 * y (NC planes of Ho x Wo, Ho = (H-1)/2+1) and `code` (one byte per output: window position of the first
 * maximum, ATen's tie-break).  NC <= 65535.  Backward: gather, no atomics. */
int dc_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* code, int NC, int H, int W, void* stream);
int dc_maxpool3x3s2_bwd(const float* gy, const uint8_t* code, float* dx, int NC, int H, int W, void* stream);
/* dx = max-pool backward + addend (same shape as dx, nullable): the pooled tensor's other consumer's gradient -- the stem's
 * output feeds the pool AND the depth decoder's skip connection (networks/depth_decoder.py:57-59) -- added on the way out
 * instead of by an elementwise pass of autograd. */
int dc_maxpool3x3s2_bwd_add(const float* gy, const uint8_t* code, float* dx, const float* addend, int NC, int H, int W,
                            void* stream);

/* Stride-1 3x3 convolution with zero padding 1 and no bias -- the conv3x3 of the ResNet trunks (reference
 * networks/resnet_encoder.py:74-98 -> torchvision BasicBlock/Bottleneck) -- as a fused Winograd F(2x2,3x3) on the
 * fp32 matrix cores.  x (B,Ci,H,W), weight (Co,Ci,3,3), y (B,Co,H,W); dgrad: gy (B,Co,H,W) -> gx (B,Ci,H,W).
 * W must be even.  ws: dc_wino3x3_workspace bytes (transformed weights, rebuilt on every call, and the partial
 * outputs of the channel-split used on small maps, summed in fixed order).
 * Result differs from a direct fp32 convolution by the usual Winograd rounding (~1e-6 relative). */
size_t dc_wino3x3_workspace(int B, int Ci, int Co, int H, int W);
int dc_wino3x3_fwd(const float* x, const float* weight, float* y, void* ws, int B, int Ci, int Co, int H, int W,
                   void* stream);
int dc_wino3x3_dgrad(const float* gy, const float* weight, float* gx, void* ws, int B, int Ci, int Co, int H, int W,
                     void* stream);
/* gx = data gradient + addend (addend: same shape as gx, may be NULL).  For a tensor with two consumers -- the input of a
 * residual block feeds conv1 AND the skip connection (torchvision BasicBlock / Bottleneck behind
 * networks/resnet_encoder.py:74-98) -- autograd sums the two gradients in a separate pass; here the skip's gradient is
 * added in the store epilogue of conv1's data-gradient kernel. */
int dc_wino3x3_dgrad_add(const float* gy, const float* weight, float* gx, const float* addend, void* ws, int B, int Ci, int Co,
                         int H, int W, void* stream);

/* The trunk's 3x3 convolution with a BatchNorm folded in (dc_bn_fold; `bn` nullable = the plain calls): forward with the
 * BatchNorm + ReLU of its input in the loader and / or the statistics epilogue, data gradient with the BatchNorm-backward
 * epilogue.  dc_wino3x3_bn_ok(): both passes of this shape take the fold (fp32 policy, <= 2 groups, unsplit reduction);
 * the *_parts queries size the partial buffers (0 = no epilogue: split reduction on a small map, bf16 policy). */
int dc_wino3x3_bn_ok(int B, int Ci, int Co, int H, int W, int groups);
int dc_wino3x3_stat_parts(int B, int Ci, int Co, int H, int W, int groups, int* ppg);
int dc_wino3x3_bwd_parts(int B, int Ci, int Co, int H, int W, int groups, int* ppg);
int dc_wino3x3_fwd_bn(const float* x, const float* weight, float* y, void* ws, int B, int Ci, int Co, int H, int W,
                      const dc_bn_fold* bn, void* stream);
int dc_wino3x3_dgrad_bn(const float* gy, const float* weight, float* gx, const float* addend, void* ws, int B, int Ci, int Co,
                        int H, int W, const dc_bn_fold* bn, void* stream);
int dc_wino3x3_wgrad_bn(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int H, int W,
                        const dc_bn_fold* bn, void* stream);

/* Measurement hook for the convolution kernels (bench.py `roofline`): when enabled, every Winograd launch brackets its
 * main kernel with hipEvents on the launch stream.  kind 0 = wino_ps_kernel (forward / data gradient of the trunk and
 * decoder convolutions), 1 = wino_wgrad_kernel.  collect returns the summed kernel milliseconds, the summed
 * algorithmic FLOPs (2 MAC of the direct convolution), the FLOPs actually issued to the matrix cores
 * (16 Winograd-domain GEMMs including tile padding), the algorithmic bytes (each operand and the result once) and the
 * launch count since the last enable / collect.
 * `every` > 1 samples every n-th launch of a kind only (an event pair per launch costs ~4 % of a training step). */
int dc_conv_profile_enable(int max_launches, int every);
int dc_conv_profile_collect(int kind, double* ms, double* algorithmic_flops, double* executed_flops,
                            double* algorithmic_bytes, int* launches);

/* Weight gradient of the same convolution, also in the Winograd domain (16 GEMMs reduced over all 2x2 tiles of the
 * batch, split over blocks and summed in fixed order -- deterministic, no atomics).  x (B,Ci,H,W), gy (B,Co,H,W)
 * -> dweight (Co,Ci,3,3), overwritten.  W even.  ws: dc_wino3x3_wgrad_workspace bytes. */
size_t dc_wino3x3_wgrad_workspace(int B, int Ci, int Co, int H, int W);
int dc_wino3x3_wgrad(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int H, int W,
                     void* stream);

/* 1x1 convolution without bias, stride 1 or 2 (the trunks' `downsample` branches and Bottleneck conv1 / conv3,
 * networks/resnet_encoder.py:74-98 via torchvision), as an fp32-MFMA GEMM on the NCHW tensors: x (B,Ci,Hi,Wi),
 * weight (Co,Ci), y / gy (B,Co,Hi/stride,Wi/stride).  stride 2 needs even Hi, Wi.  dgrad writes every element of dx
 * (zeros where the stride skips).  wgrad: split reduction, fixed-order sum, ws = dc_conv1x1_wgrad_workspace bytes.
 * Shapes with Ci % 4 == 0 and (Hi/stride * Wi/stride) % 4 == 0 (stride 2: also Wi/stride % 4 == 0) whose reduction
 * extent (forward: Ci, data gradient: Co, weight gradient: B * pixels) is a multiple of 32 run on tiled kernels (up to
 * 128 x 128 outputs per block, 16-byte staging, pixels flattened over the batch); others on a general 64 x 64 kernel.
 * All variants are deterministic. */
int dc_conv1x1_fwd(const float* x, const float* weight, float* y, int B, int Ci, int Co, int Hi, int Wi, int stride, void* stream);
int dc_conv1x1_dgrad(const float* gy, const float* weight, float* dx, int B, int Ci, int Co, int Hi, int Wi, int stride,
                     void* stream);
/* dx = data gradient + addend (see dc_wino3x3_dgrad_add); in the store epilogue of the tiled GEMM for stride 1, a separate
 * in-place pass otherwise. */
int dc_conv1x1_dgrad_add(const float* gy, const float* weight, float* dx, const float* addend, int B, int Ci, int Co, int Hi, int Wi,
                         int stride, void* stream);
/* ... + addend + addend2: an input with three consumers (a stage's first block: the 3x3 / 2 convolution, the 1x1 / 2 `downsample`
 * and the decoder's skip connection).  At stride 2 the tiled kernel writes every cell of dx -- values and the skipped zeros --
 * once, so the addends ride in that store. */
int dc_conv1x1_dgrad_add2(const float* gy, const float* weight, float* dx, const float* addend, const float* addend2, int B, int Ci,
                          int Co, int Hi, int Wi, int stride, void* stream);
size_t dc_conv1x1_wgrad_workspace(int B, int Ci, int Co, int Hi, int Wi, int stride);
int dc_conv1x1_wgrad(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int Hi, int Wi, int stride,
                     void* stream);

/* The 1x1 convolution with a BatchNorm folded in (dc_bn_fold above; `bn` nullable = the plain calls).  Tiled-kernel shapes
 * only: dc_conv1x1_bn_ok() tells whether all three passes of a stride-1 shape take the fold, the *_parts queries size the
 * partial buffers of the two epilogues (0 = not on this shape / group layout). */
int dc_conv1x1_bn_ok(int B, int Ci, int Co, int Hi, int Wi);
int dc_conv1x1_stat_parts(int B, int Ci, int Co, int Hi, int Wi, int stride, int groups, int* ppg);
int dc_conv1x1_bwd_parts(int B, int Ci, int Co, int Hi, int Wi, int groups, int* ppg);
int dc_conv1x1_fwd_bn(const float* x, const float* weight, float* y, int B, int Ci, int Co, int Hi, int Wi, int stride,
                      const dc_bn_fold* bn, void* stream);
int dc_conv1x1_dgrad_bn(const float* gy, const float* weight, float* dx, const float* addend, int B, int Ci, int Co, int Hi, int Wi,
                        int stride, const dc_bn_fold* bn, void* stream);
int dc_conv1x1_wgrad_bn(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int Hi, int Wi, int stride,
                        const dc_bn_fold* bn, void* stream);

/* The same GEMMs on the bf16 matrix cores with SPLIT fp32 operands (csrc/gemm1x1_x3.hip): every fp32 operand is the sum of three
 * bf16 pieces and the six partial products down to 2^-16 |a||b| are accumulated in fp32 -- fp32 inputs, fp32 outputs, an error
 * below that of an fp32 multiply-add chain of the same length, at 6/16 of the fp32 matrix time.  Shapes: dc_gemm1x1x3_*_ok
 * (pixels per image a multiple of 16, reduction extent a multiple of 32, >= 64 output rows; the data gradient at stride 1);
 * ws = dc_gemm1x1x3_workspace(Ci, Co) bytes, 16-byte aligned (the split weights of the launch; weights registered with the
 * per-step weight cache, dc_wino_cache_*, are split once per step by its refresh instead).  dc_set_gemm_split (default 1; env
 * DC_G1_X3): whether callers that can supply the workspace (depthcore/ops.py, depthcore/bnfold.py) take this path where it
 * applies; 0 keeps the fp32-MFMA kernels (the A/B); 3 = 1 + the forward and weight gradient of the 3x3 / 2 trunk convolutions
 * (dc_convs2_*) on the same kernels with gather loaders -- an experiment: more accurate, slower on most shapes, not the default.
 * Accumulation: each 32-deep chunk's six products are summed from zero inside the matrix pipe and enter the fp32 accumulator through
 * ONE round-to-nearest add, with alternating operand signs per chunk -- v_mfma_f32_16x16x32_bf16 truncates its fp32 result toward
 * -inf (tools/diag_x3_bias.py), which biased long reductions before; now the error against fp64 is 0.3 - 0.5x the fp32-MFMA kernels'. */
int dc_set_gemm_split(int mode);
int dc_get_gemm_split(void);
int dc_gemm1x1x3_fwd_ok(int B, int Ci, int Co, int Hi, int Wi, int stride);
int dc_gemm1x1x3_dgrad_ok(int B, int Ci, int Co, int Hi, int Wi, int stride);
size_t dc_gemm1x1x3_workspace(int Ci, int Co);
int dc_gemm1x1x3_fwd(const float* x, const float* weight, const float* bias, float* y, void* ws, int B, int Ci, int Co, int Hi, int Wi,
                     int stride, int act, void* stream);
int dc_gemm1x1x3_dgrad(const float* gy, const float* weight, float* dx, void* ws, const float* addend, const float* addend2, int B,
                       int Ci, int Co, int Hi, int Wi, int stride, void* stream);
/* ... with a BatchNorm folded in (dc_bn_fold; `bn` nullable): the contract of dc_conv1x1_*_bn on the split kernels -- loader
 * relu(scale x + shift) in the forward and the weight gradient, statistics epilogue of the forward, BatchNorm-backward epilogue of
 * the data gradient; the *_parts queries size the partial buffers for THESE kernels' tiles. */
int dc_gemm1x1x3_bn_ok(int B, int Ci, int Co, int Hi, int Wi);
int dc_gemm1x1x3_stat_parts(int B, int Ci, int Co, int Hi, int Wi, int stride, int groups, int* ppg);
int dc_gemm1x1x3_bwd_parts(int B, int Ci, int Co, int Hi, int Wi, int groups, int* ppg);
int dc_gemm1x1x3_fwd_bn(const float* x, const float* weight, const float* bias, float* y, void* ws, int B, int Ci, int Co, int Hi, int Wi,
                        int stride, int act, const dc_bn_fold* bn, void* stream);
int dc_gemm1x1x3_dgrad_bn(const float* gy, const float* weight, float* dx, void* ws, const float* addend, const float* addend2, int B,
                          int Ci, int Co, int Hi, int Wi, int stride, const dc_bn_fold* bn, void* stream);
int dc_gemm1x1x3_wgrad_bn(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int Hi, int Wi, int stride,
                          const dc_bn_fold* bn, void* stream);
/* weight gradient: both operands split while staging, reduction split over blocks into fp32 slabs (ws = dc_gemm1x1x3_wgrad_workspace
 * bytes, 16-byte aligned), summed in fixed order: deterministic */
int dc_gemm1x1x3_wgrad_ok(int B, int Ci, int Co, int Hi, int Wi, int stride);
size_t dc_gemm1x1x3_wgrad_workspace(int B, int Ci, int Co, int Hi, int Wi, int stride);
int dc_gemm1x1x3_wgrad(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int Hi, int Wi, int stride,
                       void* stream);

/* The same convolution with bias and activation fused into the epilogue: y = act(conv1x1(x) + bias), act as in
 * dc_conv3x3_fwd (0 none, 1 ELU, 2 sigmoid, 3 ReLU, 4 tanh); bias may be NULL.  This is `relu(squeeze(f))` and the final
 * `pose_2` convolution of networks/pose_decoder.py:25,30,40-48.
 * Backward: dc_bias_act_bwd turns gy into the gradient of the pre-activation, gpre = gy * act'(y) (gpre may alias gy, or
 * be NULL when only dbias is wanted), and reduces dbias[c] = sum_{b,p} gpre (fixed-order, deterministic; dbias may be
 * NULL); gpre then feeds dc_conv1x1_dgrad / dc_conv1x1_wgrad.  y, gy, gpre: (B,C,P). */
int dc_conv1x1_bias_act_fwd(const float* x, const float* weight, const float* bias, float* y, int B, int Ci, int Co, int Hi, int Wi,
                            int stride, int act, void* stream);
int dc_bias_act_bwd(const float* y, const float* gy, float* gpre, float* dbias, int B, int C, int P, int act, void* stream);

/* The 7x7 / 2 stem reading the RAW frames: `x = (input_image - 0.45) / 0.225` (networks/resnet_encoder.py:89) and, for the
 * pose encoder, the temporal pair concat `torch.cat([f_a, f_b], 1)` of trainer.py:398-412 are index arithmetic of the
 * kernels' patch loader (the same two IEEE operations per pixel; conv1's zero padding stays zero), so neither the
 * normalised image nor the 6-channel pair tensor exists in HBM.
 *   frames: HOST array of nf device pointers, each (Bf,3,Hi,Wi).  nf = 1: Ci = 3, output batch Bf.  nf = 3: the two pairs
 *   (f0,f1), (f1,f2) stacked along the batch -- output batch 2*Bf, item b < Bf = cat(f0[b], f1[b]), item Bf + b = cat(f1[b],
 *   f2[b]) (= trainer.py's pairs (-1,0), (0,+1) for frames (f-1, f0, f+1)).
 *   weight (64, 3 or 6, 7, 7); y / gy (Bf or 2*Bf, 64, Hi/2, Wi/2); ws: dc_convs2_fwd_workspace / _wgrad_workspace bytes of
 *   the equivalent dc_convs2_* call (B = output batch, Ci, ksize 7).  Same arithmetic as dc_convs2_fwd / _wgrad on the
 *   materialised input, bit for bit.  No data gradient (the input is the image). */
int dc_stem_supported(int nf, int Bf, int Co, int Hi, int Wi);
int dc_stem_fwd(const float* const* frames, int nf, float mean, float stdv, const float* weight, float* y, void* ws, int Bf, int Hi,
                int Wi, int Co, void* stream);
int dc_stem_wgrad(const float* const* frames, int nf, float mean, float stdv, const float* gy, float* dweight, void* ws, int Bf, int Hi,
                  int Wi, int Co, void* stream);

/* Adam update of all trainable tensors of a step (reference trainer.py:110-113 `optim.Adam(self.parameters_to_train,
 * self.opt.learning_rate)` and `self.model_optimizer.step()` at trainer.py:238): torch.optim.Adam arithmetic in fp32 -- no
 * weight decay, no amsgrad -- one streaming pass, 28 bytes per parameter.
 *   slots_dev   device array of ntensors records {float* param, float* exp_avg, float* exp_avg_sq, float* step, int64 numel}
 *               (40 bytes each; `step` points at the tensor's step count, a float in device memory, as torch's capturable
 *               Adam keeps it);
 *   chunks_dev  device array of int32 pairs {tensor, first element}: every tensor cut into pieces of dc_adam_chunk()
 *               elements, sorted by tensor;  chunk_start_host[t] = index of tensor t's first pair (ntensors + 1 entries, host);
 *   grads_host  host array of ntensors device pointers: this step's gradients (contiguous fp32), NULL = the tensor got no
 *               gradient: it is skipped and its step count does not advance.
 * beta1 / beta2 are doubles: the scalar factors (1 - beta, the bias corrections) are formed in double as torch forms them.
 * Both tables are static for a model; only grads_host changes per step.  Capturable in a hipGraph (nothing step-dependent
 * is baked into the launch except the gradient addresses and lr). */
int dc_adam_chunk(void);
int dc_adam_step(const void* slots_dev, const void* chunks_dev, const int* chunk_start_host, const void* const* grads_host,
                 int ntensors, float lr, double beta1, double beta2, float eps, void* stream);

/* The strided convolutions of the trunks (networks/resnet_encoder.py:87-98 via torchvision): ksize 7 = the 7x7 / 2 stem
 * (padding 3, Ci = 3 or 6), ksize 3 = the 3x3 / 2 first convolution of layer2-4 (padding 1); no bias.  Implicit GEMMs on
 * the fp32 matrix cores straight on NCHW (no im2col tensor, no layout transposes), exact fp32 products, deterministic.
 * x (B,Ci,Hi,Wi) with even Hi, Wi and (Wi/2) % 4 == 0 (dc_convs2_supported); weight (Co,Ci,k,k); y / gy
 * (B,Co,Hi/2,Wi/2).  dgrad exists for ksize 3 (Ci % 4 == 0, Co % 32 == 0); the stem's input is the image.
 * ws: the matching *_workspace bytes (re-laid-out weights / split-reduction slabs). */
int dc_convs2_supported(int B, int Ci, int Co, int Hi, int Wi, int ksize);
size_t dc_convs2_fwd_workspace(int B, int Ci, int Co, int Hi, int Wi, int ksize);
int dc_convs2_fwd(const float* x, const float* weight, float* y, void* ws, int B, int Ci, int Co, int Hi, int Wi, int ksize,
                  void* stream);
size_t dc_convs2_dgrad_workspace(int B, int Ci, int Co, int Hi, int Wi, int ksize);
int dc_convs2_dgrad(const float* gy, const float* weight, float* dx, void* ws, int B, int Ci, int Co, int Hi, int Wi, int ksize,
                    void* stream);
size_t dc_convs2_wgrad_workspace(int B, int Ci, int Co, int Hi, int Wi, int ksize);
int dc_convs2_wgrad(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int Hi, int Wi, int ksize,
                    void* stream);

/* Any other nn.Conv2d shape of the trunks (networks/resnet_encoder.py:74-98 via torchvision: square kernel, symmetric stride
 * and zero padding, no groups / dilation): odd or tiny maps, a 1x1 / 2 on an odd map, a stem whose input needs a gradient.
 * Plain direct kernels (one thread per output element, fixed-order sums) so that NO shape reaches the framework's
 * convolution on the GPU; not a fast path, no BASELINE configuration uses it.
 * x (B,Ci,Hi,Wi); weight (Co,Ci,k,k), 1 <= k <= 11; 1 <= stride <= 4; 0 <= pad < k; y / gy (B,Co,Ho,Wo) with
 * Ho = (Hi + 2 pad - k) / stride + 1; bias / dbias (Co) or NULL.  DC_EINVAL outside these ranges. */
int dc_conv2d_direct_fwd(const float* x, const float* weight, const float* bias, float* y, int B, int Ci, int Co, int Hi, int Wi,
                         int ksize, int stride, int pad, void* stream);
int dc_conv2d_direct_dgrad(const float* gy, const float* weight, float* dx, int B, int Ci, int Co, int Hi, int Wi, int ksize,
                           int stride, int pad, void* stream);
int dc_conv2d_direct_wgrad(const float* x, const float* gy, float* dweight, float* dbias, int B, int Ci, int Co, int Hi, int Wi,
                           int ksize, int stride, int pad, void* stream);

/* ------------------------------------------------------------------ f1 Fusion_v3 front-end */
/* AttentionConv of networks/fusion_v2.py:46-98 as instantiated by ResidualAttentionUnit (:101-137): kernel 3, stride 1,
 * padding 1, groups 1, bias=True, C = 2 or 4 channels.  One fused kernel per direction (no q / k / v / unfold / softmax
 * tensors in HBM):
 *   r = relu_in ? relu(x) : x;   q = Wq r + bq;   k, v = 1x1 convolutions of the zero-padded r;
 *   y[c] = sum_t softmax_t( q[c] * (k_t[c] + rel_t[c]) ) * v_t[c]   (+ res, or relu(res) when relu_res)
 * rel_t = rel_h[dy] for c < C/2, rel_w[dx] otherwise.  relu_in / relu_res reproduce the unit's in-place ReLUs
 * (fusion_v2.py:130-136: the skip connection adds the ReLU'd input).
 * Weights are (C,C) row-major [out][in] (the (C,C,1,1) conv weights), biases (C), rel_h / rel_w 3 floats.
 *
 * The input channels are GATHERED from up to C tensors through a dc_attn_map, so the `torch.cat`s of
 * FeatureFusionBlock_v3.forward (fusion_v2.py:309-313) and the PixelShuffle of UpscalePS (:226-236) never exist in memory:
 * channel c of pixel (b, y, x) is ptr[c][b * batch_stride[c] + y * W + x] (DC_ATTN_PLAIN: ptr[c] = that channel's plane of
 * batch element 0), or, for DC_ATTN_PIXEL_SHUFFLE2, element ((y&1)*2 + (x&1), y>>1, x>>1) of a (4, H/2, W/2) block at
 * ptr[c] + b * batch_stride[c] (the pre-shuffle output of UpscalePS's convolution; H, W even).  y and gy are plain
 * contiguous (B,C,H,W).
 * Backward: the gradient of input channel c is stored through the map `dx` (same addressing; every element written once),
 * after adding dx_add (plain (B,C,H,W), nullable: the gradient that reaches the same input through the unit's skip
 * connection); `dres` (nullable / ptr[0] NULL) receives gy masked by relu_res;
 * dparams = dc_attnconv_param_count(C) floats laid out [dWq (C*C), dbq (C), dWk, dbk, dWv, dbv, drel_h (3), drel_w (3)],
 * overwritten; deterministic (per-block partial rows in ws, summed in fixed order).  ws: dc_attnconv_bwd_workspace bytes. */
enum { DC_ATTN_PLAIN = 0, DC_ATTN_PIXEL_SHUFFLE2 = 1 };
typedef struct dc_attn_map {
    const float* ptr[4];
    long long batch_stride[4];   /* elements */
    int mode[4];
} dc_attn_map;
typedef struct dc_attn_params {
    const float *wq, *bq, *wk, *bk, *wv, *bv, *rel_h, *rel_w;
} dc_attn_params;
int dc_attnconv_fwd(const dc_attn_map* x, const dc_attn_params* p, const dc_attn_map* res, float* y, int B, int C, int H, int W,
                    int relu_in, int relu_res, void* stream);
int dc_attnconv_param_count(int C);
size_t dc_attnconv_bwd_workspace(int B, int C, int H, int W);
int dc_attnconv_bwd(const dc_attn_map* x, const dc_attn_params* p, const dc_attn_map* res, const float* gy, const dc_attn_map* dx,
                    const float* dx_add, const dc_attn_map* dres, float* dparams, void* ws, int B, int C, int H, int W, int relu_in,
                    int relu_res, void* stream);

/* ------------------------------------------------------------------ f2 ConvGRU temporal fusion (gru_version v5) */
/* networks/rnn.py:101-143 `ConvGRUCell`: the two 3x3 convolutions are dc_conv3x3_fwd calls with (x, h) / (x, r*h) as the two
 * concatenated sources, zero padding, bias, sigmoid / tanh epilogue; these entry points are the gate arithmetic between
 * them.  gates (B,2C,P) = [reset r | update u]; h, cnm, rh, h_next (B,C,P).
 *   dc_gru_rh_*:     rh = r * h                        backward: d_gates = [g * h | 0], d_h = g * r
 *   dc_gru_blend_*:  h_next = (1 - u) * h + u * cnm    backward: d_gates = [0 | g * (cnm - h)], d_h = g * (1 - u), d_cnm = g * u
 * (each backward writes all of its d_gates; autograd sums the two.)
 *   dc_gru_residual_*: out[i] = f[i] + (H[i+1] + H[i]) / 2 over a sequence of n frames, H = its n+1 hidden states stacked
 *   (trainer_gru.py:637-639); f, out (n, M), H (n+1, M); backward: d_f = g, d_H[j] = (g[j] + g[j-1]) / 2. */
int dc_gru_rh_fwd(const float* gates, const float* h, float* rh, int B, int C, int P, void* stream);
int dc_gru_rh_bwd(const float* gates, const float* h, const float* g, float* d_gates, float* d_h, int B, int C, int P, void* stream);
int dc_gru_blend_fwd(const float* gates, const float* h, const float* cnm, float* h_next, int B, int C, int P, void* stream);
int dc_gru_blend_bwd(const float* gates, const float* h, const float* cnm, const float* g, float* d_gates, float* d_h, float* d_cnm,
                     int B, int C, int P, void* stream);
/* accumulating forms for a backward that walks a sequence's frames in reverse (depthcore.ops._GruLevel): d_h += instead of =;
 * dc_gru_rh_bwd_acc writes ONLY the reset half of d_gates (the blend backward of the same step wrote the update half) */
int dc_gru_rh_bwd_acc(const float* gates, const float* h, const float* g, float* d_gates, float* d_h, int B, int C, int P, void* stream);
int dc_gru_blend_bwd_acc(const float* gates, const float* h, const float* cnm, const float* g, float* d_gates, float* d_h, float* d_cnm,
                         int B, int C, int P, void* stream);
/* The sequence trainer's stacking of per-frame tensors along the batch (trainer_gru.py:819-821, 886-896, 943-944: torch.cat over
 * the frames at every use) as ONE launch: segment i copies n[i] floats src[i] -> dst[i] (HOST arrays, nseg <= 96). */
int dc_gather_copy(const float* const* src, float* const* dst, const size_t* n, int nseg, void* stream);
int dc_gru_residual_fwd(const float* f, const float* H, float* out, int n, size_t M, void* stream);
int dc_gru_residual_bwd(const float* g, float* d_H, int n, size_t M, void* stream);

/* ------------------------------------------------------------------ transformed-weight cache of the Winograd kernels */
/* ------------------------------------------------------------------ reduced-precision networks (BASELINE configs[4]) */
/* "Networks in reduced precision, loss in fp32" (trainer_fusion_v3.py:311-330 under mixed precision; the reference itself
 * has no autocast -- this is the build's statement of that configuration).  Tensors stay fp32 in HBM (activations, master
 * weights, gradients, BatchNorm statistics, the whole photometric chain); with DC_PREC_BF16 in effect a convolution entry
 * point (dc_conv3x3_*, dc_wino3x3_*, dc_convs2_* 3x3, dc_conv1x1_*) rounds its two matrix operands to bf16 (round to nearest
 * even) on the way into LDS and accumulates in fp32 on v_mfma_f32_16x16x32_bf16.  bf16 keeps fp32's exponent range: no loss
 * scaling.  Shapes outside the bf16 kernels' 16-byte staging (W % 16 != 0, single-channel heads, the 7x7 stem) keep their
 * fp32 kernels.  The setting is PER CALLING THREAD (autograd's backward thread sets its own); returns the previous value,
 * or DC_EINVAL. */
enum { DC_PREC_F32 = 0, DC_PREC_BF16 = 1 };
int dc_set_matrix_precision(int precision);
int dc_get_matrix_precision(void);

/* Every stride-1 3x3 convolution on the Winograd kernels (dc_wino3x3_fwd / _dgrad and the Winograd branch of
 * dc_conv3x3_fwd / _bwd) starts by transforming its filter (U = G g G^T; the data gradient uses the rotated, transposed
 * filter): one small launch in front of every convolution, although the weights only change in the optimiser step.
 *   dc_wino_cache_new_owner(): an id for one model (one Trainer).  Every owner has its OWN descriptor table: a refresh
 *     transforms -- and a hipGraph that captured it replays the transform of -- that owner's weights only, never memory
 *     whose lifetime belongs to another model.
 *   dc_wino_cache_register(owner, weight, Ci, Co): `weight` (Co,Ci,3,3) stays at this address, alive, until the owner is
 *     released.
 *   dc_wino_cache_refresh(owner, stream): ONE launch that transforms every (weight, pass, tile layout) variant the kernels
 *     have asked for so far among the owner's weights; from then on those launches read the cached U.  Call it at the
 *     start of a training step, after the weights were last written, on a stream every consumer stream waits for.
 *   dc_wino_cache_invalidate(owner): the owner's cached transforms are stale (call after the step's backward, before the
 *     optimiser).
 *   dc_wino_cache_release_owner(owner): forget the owner and all its weights.  Tables and buffers are parked, not freed --
 *     a captured hipGraph of the owner may still name them -- until dc_wino_cache_clear().
 *   dc_wino_cache_clear(): drop ALL owners and free every buffer (device-synchronising; only when no captured graph that
 *     used the cache will be replayed again).
 *   dc_wino_cache_variants(): number of cached variants over all owners (diagnostics / tests).
 * The prepared bf16 weights of the bf16 direct kernels (DC_PREC_BF16: [m-block][chunk][tap][k-group][m][8], one ~8 us packing launch
 * in front of every forward / data-gradient call otherwise) are variants of the same registry and ride in the same refresh launch.
 * A convolution whose weight is not registered, or met before the first refresh, transforms in place exactly as before:
 * the cache changes launch counts, never results (the transform is the same device function).  Nothing here allocates,
 * synchronises or copies while `stream` is being captured into a hipGraph: a variant first met inside a capture keeps its
 * per-launch transform, and a refresh inside a capture launches the owner's table as it stood before the capture. */
/* Winograd F(4x4,3x3) for the plain trunk convolutions (dc_wino3x3_fwd / _dgrad / _dgrad_add; the weight gradient and the
 * decoder's fused blocks keep F(2x2,3x3)): 0.5625 of the matrix-core work at ~5x the rounding error (1.1-1.4e-6 relative L2
 * against an fp64 direct convolution; F(2x2,3x3): 2-3e-7).  OFF by default; mode 1 uses it where W % 4 == 0 and its tile
 * groups cover >= 85 % of the map.  Process-global; returns the previous mode, or DC_EINVAL.  Measurements: DESIGN.md 4a. */
int dc_set_wino_f4(int mode);
/* Persistent form of the plain trunk launches of dc_wino3x3_fwd / _dgrad / _dgrad_add that run in more than one round of
 * resident blocks (unsplit reduction, an even number >= 4 of 8-channel chunks): one block per slot walks several work items
 * as ONE software pipeline -- the next item's first loads fly under the current item's row-exchange epilogue.  Same
 * arithmetic in the same order per output: results are bitwise those of the classic launch.  Process-global; the initial
 * mode comes from DC_WINO_PERSIST (default 0); returns the previous mode, or DC_EINVAL.  Measurements: DESIGN.md 4a. */
int dc_set_wino_persist(int mode);
/* Data gradient of the fused blocks on the Winograd kernels (dc_conv3x3_bwd / _bwd_add): 1 (default) writes the interior of the
 * correlation straight to dx0 / dx1 -- concat split, the 2 x 2 sums of an upsampled x0 and the addends in the store epilogue -- and
 * adds what ReflectionPad folds back from the padded ring in one small launch; 0 = the full correlation over the padded domain into a
 * scratch + a fold pass (same sums in another order: results agree to rounding).  Mode 1 takes the new path where it pays (every
 * zero-padded block; under ReflectionPad the wide levels: >= 6000 pixels, <= 64 output channels, a fold pass of >= 16 MB); 2 = wherever it
 * exists (tests).  Process-global; the initial mode comes from DC_DGRAD_SPLIT (default 1); returns the previous mode, or DC_EINVAL. */
int dc_set_dgrad_split(int mode);
int dc_wino_cache_new_owner(void);
int dc_wino_cache_register(int owner, const float* weight, int Ci, int Co);
int dc_wino_cache_release_owner(int owner);
int dc_wino_cache_refresh(int owner, void* stream);
int dc_wino_cache_invalidate(int owner);
int dc_wino_cache_clear(void);
int dc_wino_cache_variants(void);

/* ------------------------------------------------------------------ f4 per-item data step (decoded frames -> `inputs`) */
/* datasets/mono_dataset.py:92-118 `preprocess` and the image part of :139-211 `__getitem__`, for a whole batch of decoded
 * frames resident in HBM as uint8 HWC (n_img images, contiguous).  Byte-exact with Pillow's `Image.resize(LANCZOS)`
 * (= the reference's Image.ANTIALIAS, :57), torchvision's PIL ColorJitter (:73-76, :186-190) and ToTensor (:67).
 *
 * dc_resample_ksize / dc_resample_table (HOST functions, no GPU): Pillow's 8-bit Lanczos coefficient table for one axis
 *   in_size -> out_size: bounds (out_size, 2) int32 [first source index, tap count], kk (out_size, ksize) int32, 22
 *   fractional bits.  Upload both once per (in_size, out_size) and reuse.
 * dc_data_resize_axis: one 8-bit pass along axis 1 (width; `flip` (n_img) uint8 or NULL mirrors the source row first, i.e.
 *   transpose(FLIP_LEFT_RIGHT) of get_color) or axis 0 (height; flip must be NULL).  src (n, Hi, Wi, 3) -> dst with the
 *   axis resized to out_size.  Pillow resizes width first, then height, and skips a pass whose size is unchanged.
 * dc_data_flip: the mirror alone (for a native image that already has the target width).
 * dc_data_jitter: ColorJitter IN PLACE on (n, npix, 3): per image four steps, steps (n, 4) int32 = DC_JITTER_* in execution
 *   order (or DC_JITTER_NONE), params (n, 4) float32 = that step's factor (for DC_JITTER_HUE: the uint8 H shift,
 *   `np.uint8(hue_factor * 255)`, as a float).  sums: (n) uint64 scratch.  steps / params / sums are DEVICE pointers.
 * dc_data_to_tensor: (n, npix, 3) uint8 -> (n, 3, npix) float32, value / 255 (true division). */
#define DC_JITTER_NONE (-1)
#define DC_JITTER_BRIGHTNESS 0
#define DC_JITTER_CONTRAST 1
#define DC_JITTER_SATURATION 2
#define DC_JITTER_HUE 3
int dc_resample_ksize(int in_size, int out_size);
int dc_resample_table(int in_size, int out_size, int* bounds, int* kk);
int dc_data_resize_axis(const uint8_t* src, uint8_t* dst, int n_img, int Hi, int Wi, int out_size, int axis, const int* bounds,
                        const int* kk, int ksize, const uint8_t* flip, void* stream);
int dc_data_flip(const uint8_t* src, uint8_t* dst, int n_img, int H, int W, const uint8_t* flip, void* stream);
int dc_data_jitter(uint8_t* img, int n_img, int npix, const int* steps, const float* params, unsigned long long* sums, void* stream);
int dc_data_to_tensor(const uint8_t* img, float* out, int n_img, int npix, void* stream);
/* The fused form the host uses: both ToTensor outputs of one pyramid level from ONE uint8 image, the jitter chain evaluated
 * per pixel in registers -- color (n,3,npix) from the untouched pixels, color_aug (n,3,npix) through the item's chain
 * (color_aug NULL: only `color`; then steps / params / sums may be NULL).  Two passes (L sum in front of the contrast step,
 * then everything), nothing written in between; results identical to dc_data_jitter + dc_data_to_tensor. */
int dc_data_jitter_to_tensor(const uint8_t* img, float* color, float* color_aug, int n_img, int npix, const int* steps,
                             const float* params, unsigned long long* sums, void* stream);
/* The pixel-interleaved copy the fused photometric kernels gather from (dc_photo_desc.packed), written by the DATA step so that
 * the training step does not repack its three full-resolution frames every iteration:
 *   dc_data_to_rgbx: (n, npix, 3) uint8 -> (n, npix, 4) float32 RGBx, value / 255 (the same true division as ToTensor: the three
 *     colour values are bit-identical to dc_data_to_tensor's), x = 0;   dc_pack_rgbx: (n, 3, npix) float32 -> the same layout, for
 *     callers whose data loader produced the planar tensors (the reference's).  `out` 16-byte aligned. */
int dc_data_to_rgbx(const uint8_t* img, float* out, int n_img, int npix, void* stream);
int dc_pack_rgbx(const float* x, float* out, int n_img, int npix, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEPTHCORE_H */
