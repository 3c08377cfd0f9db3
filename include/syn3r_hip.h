/*
 * syn3r_hip.h — C-ABI of libsyn3r_hip.so: the MI355X (gfx950) hot path of SYN3R.
 *
 * The reference (DecaYale/SYN3R) is pure Python; it has no FFI of its own
 * (SURVEY.md §8b).  Every entry point below states which reference Python
 * function (file:line under /root/reference) it replaces.  A maintainer binds
 * these with ctypes (INTEGRATION.md shows the stubs); syn3r_amd/_lib.py is that
 * binding for this repo.
 *
 * Conventions
 *   - extern "C", plain pointers + sizes, no torch types.
 *   - every pointer is a DEVICE pointer unless the parameter is documented
 *     "host"; small matrices (4x4 poses, 3x3 intrinsics) are passed BY VALUE
 *     through host pointers and copied into kernel arguments.
 *   - the caller owns every buffer; the library never allocates or frees
 *     user tensors.  Scratch comes from a caller-provided workspace whose size
 *     is returned by the matching *_workspace_bytes query.
 *   - every call is asynchronous on the hipStream_t passed as `stream`
 *     (a void* here so that the header needs no HIP include).
 *   - return value: 0 = ok, negative = error (SYN3R_E_*); the message is
 *     available from syn3r_last_error() (thread-local).
 *   - no process-wide settings: the library reads no environment variable and keeps no mutable state
 *     shared between host threads that changes what a call computes or which kernel it launches.  What
 *     state there is: the thread-local error string, the per-thread test hook and split-K workspace
 *     (syn3r_gemm_set_tile, syn3r_gemm_set_splitk_workspace), the opt-in tracer (syn3r_trace_*), handles
 *     the caller creates (syn3r_unet_create), and per-device launch attributes set on first use.
 */
#ifndef SYN3R_HIP_H
#define SYN3R_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SYN3R_OK 0
#define SYN3R_E_INVALID (-1)   /* bad argument (null pointer, non-positive size, unsupported mode) */
#define SYN3R_E_WORKSPACE (-2) /* workspace too small */
#define SYN3R_E_HIP (-3)       /* a HIP runtime call failed */

/* dtype tags for entry points that accept fp16 or fp32 activations */
#define SYN3R_F32 0
#define SYN3R_F16 1

const char* syn3r_last_error(void);
/* library/ABI version: major*10000 + minor*100 + patch */
int syn3r_version(void);
/* name of the GPU architecture the code objects were built for ("gfx950") */
const char* syn3r_arch(void);

/*
 * Per-kernel timing for bench.py's roofline line: while enabled, every kernel
 * launch is bracketed by HIP events recorded on the launch stream.
 * syn3r_trace_report synchronises them, writes one "kernel calls total_ms"
 * line per kernel into buf and clears the trace.  on = 2 additionally puts
 * the contraction shape into the kernel name (per-shape tuning tables).
 * The switch, the filter and the recorded spans form a SESSION owned by the
 * thread that called syn3r_trace_enable; there is no process-wide switch.
 * Another thread records into that session only after syn3r_trace_attach
 * (session) with the pointer syn3r_trace_session() returned on the owner
 * (NULL detaches) - e.g. PyTorch's autograd thread around a backward launch.
 * syn3r_trace_report reports the calling thread's own session.
 */
int syn3r_trace_enable(int on);
void* syn3r_trace_session(void);
int syn3r_trace_attach(void* session);
/* restrict the tracer to kernels whose name contains one of the comma-separated substrings ("" = all):
 * two event records per launch cost a few microseconds, which matters for 30-microsecond kernels */
int syn3r_trace_filter(const char* substrings);
int syn3r_trace_report(char* buf, size_t cap);

/* ------------------------------------------------------------------------
 * Geometry (solver_utils/)
 * ------------------------------------------------------------------------ */

/*
 * inverse_warp + consistency_check_with_depth, fused.
 * Replaces solver_utils/forward_warp.py:187-279 (inverse_warp) and the
 * consistency.py:44-91 call it makes at forward_warp.py:257.
 *
 * Batched over `nb` target views that share one source view:
 *   img          [3,H,W]   f32  source image (closest view)
 *   depth        [H,W]     f32  source depth
 *   depth_pseudo [nb,H,W]  f32  target-view depths
 *   pose12       host [nb,16] f32 row-major  pose1 @ inverse(pose2)   (forward_warp.py:217)
 *   pose21       host [nb,16] f32 row-major  pose2 @ inverse(pose1)   (consistency.py:37 on the way back)
 *   K, Kinv      host [9] f32 row-major      intrinsics and torch.inverse(K) (consistency.py:21)
 *   bandwidth    reprojection bandwidth (20 or 10 in the reference)
 * Outputs (all [nb,...]):
 *   warped_img [nb,3,H,W] f32, warped_depth [nb,H,W] f32,
 *   mask_warp, mask_depth, mask, mask_inv, mask_depth_strict, mask_reproj : [nb,H,W] u8 (0/1, = torch.bool)
 *   warped_masked_img [nb,3,H,W] f32, soft_mask_reproj [nb,H,W] f32,
 *   reproj_error [nb,H,W] f32 (may be NULL; the reference does not return it)
 * workspace: syn3r_inverse_warp_workspace_bytes(nb) bytes (min/max slots).
 */
size_t syn3r_inverse_warp_workspace_bytes(int nb);
int syn3r_inverse_warp(const float* img, const float* depth, const float* depth_pseudo,
                       const float* pose12, const float* pose21, const float* K, const float* Kinv,
                       float bandwidth, int nb, int H, int W,
                       float* warped_img, float* warped_depth, uint8_t* mask_warp, uint8_t* mask_depth,
                       uint8_t* mask, float* warped_masked_img, uint8_t* mask_inv,
                       uint8_t* mask_depth_strict, uint8_t* mask_reproj, float* soft_mask_reproj,
                       float* reproj_error, void* workspace, size_t workspace_bytes, void* stream);

/*
 * consistency_check_with_depth alone (solver_utils/consistency.py:44-91).
 *   T12 = pose2 @ inverse(pose1), T21 = pose1 @ inverse(pose2) (host, [16] f32 row-major)
 *   K1inv = inverse(K1); K1, K2 host [9].
 *   depth1, depth2 [H,W] f32 -> err [H,W] f32
 */
int syn3r_reproj_error(const float* depth1, const float* depth2, const float* T12, const float* T21,
                       const float* K1, const float* K1inv, const float* K2, int H, int W, float* err,
                       void* stream);

/*
 * Orchestrator post-processing of a batch of inverse warps, on the device.
 * Replaces the per-frame host code of `warp_images_bw` (model/diffusionGS.py:1447-1483): binary mask
 * 1 - mask_reproj >= 0.5, cv2.dilate 5x5 (default border), cond = uint8(warped * (1 - ero)) / 255 (the uint8
 * round trip of :1469,1475), the (h, H/h, w, W/w) block-mean pooling of the hard mask (threshold 0.2) and of
 * the soft reprojection uncertainty 1 - soft_mask_reproj.
 *   mask_reproj [n,H,W] u8, warped_img [n,3,H,W] f32 (0..255), soft_mask_reproj [n,H,W] f32 (outputs of
 *   syn3r_inverse_warp) -> ero [n,H,W] u8 (0/1), cond_image, cond_ori [n,H,W,3] f32, soft [n,H,W] f32,
 *   masks, soft_pool [n,h,w] f32.  H % h == 0, W % w == 0.
 */
int syn3r_warp_post(const uint8_t* mask_reproj, const float* warped_img, const float* soft_mask_reproj,
                    int n, int H, int W, int h, int w, uint8_t* ero, float* cond_image, float* cond_ori,
                    float* soft, float* masks, float* soft_pool, void* stream);

/*
 * Uncertainty fusion + condition-image selection (model/diffusionGS.py:821-862):
 *   conf = exp(-(|cond_ori - gs|_2 / 0.5)^3) * (sum_c cond_ori > 0);  u = 1 - conf * (1 - soft);
 *   cond_image = clip(u > 0.5 ? gs : cond_ori, 0, 1);  masks = block mean of u.
 *   cond_ori, gs_images [n,H,W,3] f32, soft [n,H,W] f32 -> uncertainty [n,H,W], cond_image [n,H,W,3], masks [n,h,w].
 */
int syn3r_fuse_uncertainty(const float* cond_ori, const float* gs_images, const float* soft, int n, int H, int W,
                           int h, int w, float* uncertainty, float* cond_image, float* masks, void* stream);

/*
 * forward_warp: depth-weighted bilinear splat.
 * Replaces solver_utils/forward_warp.py:141-182 (forward_warp),
 * :7-38 (compute_transformed_points) and :42-127 (bilinear_splatting).
 * float64 throughout, as the numpy reference.
 *   frame1 [H,W,3] f64 (0..255), mask1 [H,W] u8 or NULL, depth1 [H,W] f64
 *   T host [16] f64 = transformation2 @ inv(transformation1); K1inv, K2 host [9] f64
 * Outputs: warped [H,W,3] u8, mask2 [H,W] u8, flow12 [H,W,2] f64
 * workspace: syn3r_forward_warp_workspace_bytes(H,W) (f64 accumulators (H+2)x(W+2)x4 + scalars).
 * Accumulation uses f64 hardware atomics (order-dependent in the last bits; the
 * u8 output is insensitive to that except at exact .5 ties).
 */
size_t syn3r_forward_warp_workspace_bytes(int H, int W);
int syn3r_forward_warp(const double* frame1, const uint8_t* mask1, const double* depth1, const double* T,
                       const double* K1inv, const double* K2, int H, int W, uint8_t* warped, uint8_t* mask2,
                       double* flow12, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Modified Euler scheduler (diffusers/schedulers/scheduling_euler_discrete.py)
 * ------------------------------------------------------------------------ */

/*
 * step_interp (scheduling_euler_discrete.py:633-814), fused:
 * v-prediction x0, per-frame quantile mask (radix select instead of the
 * reference's full sort + host sync), closed-form guidance gradient
 * (SURVEY.md §8a S2), Euler update.
 *   model_output [F,C,h,w]  vdtype (SYN3R_F16 / SYN3R_F32)
 *   sample       [F,C,h,w]  sdtype (upcast to f32 as :711)
 *   cond         [F,C,h,w]  f32 = temp_cond_latents[1]
 *   mask         [F-2,C,h,w] f32 (the reference's `mask`; valid = (1-mask)>0.5)
 *   lambda_row   host [F] f64 = lambda_ts[step_i]  (F <= 64)
 *   sigma f32 = sigmas[step_i]; dt f32 = sigmas[step_i+1] - sigma (:800)
 *   c_out f32 = -sigma/(sigma^2+1)^0.5, denom f32 = sigma^2+1, sqrt_sigma f32 = sigma^0.5,
 *                 all computed by the caller with the reference's own 0-dim fp32
 *                 CPU tensor arithmetic (:728,:792) so that the roundings agree
 * Outputs:
 *   prev_sample [F,C,h,w] vdtype; pred_x0 [F,C,h,w] f32 (may be NULL);
 *   grad [F,C,h,w] f32 (only when compute_grad; else may be NULL)
 * workspace: syn3r_step_workspace_bytes(F,C,h,w).
 */
size_t syn3r_step_workspace_bytes(int F, int C, int h, int w);
int syn3r_step_interp(const void* model_output, int vdtype, const void* sample, int sdtype,
                      const float* cond, const float* mask, const double* lambda_row, float sigma,
                      float dt, float c_out, float denom, float sqrt_sigma, float lr, int compute_grad,
                      void* prev_sample, float* pred_x0, float* grad, int F, int C, int h, int w,
                      void* workspace, size_t workspace_bytes, void* stream);

/*
 * step_interp_prob_uncertain (scheduling_euler_discrete.py:1343-1515):
 * same quantile, soft replacement of x0 by the conditioning latents, first
 * and last frame hard-set, Euler update.  Arguments as syn3r_step_interp.
 */
int syn3r_step_replace(const void* model_output, int vdtype, const void* sample, int sdtype,
                       const float* cond, const float* mask, const double* lambda_row, float sigma,
                       float dt, float c_out, float denom, void* prev_sample, float* pred_x0,
                       int F, int C, int h, int w, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Gaussian rasteriser (diff-gaussian-rasterization-confidence)
 *
 * The reference consumes it through gsTrainer.render_view(cam) ->
 * {'render','depth','alpha'} (model/diffusionGS.py:154,166) and differentiates
 * it inside gsTrainer.training()/finetune() (:139,1640).  Its CUDA source is an
 * un-vendored submodule: these entry points follow the
 * published 3DGS rasteriser interface (forward: preprocess + tile binning +
 * radix sort + blend; backward: blend backward + preprocess backward).
 *
 * Layouts: means3D [N,3], scales [N,3], rotations [N,4] (r,x,y,z), opacities [N],
 * shs [N,sh_coeffs,3], confidence [N] or NULL (=1); all f32, device.
 * viewmatrix / projmatrix: host [16] f32, column-major (the transposed
 * world_view_transform / full_proj_transform of FSGS cameras); campos host [3].
 * State buffers (geom, binning, image) are caller-owned byte buffers sized by
 * the *_bytes queries; they carry the forward's intermediates to the backward.
 * ------------------------------------------------------------------------ */
size_t syn3r_raster_geom_bytes(int N);
size_t syn3r_raster_image_bytes(int H, int W);
size_t syn3r_raster_binning_bytes(long long P);

/*
 * Stage 1: project every Gaussian and count the tiles it touches (shapes that
 * take stage 2's pair-sort path also argsort the Gaussians by depth here).
 * radii [N] i32 out.  If num_rendered_host != NULL the
 * number of (Gaussian, tile) pairs P is summed, copied to it and the stream is
 * synchronised (the reference implementation performs the same device->host
 * read to size its binning buffers); with NULL nothing synchronises and stage 2
 * counts the pairs on the device against the capacity it is given.
 */
int syn3r_raster_preprocess(int N, int sh_degree, int sh_coeffs, const float* means3D, const float* scales,
                            const float* rotations, const float* opacities, const float* shs,
                            const float* confidence, float scale_modifier, const float* viewmatrix,
                            const float* projmatrix, const float* campos, float tanfovx, float tanfovy, int H,
                            int W, int* radii, void* geom, size_t geom_bytes, long long* num_rendered_host,
                            void* stream);
/*
 * The same stage on the trainer's PARAMETERS: log_scales [N,3], raw_rotations [N,4] (unnormalised), opacity_logits [N] - the
 * published activations (exp / normalize / sigmoid: GaussianModel.get_scaling / get_rotation / get_opacity, applied before every
 * render inside gsTrainer.training() / finetune(), model/diffusionGS.py:139,1640) happen inside the projection kernel, in
 * syn3r_gaussian_activate's arithmetic: bit for bit the state syn3r_gaussian_activate + syn3r_raster_preprocess leave, one launch
 * and three intermediate tensors less per training iteration.  Pair it with syn3r_raster_backward_raw.
 */
int syn3r_raster_preprocess_raw(int N, int sh_degree, int sh_coeffs, const float* means3D, const float* log_scales,
                                const float* raw_rotations, const float* opacity_logits, const float* shs,
                                const float* confidence, float scale_modifier, const float* viewmatrix,
                                const float* projmatrix, const float* campos, float tanfovx, float tanfovy, int H,
                                int W, int* radii, void* geom, size_t geom_bytes, long long* num_rendered_host,
                                void* stream);

/*
 * Stage 2: the per-tile lists of Gaussians in depth order - entry for entry the
 * result of the published "duplicate (tile<<32 | depth) keys, sort, find tile
 * ranges", built here by filtering the Gaussians through super-tiles of 4x4 or
 * 8x8 tiles (whose short lists are sorted by (depth bits, index) in LDS) and then
 * per tile (csrc/raster_fwd.hip; images with more than 512 super-tiles take the
 * argsort + pair sort) - then the blend.  P is the pair CAPACITY of
 * `binning` (an estimate is fine: a list that does not fit sets the overflow
 * flag in the geometry header and is truncated, never written past the buffer).
 * bg host [3].  out_color [3,H,W], out_depth [1,H,W] (sum of alpha*T*z),
 * out_alpha [1,H,W] (1 - final transmittance).  *point_list_out (host pointer
 * to a device pointer, may be NULL) receives the sorted Gaussian-index list
 * inside `binning`, needed by the backward.
 */
int syn3r_raster_render(int N, int H, int W, const float* bg, const int* radii, void* geom, size_t geom_bytes,
                        void* binning, size_t binning_bytes, void* image, size_t image_bytes, long long P,
                        float* out_color, float* out_depth, float* out_alpha, unsigned** point_list_out,
                        void* stream);

/*
 * Backward of both stages.  dL_dcolor [3,H,W]; dL_ddepth, dL_dalpha [H,W] or NULL.
 * Outputs: dL_dmeans3D [N,3], dL_dscales [N,3], dL_drotations [N,4],
 * dL_dopacities [N], dL_dshs [N,sh_coeffs,3], dL_dmeans2D [N,3] (NDC-space
 * gradient of the projected mean, used by densification), dL_dconfidence [N] or NULL.
 */
size_t syn3r_raster_backward_workspace_bytes(int N);
int syn3r_raster_backward(int N, int sh_degree, int sh_coeffs, long long P, const float* means3D,
                          const float* scales, const float* rotations, const float* opacities, const float* shs,
                          const float* confidence, float scale_modifier, const float* viewmatrix,
                          const float* projmatrix, const float* campos, float tanfovx, float tanfovy, int H, int W,
                          const float* bg, const int* radii, void* geom, size_t geom_bytes,
                          const unsigned* point_list, void* image, size_t image_bytes, const float* dL_dcolor,
                          const float* dL_ddepth, const float* dL_dalpha, float* dL_dmeans3D, float* dL_dscales,
                          float* dL_drotations, float* dL_dopacities, float* dL_dshs, float* dL_dmeans2D,
                          float* dL_dconfidence, void* workspace, size_t workspace_bytes, void* stream);
/*
 * Backward of a forward that started with syn3r_raster_preprocess_raw: the same three parameter tensors in, the gradients
 * with respect to THEM out (syn3r_gaussian_activate_backward's chain rule applied where the activated gradients are formed:
 * d_log_scales = d_scales * scales, d_raw_rotations = (g - qhat (qhat . g)) / max(|q|, 1e-12), d_logits = d_opacities * s (1 - s);
 * the same bits as syn3r_raster_backward followed by syn3r_gaussian_activate_backward).
 */
int syn3r_raster_backward_raw(int N, int sh_degree, int sh_coeffs, long long P, const float* means3D,
                              const float* log_scales, const float* raw_rotations, const float* opacity_logits,
                              const float* shs, const float* confidence, float scale_modifier, const float* viewmatrix,
                              const float* projmatrix, const float* campos, float tanfovx, float tanfovy, int H, int W,
                              const float* bg, const int* radii, void* geom, size_t geom_bytes,
                              const unsigned* point_list, void* image, size_t image_bytes, const float* dL_dcolor,
                              const float* dL_ddepth, const float* dL_dalpha, float* dL_dmeans3D, float* dL_dlog_scales,
                              float* dL_draw_rotations, float* dL_dopacity_logits, float* dL_dshs, float* dL_dmeans2D,
                              float* dL_dconfidence, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Stable LSD radix sort of (u64 key, u32 value) pairs on bits [0, nbits) — the
 * tile/depth sort of the rasteriser (cub::DeviceRadixSort::SortPairs in the
 * published implementation).  Ping-pongs between the two buffer pairs;
 * *result_in_tmp (host) tells which pair holds the sorted output.
 */
size_t syn3r_sort_pairs_workspace_bytes(long long n);
int syn3r_sort_pairs(unsigned long long* keys, unsigned* vals, unsigned long long* keys_tmp, unsigned* vals_tmp,
                     long long n, int nbits, void* workspace, size_t workspace_bytes, int* result_in_tmp,
                     void* stream);

/* ------------------------------------------------------------------------
 * SVD spatio-temporal UNet operators (fp16 storage, fp32 accumulation)
 *
 * Activations are channels-last token matrices: a tensor the reference holds as
 * [B*F, C, h, w] (or [B, C, F, h, w]) is the row-major fp16 matrix
 * [B*F*h*w, C] with row = ((b*F + f)*h + y)*w + x.  With that layout every
 * permute/reshape of the reference's forward
 * (diffusers/models/unets/unet_spatio_temporal_condition.py:356-489,
 * resnet.py:691-721, transformers/transformer_temporal.py:277-379,
 * attention.py:478-533) is an index calculation inside a kernel.
 * ------------------------------------------------------------------------ */

/*
 * out[M,N] = s_acc*(A[M,K] . W[N,K]^T + bias[n] + rowvec[m / rows_per_vec, n])
 *            + s_res*residual[m,n] + s_aux*aux[m,n]
 * nn.Linear / 1x1 convolutions (attention_processor.py:187-202, resnet.py:316,
 * transformer_temporal.py:236,273) with the adds that follow them fused.
 * W is the nn.Linear weight as stored ([out_features, in_features]).
 * rows_per_vec = -P selects rowvec[m mod P] instead (the batch-interleaved context of the temporal
 * cross-attention, transformer_temporal.py:310-317, for a batch of P).  rv_group_rows = G > 0 (with
 * rows_per_vec < 0): the rows come in groups of G, each emulating a SEPARATE batch-of-P call of the reference:
 * row m adds rowvec[(m / G) * P + m mod P] (the forward- and backward-in-time passes of a denoising step,
 * SVD_2pass_prob_uncertain.py:661-742, in one launch).  rv_group_rows = 0: one group.
 * K % 64 == 0; strides in elements, multiples of 8; pointers 16-byte aligned;
 * bias / rowvec / residual / aux may be NULL.
 */
int syn3r_gemm_f16(const void* A, long long lda, const void* W, void* out, long long ldc, const void* bias,
                   const void* rowvec, long long ldrv, int rows_per_vec, int rv_group_rows, const void* residual, long long ldr,
                   const void* aux, long long ldaux, float s_acc, float s_res, float s_aux, int M, int N, int K,
                   void* stream);

/* out[M,N] = [A1 | A2][M, K1+K2] . W[N, K1+K2]^T + bias: the contraction reads the two halves of a channel concatenation
 * in place (resnet.py:316 conv_shortcut applied to torch.cat([hidden, skip], dim=1), unet_3d_blocks.py up blocks), so the
 * concatenated tensor is never written.  K1, K2 % 64 == 0; served by the persistent 256 x 320 kernel only:
 * syn3r_gemm_2src_supported() != 0 is the admission test (otherwise concatenate and call syn3r_gemm_f16). */
int syn3r_gemm_2src_supported(int M, int N, int K1, int K2, long long lda1, long long lda2);
int syn3r_gemm_2src_f16(const void* A1, long long lda1, int K1, const void* A2, long long lda2, int K2, const void* W,
                        void* out, long long ldc, const void* bias, int M, int N, void* stream);

/* Test / tuning hook, PER CALLING THREAD (thread-local: the library holds no state shared between host threads): the
 * contraction kernel family this thread's next launches use.  0 = chosen per shape (default); -128 / -256 = the 160-column
 * LDS-DMA kernel of that block height; -320 = the persistent 256 x 320 kernel wherever it admits the shape; -322 = the
 * software-pipelined persistent 256 x 320 kernel (dense, two-source and implicit-GEMM convolution modes) wherever it admits
 * the shape.  This hook and the split-K workspace below are the library's only settings, both per calling thread; it reads no
 * environment variable (dispatch switches for A/B measurements exist in -DSYN3R_TUNING developer builds only). */
int syn3r_gemm_set_tile(int bm);

/* Split-K scratch, PER CALLING THREAD (round 4).  With a workspace set, the contractions (syn3r_gemm_f16, syn3r_gemm_2src_f16
 * when no part would straddle K1, net.2 of syn3r_feedforward_f16, syn3r_conv2d3x3_f16, syn3r_tconv3_f16) whose tile grid would leave more than half of the CUs idle (level 3 of the SVD UNet at F = 14: 4 032 rows) run
 * as two or four equal K parts - one block per (tile, part), fp32 partial tiles in the workspace - followed by a small
 * launch that sums the parts IN ORDER and applies the epilogue (same arithmetic as the one-pass epilogue; the fp32 sum is
 * associated differently, so results differ from the one-pass launch in the last bits).  Up to 4 * M * N * 4 bytes are used per launch
 * (a launch that needs more runs in one pass); the workspace must stay valid until the launches that used it have completed and
 * must not be shared by launches in flight on different streams.  (NULL, 0) turns split-K off (the default). */
int syn3r_gemm_set_splitk_workspace(void* workspace, size_t bytes);

/*
 * FeedForward's first projection fused with its GEGLU gate (attention.py:608-665, activations.py GEGLU):
 * out[M,D] = h * gelu_erf(g) with [h | g] = A[M,K] . W[2D,K]^T + bias, h and g rounded to fp16 first as
 * the reference's projection output is.  Wpacked / bias_packed hold the rows of W regrouped per tile of
 * 80 output columns as [80 hidden rows | 80 gate rows] (zero rows past D): ceil(D/80)*160 rows
 * (syn3r_amd.unet.ops.pack_geglu).
 */
int syn3r_gemm_geglu_f16(const void* A, long long lda, const void* Wpacked, const void* bias_packed, void* out,
                         long long ldc, int M, int D, int K, void* stream);

/*
 * FeedForward(dim, activation_fn="geglu").forward (diffusers attention.py:608-665, activations.py GEGLU): 
 *   out = s_acc * (geglu(x @ W1^T + b1) @ W2^T + b2) + s_res * residual + s_aux * aux
 * in two launches (syn3r_gemm_geglu_f16, syn3r_gemm_f16) whose intermediate, the gated hidden activation
 * [M, D], lives in `workspace` in a tiled layout ([ceil(M/128)][D/64][128][64]: the second projection reads its A
 * operand as contiguous 16 KB tile images instead of rows at a multi-KB pitch).
 *   x [M, C_in] (row stride ldx); w1_packed / b1_packed: GEGLU packing of syn3r_gemm_geglu_f16 ([80 hidden | 80 gate]
 *   row groups); D = hidden width (multiple of 64); w2 [C_out, D]; b2 [C_out] or NULL; residual / aux [M, C_out] or NULL.
 */
size_t syn3r_feedforward_workspace_bytes(int M, int D);
int syn3r_feedforward_f16(const void* x, long long ldx, const void* w1_packed, const void* b1_packed, int D,
                          const void* w2, const void* b2, void* out, long long ldc, const void* residual,
                          long long ldr, const void* aux, long long ldaux, float s_acc, float s_res, float s_aux,
                          int M, int C_in, int C_out, void* workspace, size_t workspace_bytes, void* stream);

/*
 * syn3r_feedforward_f16 with net.0 on the persistent 256 x 256 tile (round 6, k_gemm_g256: a second look-ahead stage for the
 * A operand, LDS-free epilogue into the tiled hidden activation; attention.py:608-665, activations.py GEGLU).
 *   w1_packed64 / b1_packed64: the rows of net.0.proj regrouped per 64-wide hidden chunk j as [hidden 64j..64j+63 | gate
 *   64j..64j+63] ([2 D, C_in] / [2 D]).  Shapes: syn3r_feedforward_p64_supported(M, D, C_in) (M % 128 == 0, M >= 256,
 *   D % 128 == 0, C_in >= 128 a multiple of 64), x with dense rows (ldx == C_in); everything else as syn3r_feedforward_f16,
 *   whose results it reproduces bit for bit (same accumulation order, same gate).
 */
int syn3r_feedforward_p64_supported(int M, int D, int C_in);
int syn3r_feedforward_p64_f16(const void* x, long long ldx, const void* w1_packed64, const void* b1_packed64, int D,
                              const void* w2, const void* b2, void* out, long long ldc, const void* residual,
                              long long ldr, const void* aux, long long ldaux, float s_acc, float s_res, float s_aux,
                              int M, int C_in, int C_out, void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same FeedForward.forward (attention.py:608-665, GEGLU of activations.py) in ONE kernel for C_in = C_out = 320
 * (the level-0 transformer blocks of SVD): the gated hidden activation never leaves the CU, no workspace.
 *   w1_chunked [D/64][128][320]: rows of net.0.proj regrouped per 64-wide hidden chunk j as
 *       4 x [hidden 64j+16q..+15 | gate 64j+16q..+15], q = 0..3;  b1_chunked [D/64][128] likewise
 *   w2 [320, D] (net.2.weight as stored), b2 [320] or NULL; epilogue as syn3r_gemm_f16 (bias, residual, aux, scales).
 * Other channel counts are rejected (SYN3R_E_INVALID): use syn3r_feedforward_f16.
 */
int syn3r_feedforward_fused_f16(const void* x, long long ldx, const void* w1_chunked, const void* b1_chunked, int D,
                                const void* w2, const void* b2, void* out, long long ldc, const void* residual,
                                long long ldr, const void* aux, long long ldaux, float s_acc, float s_res, float s_aux,
                                int M, int C, void* stream);

/*
 * LayerNorm + FeedForward in that kernel: the `norm3(hidden_states)` -> `ff(...)` pair of BasicTransformerBlock.forward and
 * TemporalBasicTransformerBlock.forward (attention.py:376-392, 519-530) for C = 320.  x is the UN-normalised activation; the
 * kernel normalises its resident 128-row x tile with (ln_gamma, ln_beta [320], ln_eps) - fp32 statistics, the arithmetic and
 * summation order of syn3r_layernorm_f16, so the result equals syn3r_layernorm_f16 followed by syn3r_feedforward_fused_f16
 * bit for bit - and the normalised activation is never written to memory.  `residual` is typically x itself.
 */
int syn3r_feedforward_fused_ln_f16(const void* x, long long ldx, const void* ln_gamma, const void* ln_beta, float ln_eps,
                                   const void* w1_chunked, const void* b1_chunked, int D, const void* w2, const void* b2,
                                   void* out, long long ldc, const void* residual, long long ldr, const void* aux,
                                   long long ldaux, float s_acc, float s_res, float s_aux, int M, int C, void* stream);

/*
 * (hidden + add vector) -> LayerNorm -> FeedForward -> + (hidden + add vector) in that kernel: the head of
 * TemporalBasicTransformerBlock.forward for C = 320 (attention.py:500-517: `residual = hidden_states` after the frame-position
 * embedding was added by the caller, transformer_temporal.py:339; `norm_in`; `ff_in`; `+ residual`).  x is the activation BEFORE
 * the embedding is added, addvec [rows, 320] holds one embedding row per rows_per_vec consecutive rows of x.  The kernel adds the
 * vector to its resident x tile (an fp16 tensor add, rounded as the reference's), normalises it, and its epilogue adds the
 * same fp16 sum as the residual: bit for bit syn3r_layernorm_f16(x, addvec, want the sum) followed by syn3r_feedforward_fused_f16 with
 * that sum as the residual, without the two [M, 320] round trips.  aux / scales as syn3r_feedforward_fused_f16.
 */
int syn3r_feedforward_fused_addln_f16(const void* x, long long ldx, const void* addvec, int rows_per_vec, const void* ln_gamma,
                                      const void* ln_beta, float ln_eps, const void* w1_chunked, const void* b1_chunked, int D,
                                      const void* w2, const void* b2, void* out, long long ldc, const void* aux, long long ldaux,
                                      float s_acc, float s_res, float s_aux, int M, int C, void* stream);

/*
 * LayerNorm + bias-free projection in ONE kernel for C = 320: `norm1(hidden_states)` followed by `attn1.to_q / to_k / to_v`
 * (attention.py:340-352 and 509-512; attention_processor.py to_q / to_k / to_v, stored here as one [960, 320] matrix) of the
 * level-0 transformer blocks.  out [M, N] = LayerNorm(x [M, 320]; ln_gamma, ln_beta, ln_eps) . W[N, 320]^T, N a multiple of 320.
 * The x tile of a block stays on chip (normalised there with the arithmetic of syn3r_layernorm_f16, so the normalised values
 * are the ones that launch would have written), the normalised activation is never written to memory.  Same fp32 accumulation
 * order along K as syn3r_gemm_f16.  Other channel counts are rejected (SYN3R_E_INVALID): syn3r_layernorm_f16 + syn3r_gemm_f16.
 */
int syn3r_layernorm_linear320_f16(const void* x, long long ldx, const void* ln_gamma, const void* ln_beta, float ln_eps,
                                  const void* W, void* out, long long ldc, int M, int N, int C, void* stream);

/*
 * 3x3 Conv2d on NHWC fp16 (resnet.py:274,290; downsampling.py:116-148 with stride 2;
 * upsampling.py:172-183 with upsample != 0: nearest-2x of the input fused into the gather).
 * pad_lo = 1: padding 1 on every side.  pad_lo = 0: the VAE encoder's Downsample2D(padding=0), which pads
 * (0,1,0,1) — nothing before the first row/column, one zero row/column after the last (downsampling.py:140-143).
 * X [NB,Hi,Wi,Cin], W [Cout, 3, 3, Cin] (= the Conv2d weight permuted to OHWI), Cin % 64 == 0.
 * out [NB*Ho*Wo, Cout] with the gemm epilogue (aux excluded); Ho = (Hg + pad_lo - 2) / stride + 1.
 */
int syn3r_conv2d3x3_f16(const void* X, const void* W, void* out, long long ldc, const void* bias,
                        const void* rowvec, long long ldrv, int rows_per_vec, const void* residual, long long ldr,
                        float s_acc, float s_res, int NB, int Hi, int Wi, int Cin, int Cout, int stride, int upsample,
                        int pad_lo, void* stream);

/*
 * (3,1,1) Conv3d over frames, padding (1,0,0) (resnet.py:571-597) on [B,F,HW,Cin] fp16.
 * W [Cout, 3, Cin] (= the Conv3d weight [Cout,Cin,3,1,1] permuted), Cin % 64 == 0.
 */
int syn3r_tconv3_f16(const void* X, const void* W, void* out, long long ldc, const void* bias, const void* rowvec,
                     long long ldrv, int rows_per_vec, const void* residual, long long ldr, float s_acc, float s_res,
                     int B, int F, int HW, int Cin, int Cout, void* stream);

/*
 * Self-attention over the S tokens of each of nseq sequences, head dim 64, scale 1/8
 * (F.scaled_dot_product_attention at attention_processor.py:1279).  q/k/v point at column 0 of
 * head 0 inside row-major matrices with row stride ld (e.g. the three thirds of a fused QKV
 * projection); out [nseq*S, ldo], head hd at columns [64 hd, 64 hd + 64).
 */
int syn3r_attention_f16(const void* q, const void* k, const void* v, long long ld, void* out, long long ldo,
                        int nseq, int S, int heads, void* stream);

/*
 * Temporal self-attention (attention.py:491-508): for every (b, pixel) the F <= 32 tokens at rows
 * (b*F + f)*HW + pixel attend to each other.  Same operand convention as syn3r_attention_f16.
 */
int syn3r_attention_temporal_f16(const void* q, const void* k, const void* v, long long ld, void* out,
                                 long long ldo, int B, int F, int HW, int heads, void* stream);

/*
 * GroupNorm(32 groups) [+ SiLU] on [samples, rows, C] fp16 (statistics per sample and group over
 * rows x C/32).  2D norms: samples = B*F, rows = h*w; the 3D norms of TemporalResnetBlock
 * (resnet.py:574,588): samples = B, rows = F*h*w.
 */
size_t syn3r_groupnorm_workspace_bytes(int samples, int rows);
int syn3r_groupnorm_f16(const void* x, void* y, int samples, int rows, int C, const void* gamma, const void* beta,
                        float eps, int silu, void* workspace, size_t workspace_bytes, void* stream);
/* The same on the channel concatenation [x1 (C1 channels) | x2 (C2 channels)] read in place (the norm1 of an up block's
 * resnet, resnet.py:272 on torch.cat([hidden, skip], dim=1)); y is [samples, rows, C1 + C2].  C1, C2 % 8 == 0. */
int syn3r_groupnorm_2src_f16(const void* x1, int C1, const void* x2, int C2, void* y, int samples, int rows,
                             const void* gamma, const void* beta, float eps, int silu, void* workspace,
                             size_t workspace_bytes, void* stream);
/*
 * GroupNorm statistics out of the PRODUCER's epilogue (round 6).  Every GroupNorm input of the UNet is the output of a
 * contraction (resnet.py:272,286,574,588: conv1 + temb -> norm2, conv2 + shortcut -> the temporal norm1, the AlphaBlender
 * output -> the next norm1; transformer_temporal.py:235), whose epilogue holds the values in registers: the statistics
 * pass over the activation (one full extra read) is not launched.
 *
 * syn3r_gemm_set_gn_partials(buf, bytes): the NEXT contraction the calling thread launches through syn3r_gemm_f16 /
 * syn3r_gemm_2src_f16 / syn3r_conv2d3x3_f16 / syn3r_tconv3_f16 also writes, for its [M, N] fp16 output as stored,
 *     buf[((m / 32) * 2 + q) * (N / 10) + n / 10]   (float; q = 0: sum of x, q = 1: sum of x^2)
 * over rows [32 rb, 32 rb + 32) and columns [10 u, 10 u + 10) - if M % 32 == 0, N % 80 == 0, bytes >=
 * syn3r_gn_partials_bytes(M, N) and the kernel chosen for the shape has the lean epilogue (the persistent 256-row
 * kernels).  The request is consumed by that one call (thread_local, like the split-K workspace; nothing is shared between
 * host threads).  syn3r_gemm_gn_partials_written() tells whether the calling thread's LAST contraction wrote them; if not,
 * the consumer runs syn3r_groupnorm_f16 as before.  Fixed summation order: bitwise reproducible run to run.
 *
 * syn3r_groupnorm_pre_f16: syn3r_groupnorm_f16 / _2src_f16 (x2 = NULL, C2 = 0: one source) with the statistics folded from
 * such partial sums (part1 for x1, part2 for x2).  rows % 32 == 0, (C1 + C2) / 32 and C1 multiples of 10.  Same workspace.
 */
size_t syn3r_gn_partials_bytes(int M, int N);
int syn3r_gemm_set_gn_partials(void* partials, size_t bytes);
int syn3r_gemm_gn_partials_written(void);
int syn3r_groupnorm_pre_f16(const void* x1, int C1, const void* part1, const void* x2, int C2, const void* part2, void* y,
                            int samples, int rows, const void* gamma, const void* beta, float eps, int silu,
                            void* workspace, size_t workspace_bytes, void* stream);

/*
 * LayerNorm over C of [M, C] fp16.  If addvec != NULL, addvec[m / rows_per_vec, :] is added first
 * (fp16 add: `hidden_states + emb`, transformer_temporal.py:355) and, if xsum != NULL, the sum is
 * also written there (it is the temporal block's residual, attention.py:490).
 */
int syn3r_layernorm_f16(const void* x, void* y, void* xsum, const void* addvec, int rows_per_vec, long long M,
                        int C, const void* gamma, const void* beta, float eps, void* stream);

/* GEGLU gate (activations.py GEGLU.forward): y[M,D] = x[:, :D] * gelu_erf(x[:, D:]) for x [M, 2D]. */
int syn3r_geglu_f16(const void* x, void* y, long long M, int D, void* stream);

/* Row softmax of an fp16 matrix in fp32 arithmetic: y[m,:] = softmax(scale * x[m,:]); x may equal y.
 * The single-head, 512-wide attention of the VAE mid blocks (attention_processor.py:1222-1299 through
 * vae.py:119-129 and unet_3d_blocks.py:1794-1805) is two syn3r_gemm_f16 calls around this kernel. */
int syn3r_softmax_rows_f16(const void* x, void* y, long long M, int N, long long ld, float scale, void* stream);

/* TemporalDecoder.time_conv_out (autoencoder_kl_temporal_decoder.py:78-84,156-160): Conv3d(3,3,(3,1,1)) over
 * frames.  x [B*F*HW, ldx] fp16 channels-last (first 3 columns used), w [3,3,3] = weight[co][ci][dt] and bias [3]: fp32 HOST
 * pointers (30 floats, passed as kernel arguments) -> out [B*F, 3, HW] fp32 (the NCHW frames the decoder returns). */
int syn3r_time_conv_out(const void* x, long long ldx, const float* w, const float* bias, float* out, int B, int F,
                        long long HW, void* stream);

/* ------------------------------------------------------------------------
 * Trainer-loop pieces adjacent to the rasteriser.  The reference runs them inside FSGS'
 * gsTrainer.training()/finetune() (call sites model/diffusionGS.py:139, 1640; submodule not vendored) as
 * torch elementwise chains; the published 3DGS step is Ll1 = |render - gt|.mean() and
 * torch.optim.Adam(eps = 1e-15).
 * ------------------------------------------------------------------------ */

/* loss[0] = weight * mean(|image - target|) over n floats (device scalar; deterministic two-level sum).
 * ws: syn3r_l1_loss_workspace_bytes(n) bytes of device scratch. */
size_t syn3r_l1_loss_workspace_bytes(long long n);
int syn3r_l1_loss(const float* image, const float* target, long long n, float weight, float* loss, void* ws,
                  size_t ws_bytes, void* stream);
/* grad_image[i] = grad_loss[0] * weight / n * sign(image[i] - target[i]); grad_loss is a DEVICE scalar
 * (NULL = 1), so autograd's upstream gradient never visits the host. */
int syn3r_l1_loss_backward(const float* image, const float* target, long long n, float weight,
                           const float* grad_loss, float* grad_image, void* stream);

/* mse[0] = mean((image - target)^2) over n floats (device scalar, same deterministic two-level sum and workspace
 * as syn3r_l1_loss): the MSE behind the PSNR column of the per-scene metric record (the reference
 * tabulates PSNR/SSIM/LPIPS in scripts/summarize_dl3dv.py:11-80 from FSGS' metrics.py, not vendored). */
int syn3r_image_mse(const float* image, const float* target, long long n, float* mse, void* ws, size_t ws_bytes,
                    void* stream);

/* The published 3DGS photometric loss in one pass:
 *   loss3[0] = weight * ((1 - lambda_dssim) * mean|I - G| + lambda_dssim * (1 - SSIM(I, G))), loss3[1] = L1, loss3[2] = SSIM
 * I, G [C,H,W] fp32; SSIM with the 11x11 Gaussian window (sigma 1.5), zero padding 5, C1 = 0.01^2, C2 = 0.03^2, mean
 * over all C*H*W.  ws (syn3r_photo_loss_workspace_bytes) also receives the three derivative maps the backward
 * reads, so the SAME workspace must be passed to syn3r_photo_loss_backward, which writes
 * grad_image = grad_loss[0] * d loss3[0] / d I  (grad_loss: device scalar, NULL = 1). */
size_t syn3r_photo_loss_workspace_bytes(int C, int H, int W);
int syn3r_photo_loss(const float* image, const float* target, int C, int H, int W, float lambda_dssim, float weight,
                     float* loss3, void* ws, size_t ws_bytes, void* stream);
int syn3r_photo_loss_backward(const float* image, const float* target, int C, int H, int W, float lambda_dssim,
                              float weight, const float* grad_loss, const void* ws, float* grad_image, void* stream);
/* Value AND gradient in two launches (a training step wants both): syn3r_photo_loss followed by syn3r_photo_loss_backward on the
 * same arguments, except that loss3 is written by the gradient pass (its first block forms the forward's sums; the gradient does
 * not depend on them) instead of by a single-block launch between the passes.  Same bits in loss3 and grad_image. */
int syn3r_photo_loss_step(const float* image, const float* target, int C, int H, int W, float lambda_dssim, float weight,
                          const float* grad_loss, float* loss3, float* grad_image, void* ws, size_t ws_bytes, void* stream);

/* One torch.optim.Adam update (no weight decay, no amsgrad) of n fp32 parameters in place, in torch's
 * operation order: exp_avg.lerp_(g, 1-beta1); exp_avg_sq = beta2*exp_avg_sq + (1-beta2)*g*g;
 * param -= lr/(1-beta1^step) * exp_avg / (sqrt(exp_avg_sq)/sqrt(1-beta2^step) + eps).  step is 1-based. */
int syn3r_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                    float beta1, float beta2, float eps, int step, void* stream);
/* The same update of up to 8 tensors in ONE launch (the trainer's parameter groups: torch's _multi_tensor_adam; element for
 * element the arithmetic of syn3r_adam_step).  Host arrays of `count` entries: device pointers, element counts, learning rates,
 * eps and 1-based steps per tensor; beta1 / beta2 are shared. */
int syn3r_adam_step_multi(int count, float* const* params, const float* const* grads, float* const* exp_avgs,
                          float* const* exp_avg_sqs, const long long* numels, const float* lrs, float beta1, float beta2,
                          const float* epss, const int* steps, void* stream);

/* The parameter activations of the published 3DGS model that FSGS' trainer applies before every render inside
 * gsTrainer.training() / finetune() (model/diffusionGS.py:139,1640; GaussianModel.get_scaling / get_rotation / get_opacity):
 * scales = exp(log_scales) [N,3], rotations_n = rotations / max(|rotations|_2, 1e-12) [N,4], opacities = sigmoid(logits) [N],
 * in one launch; and their chain rule (torch's formulas) in one launch: d_log_scales = d_scales * scales,
 * d_rotations = (g - qhat (qhat . g)) / max(|q|, 1e-12), d_opacity_logits = d_opacities * s (1 - s).  fp32, caller-owned. */
int syn3r_gaussian_activate(int N, const float* log_scales, const float* rotations, const float* opacity_logits, float* scales,
                            float* rotations_n, float* opacities, void* stream);
int syn3r_gaussian_activate_backward(int N, const float* rotations, const float* scales, const float* rotations_n,
                                     const float* opacities, const float* d_scales, const float* d_rotations_n,
                                     const float* d_opacities, float* d_log_scales, float* d_rotations, float* d_opacity_logits,
                                     void* stream);

/* The per-iteration statistics of the published adaptive density control (3DGS section 5.2; GaussianModel.add_densification_stats
 * as FSGS' loop behind gsTrainer.training() applies it), for the Gaussians the render saw (radii[i] > 0), in one launch and without
 * the host synchronisations of boolean-mask indexing: grad_accum[i] += |viewspace_grad[i, :2]|_2 (sqrt(gx*gx + gy*gy), fp32),
 * denom[i] += 1, max_radii[i] = max(max_radii[i], radii[i]).  viewspace_grad [N,3] fp32; radii [N] i32; the rest [N] fp32. */
int syn3r_densification_stats(int N, const int* radii, const float* viewspace_grad, float* grad_accum, float* denom,
                              float* max_radii, void* stream);

/* out[i] = mean of the three smallest squared Euclidean distances from point i to the OTHER points of the cloud
 * (points [n,3] fp32, n >= 4): the quantity FSGS' GaussianModel.create_from_pcd takes from `distCUDA2` of the
 * simple-knn CUDA extension to initialise the Gaussian scales (reached from reset_gaussians_from_pcd,
 * model/diffusionGS.py:1685-1687; the extension is an un-vendored submodule).  Exact search (Morton
 * order + bounding-box pruning), d2 = (dx*dx + dy*dy) + dz*dz without fused multiply-add, sum (b0 + b1) + b2 ascending.
 * ws: syn3r_knn3_workspace_bytes(n) bytes of 256-byte aligned device scratch. */
size_t syn3r_knn3_workspace_bytes(int n);
int syn3r_knn3_mean_dist2(const float* points, int n, float* out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * LPIPS (VGG16) perceptual loss term of the trainer (the reference raises `opt.use_lpips_loss` around every refine,
 * model/diffusionGS.py:1690,1697; the loss lives in un-vendored FSGS and calls the `lpips` package: the PUBLISHED
 * definition is restated, csrc/lpips.hip).  Activations are channels-last fp16 [H*W, C].
 * ------------------------------------------------------------------------ */
/* 3x3 convolution, stride 1, padding 1, + bias, with the activation options VGG needs: relu != 0: out = max(out, 0);
 * relu_mask (nullable) [NB*Hi*Wi, Cout]: out zeroed where mask <= 0 (the ReLU backward of the layer below, fused into the
 * backward-data convolution, which is this same kernel on the transposed, flipped weights).  Cin % 64 == 0, Cout % 8 == 0. */
int syn3r_conv2d3x3_act_f16(const void* X, const void* W, void* out, const void* bias, int relu, const void* relu_mask,
                            int NB, int Hi, int Wi, int Cin, int Cout, void* stream);
/* image [3,H,W] fp32 in [0,1] -> LPIPS input scaling ((2x-1) - shift) / scale -> [H*W, 64] fp16 (channels 3..63 zero) */
int syn3r_lpips_image_f16(const float* img, int H, int W, void* out, void* stream);
/* gradient wrt that [H*W, 64] tensor (carrying loss_scale) -> d loss / d image [3,H,W] fp32 */
int syn3r_lpips_image_bwd(const void* grad64, int H, int W, float loss_scale, float* d_img, void* stream);
/* nn.MaxPool2d(2, 2) on [H,W,C] fp16 -> [H/2,W/2,C], and its backward (gradient to the first maximum of each window) */
int syn3r_maxpool2_f16(const void* x, int H, int W, int C, void* y, void* stream);
int syn3r_maxpool2_bwd_f16(const void* x, const void* gy, int H, int W, int C, void* gx, void* stream);
/* One LPIPS layer: value[0] (=, or += with accumulate) mean_p sum_c w_c (a_c/(|a|+1e-10) - b_c/(|b|+1e-10))^2 for feature
 * maps a, b [P, C] fp16, C in {64,128,256,512}, w [C] fp32 device; and its gradient wrt a times gscale into grad_a [P, C]. */
size_t syn3r_lpips_layer_workspace_bytes(long long P, int C);
int syn3r_lpips_layer_f16(const void* a, const void* b, const float* w, long long P, int C, int accumulate, float* value,
                          void* ws, size_t ws_bytes, void* stream);
int syn3r_lpips_layer_bwd_f16(const void* a, const void* b, const float* w, long long P, int C, float gscale, int accumulate,
                              void* grad_a, void* stream);

/* Statistical outlier removal of a point cloud: what model/diffusionGS.py:321 asks of open3d
 * (`down_pcd.remove_statistical_outlier(nb_neighbors=20, std_ratio=3.0)`; open3d 0.17.0 is not in the reference tree, the
 * published algorithm is restated in csrc/knn.hip).  points [n,3] float64 (open3d holds doubles).  Outputs, all device
 * memory: avg_dist [n] float64 = mean distance to the nb_neighbors nearest points (the point itself included, as the k-d
 * tree query returns it); keep [n] bytes = 0 < avg < mean + std_ratio * std; stats [4] = {mean, std (n-1), threshold,
 * valid count}.  nb_neighbors must be 20 (the reference's call).  ws: syn3r_pcd_outlier_workspace_bytes(n), 256-byte aligned. */
size_t syn3r_pcd_outlier_workspace_bytes(int n);
int syn3r_pcd_statistical_outlier(const double* points, int n, int nb_neighbors, double std_ratio, double* avg_dist,
                                  unsigned char* keep, double* stats, void* ws, size_t ws_bytes, void* stream);

/* Forward / backward optical-flow cycle-consistency mask: the test behind
 * `gsTrainer.generate_corresp_mask(gs_renderings, svd_outputs, dist_thresh=3, desc_only=False)` (model/diffusionGS.py:377;
 * the FSGS wrapper and GMFlow are absent, see csrc/warp.hip).  flow_fw / flow_bw [n,2,H,W] fp32 (A->B and B->A, channel 0 =
 * dx); mask [n,H,W] fp32 in {0,1}: |fw(x) + bw(x + fw(x))| < thresh and x + fw(x) inside the image; dist (nullable)
 * [n,H,W] = that cycle error (+inf outside). */
int syn3r_flow_cycle_mask(const float* flow_fw, const float* flow_bw, int n, int H, int W, float thresh, float* mask, float* dist,
                          void* stream);

/*
 * The whole UNet forward as ONE call: `self.unet(latent_model_input, t, encoder_hidden_states=image_embeddings,
 * added_time_ids=added_time_ids, return_dict=False)[0]` (model/SVD_2pass_prob_uncertain_post.py:763,786;
 * UNetSpatioTemporalConditionModel.forward, diffusers/models/unets/unet_spatio_temporal_condition.py:356-489) for hosts that
 * are not Python.  The launch sequence is the one syn3r_amd/unet/model.py issues (same operators, same order, same results).
 *
 * syn3r_unet_create: reads `<weights_dir>/config.json` and `<weights_dir>/diffusion_pytorch_model[.<variant>].safetensors`
 *   (the `unet/` directory of a diffusers checkpoint: what `from_pretrained(<local dir>, torch_dtype=float16, variant="fp16")`
 *   reads at model/diffusionGS.py:1089; variant may be NULL; F16 / F32 / BF16 tensors, kept in fp16), repacks the weights for the
 *   kernels and uploads them to the CURRENT device.  One handle per device; a handle is not re-entrant (one forward at a time).
 * syn3r_unet_workspace_bytes: exact scratch need of one forward of that shape (0 on a bad shape).
 * syn3r_unet_forward, all pointers device memory on the handle's device:
 *   sample [B, F, in_channels, h, w] fp16; timestep (the scheduler's continuous timestep); encoder_hidden_states
 *   [ehs_rows, cross_attention_dim] fp16 with ehs_rows = B, or 1 for one context shared by the batch; added_time_ids
 *   [B, 3] fp32 (fps, motion bucket, noise augmentation); out [B, F, out_channels, h, w] fp16.  ctx_group: 0 = the batch is
 *   one call of the reference; G > 0 = B / G independent calls of batch G stacked (the two passes of a denoising step: the
 *   reference's batch-interleaved temporal context is applied per group).  h, w multiples of 8, F <= 32.
 *   workspace: syn3r_unet_workspace_bytes(...) bytes, 256-byte aligned.  Every launch goes to `stream`; nothing synchronises
 *   except the first forward of a new (F, B), which allocates the per-block frame-position embeddings (hipMalloc).
 */
typedef struct syn3r_unet syn3r_unet;
int syn3r_unet_create(const char* weights_dir, const char* variant, syn3r_unet** out);
int syn3r_unet_destroy(syn3r_unet* unet);
size_t syn3r_unet_workspace_bytes(syn3r_unet* unet, int B, int F, int h, int w, int ehs_rows);
int syn3r_unet_forward(syn3r_unet* unet, const void* sample, double timestep, const void* encoder_hidden_states, int ehs_rows,
                       const float* added_time_ids, void* out, int B, int F, int h, int w, int ctx_group, void* workspace,
                       size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SYN3R_HIP_H */
