/* keymorph_hip.h -- C ABI of libkeymorph_hip.so (MI355X / gfx950 only).
 *
 * The reference (alanqrwang/keymorph @ 2.0.1) is pure Python on ATen: it has no
 * FFI of its own, so the drop-in boundary is the Python call surface of
 * keymorph/*  and THIS header is the native layer directly
 * underneath it -- one entry point per ATen call site of the hot path, each citing
 * the reference line(s) it replaces.  INTEGRATION.md shows the ctypes binding a
 * reference maintainer would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (hipMalloc / torch.cuda storage), fp32
 *     unless noted, densely packed in the stated layout;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - no allocation, no synchronisation, no ownership transfer inside; scratch
 *     is caller-provided (`ws`) and its size comes from the matching *_ws_bytes();
 *   - return value: 0 on success, otherwise the hipError_t of the failed launch
 *     or a negative KMH_E* code for bad arguments.
 */
#ifndef KEYMORPH_HIP_H
#define KEYMORPH_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define KMH_EINVAL (-22)
#define KMH_ABI_VERSION 1
int kmh_abi_version(void);

/* ---- a11: align_img = F.grid_sample(bilinear|nearest, border, align_corners=False)
 *      keymorph/utils.py:14-21.  x (N,C,D,H,W); grid (N,Do,Ho,Wo,3) xyz; out (N,C,Do,Ho,Wo).
 *      mode: 0 bilinear, 1 nearest. */
int kmh_grid_sample3d_fwd(const float* x, const float* grid, float* out, int N, int C, int D, int H,
                          int W, int Do, int Ho, int Wo, int mode, void* stream);
/* d(out)/d(grid) contracted with gout (N,C,Do,Ho,Wo) -> dgrid (N,Do,Ho,Wo,3) (bilinear). */
int kmh_grid_sample3d_bwd_grid(const float* x, const float* grid, const float* gout, float* dgrid,
                               int N, int C, int D, int H, int W, int Do, int Ho, int Wo, void* stream);
/* d(out)/d(x) contracted with gout -> dx (N,C,D,H,W), dx must be zero-filled by the caller. */
int kmh_grid_sample3d_bwd_input(const float* grid, const float* gout, float* dx, int N, int C, int D,
                                int H, int W, int Do, int Ho, int Wo, void* stream);

/* ---- a12: MSELoss, keymorph/loss_ops.py:9-13.  out[0] = mean((a-b)^2); ws >= kmh_reduce_ws_bytes(). */
size_t kmh_reduce_ws_bytes(void);
int kmh_mse_fwd(const float* a, const float* b, long long n, float* out, void* ws, void* stream);
/* da = gscale[0] * 2 (a-b)/n  (db = -da) */
int kmh_mse_bwd(const float* a, const float* b, const float* gscale, long long n, float* da, void* stream);
/* fused warp + MSE against `fixed` (same shape as out): writes out and out_loss[0]. */
int kmh_warp_mse_fwd(const float* x, const float* grid, const float* fixed, float* out, float* out_loss,
                     int N, int C, int D, int H, int W, int Do, int Ho, int Wo, void* ws, void* stream);
/* the same pass also writes dgrid = d(out_loss)/d(grid) (the MSE cotangent 2 (out - fixed) / count is known inside the
 * warp): align_img + MSELoss + autograd of both in ONE launch, 36 B per voxel instead of 68 in three; out may be NULL.
 * -22 when the shape needs the generic sampler (then use the separate entry points). */
int kmh_warp_mse_fwd_grad(const float* x, const float* grid, const float* fixed, float* out, float* out_loss,
                          float* dgrid, int N, int C, int D, int H, int W, int Do, int Ho, int Wo, void* ws,
                          void* stream);
/* a (n floats, n % 4 == 0) *= g[0], skipped on the device when g[0] == 1 (loss.backward() with the default cotangent) */
int kmh_scale_unless_one(float* a, long long n, const float* g, void* stream);

/* ---- a13: DiceLoss sums, keymorph/loss_ops.py:43-57.  pred/target (R, V) rows = n*c;
 *      sums (R,3) = {sum t*p, sum p*p, sum t*t}. */
int kmh_dice_sums(const float* pred, const float* target, int R, long long V, float* sums, void* ws,
                  void* stream);
/* out[r,i] = ca[r]*t[r,i] + cb[r]*p[r,i] (Dice backward wrt pred) */
int kmh_rows_axpby(const float* t, const float* p, const float* ca, const float* cb, int R, long long V,
                   float* out, void* stream);
/* ---- a11 + a13 fused (scripts/train.py:146-164 with the Dice loss): align_img (keymorph/utils.py:14-21) followed by
 *      the Dice sums (keymorph/loss_ops.py:28-52) WITHOUT storing the warped segmentation.
 *      sums[(n*C + c)*3 + {0,1,2}] = {sum t p, sum p^2, sum t^2}, p = grid_sample(x, grid)[n,c], t = fixed[n,c].
 *      -22 when the lane-contiguous sampler does not apply (W < 2, plane >= 2^31 voxels, C > 128). */
/* 1 when the fused warp + Dice entry points below serve these sizes (they return -22 otherwise): W >= 2, < 2^30 voxels per
 * channel plane, C <= 128, N * C <= 65536, the lane-contiguous sampler not switched off (KMH_SAMPLER_OLD).  The host asks
 * BEFORE building an autograd node and otherwise composes keymorph/utils.py:14-21 with keymorph/loss_ops.py:16-63 unfused. */
int kmh_warp_dice_ok(int N, int C, int D, int H, int W);
int kmh_warp_dice_sums(const float* x, const float* grid, const float* fixed, float* sums, int N, int C, int D, int H,
                       int W, int Do, int Ho, int Wo, const unsigned char* lab_x, const unsigned char* lab_fixed,
                       const int* gate, void* ws, void* stream);
/* its backward (autograd of keymorph/loss_ops.py:16-63 through keymorph/utils.py:14-21, one pass):
 *      dgrid[n,v,:] = sum_c (ca[n*C+c] t + cb[n*C+c] p) * d p / d grid, with ca = -2 g / den, cb = 2 g num / den^2. */
int kmh_warp_dice_bwd_grid(const float* x, const float* grid, const float* fixed, const float* ca, const float* cb,
                           float* dgrid, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                           const unsigned char* lab_x, const unsigned char* lab_fixed, const int* gate, void* stream);
/*      lab_x (N, D*H*W) / lab_fixed (N, Do*Ho*Wo) / gate: all NULL, or the label maps and the device flag written by
 *      kmh_onehot_to_labels: one-hot segmentations (what keymorph/utils.py:200-240 + nearest-sampled augmentation produce,
 *      scripts/train.py:54-98) are then read as ONE byte per voxel (8 corner labels gathered once instead of 8 corner
 *      values per channel), bit-identical results; gate[0] == 0 (a soft segmentation) -> the float tensors are read. */
int kmh_onehot_to_labels(const float* x, int N, int C, long long V, unsigned char* lab, int* ok, void* stream);
/* hard Dice: onehot(argmax_c pred) over (N,C,V) -> out (N,C,V); first max wins like torch.argmax */
int kmh_argmax_onehot(const float* pred, int N, int C, long long V, float* out, void* stream);
/* keymorph/loss_ops.py:161-247 (_jacobian_determinant, jdstd, jdlessthan0): central differences with zero padding
 * + identity, cropped by 2 voxels; component c of voxel v at disp[c*cstride + v*vstride].  jd (D-4,H-4,W-4) or
 * NULL; stats[4] (double) = {mean, population std, #(det <= 0), #voxels}; ws = kmh_reduce_ws_bytes(). */
int kmh_jacobian_det(const float* disp, long long cstride, long long vstride, int D, int H, int W, float* jd,
                     double* stats, void* ws, void* stream);

/* ---- label-map encodings in front of the Dice branch: keymorph/utils.py:200-240 (one_hot,
 * one_hot_subsampled_pair; callers scripts/train.py:54-61) ---- */
/* flags: nflags + 1 ints zeroed by the caller; flags[l] = 1 iff label l occurs among the n int64 values of seg,
 * flags[nflags] = 1 iff a label lies outside [0, nflags) (replaces np.unique on the host). */
int kmh_label_presence(const long long* seg, long long n, int nflags, int* flags, void* stream);
/* out[n,c,v] = (seg[n,v] == labels[c]), C <= 256; float32 out (out_i64 = 0) or int64 out like F.one_hot (1). */
int kmh_one_hot_select(const long long* seg, int N, long long V, const long long* labels, int C, void* out,
                       int out_i64, void* stream);

/* ---- a9: AffineTransform.get_flow_field, keymorph/transformations.py:37-79 and
 *      uniform_norm_grid keymorph/utils.py:387-398.  mat (N,3,4) = inverse_transform_matrix[:, :3, :]
 *      acting on ij coords; out (N,D,H,W,3) already flipped to xyz. */
int kmh_affine_grid_fwd(const float* mat, float* out, int N, int D, int H, int W, void* stream);
int kmh_affine_grid_bwd(const float* dgrid, float* dmat, int N, int D, int H, int W, void* ws, void* stream);

/* ---- a8: TPS.get_flow_field / transform_points, keymorph/keypoint_aligners.py:365-433.
 *      theta (N,T+4,3) rows [w_0..w_{T-1}, a_0, a_z, a_y, a_x]; ctrl (N,T,3) ij; out (N,D,H,W,3) xyz. */
int kmh_tps_grid_fwd(const float* theta, const float* ctrl, float* out, int N, int T, int D, int H, int W,
                     void* stream);
size_t kmh_tps_grid_bwd_ws_bytes(int N, int T, int D, int H, int W);
int kmh_tps_grid_bwd(const float* dgrid, const float* theta, const float* ctrl, float* dtheta,
                     float* dctrl, int N, int T, int D, int H, int W, void* ws, void* stream);
/* same spline evaluated at explicit points (N,P,3) ij -> out (N,P,3) ij (a10: points_a). */
int kmh_tps_points_fwd(const float* theta, const float* ctrl, const float* pts, float* out, int N, int T,
                       int P, void* stream);
size_t kmh_tps_points_bwd_ws_bytes(int N, int T, int P);
int kmh_tps_points_bwd(const float* dout, const float* theta, const float* ctrl, const float* pts,
                       float* dtheta, float* dctrl, float* dpts, int N, int T, int P, void* ws,
                       void* stream);

/* ---- a7: TPS.fit, keymorph/keypoint_aligners.py:276-363.  Assembles A = [[U+lambda I, P],[P^T,0]]
 *      in fp32 exactly as the reference, factorises in fp64 (partial-pivot LU, one workgroup per
 *      sample) and solves 3 right-hand sides.  ctrl,tgt (N,T,3); lmbda (N); w (N,T) or NULL;
 *      theta (N,T+4,3).  ws keeps the LU factors + pivots for the backward. */
size_t kmh_tps_fit_ws_bytes(int N, int T);
int kmh_tps_fit_fwd(const float* ctrl, const float* tgt, const float* lmbda, const float* w, float* theta,
                    int N, int T, void* ws, void* stream);
/* Large systems are factorised by a cluster of workgroups per system that hand panels over through global counters; a
 * cluster that could not get all its workgroups resident gives up (bounded waits) and the SAME call redoes that system on
 * the one-workgroup kernel (device-side retry pass, no host round trip).  Test hook: the next `count` clustered calls mark
 * every system as "gave up" so that the retry pass does all the work; returns the previous count. */
int kmh_tps_fit_force_retry(int count);
/* backward: given dtheta -> dctrl, dtgt (uses the factors in ws written by the forward) and, when the fit was
 * weighted and dw != NULL, dw (N,T) = d/dw of the lmbda / (w + 1e-6) diagonal. */
int kmh_tps_fit_bwd(const float* dtheta, const float* theta, const float* ctrl, const float* lmbda, const float* w,
                    float* dctrl, float* dtgt, float* dw, int N, int T, void* ws, void* stream);

/* ---- a5/a6: closed-form affine (keymorph/keypoint_aligners.py:76-114) and rigid/Kabsch
 *      (:151-213) fits.  x,y (N,K,3); w (N,K) or NULL; M (N,3,4). */
int kmh_affine_fit_fwd(const float* x, const float* y, const float* w, float* M, int N, int K, void* stream);
/* backwards: dw (N,K) or NULL = gradient w.r.t. the keypoint weights (needs w != NULL). */
int kmh_affine_fit_bwd(const float* dM, const float* x, const float* y, const float* w, const float* M,
                       float* dx, float* dy, float* dw, int N, int K, void* stream);
int kmh_rigid_fit_fwd(const float* x, const float* y, const float* w, float* M, int N, int K, void* stream);
int kmh_rigid_fit_bwd(const float* dM, const float* x, const float* y, const float* w, float* dx, float* dy,
                      float* dw, int N, int K, void* stream);
/* 4x4 homogeneous inverse of [M;0 0 0 1] (keymorph/transformations.py:23-35) and its backward. */
int kmh_affine_inverse_fwd(const float* M, float* Minv, int N, void* stream);
int kmh_affine_inverse_bwd(const float* dMinv, const float* Minv, float* dM, int N, void* stream);
/* points (N,P,3) -> M[:, :3, :] @ [p;1] (keymorph/transformations.py:81-114) */
int kmh_affine_points_fwd(const float* M, const float* pts, float* out, int N, int P, void* stream);
/* keymorph/augmentation.py:85-158 AffineDeformation3d.build_affine_matrix: scale (B,3), offset (B,3), theta (B,3),
 * shear (B,6) -> out (B,4,4) = Mz Ms Mt (R3 R2 R1) */
int kmh_affine_build_matrix(const float* scale, const float* offset, const float* theta, const float* shear,
                            float* out, int B, void* stream);
int kmh_affine_points_bwd(const float* dout, const float* M, const float* pts, float* dM, float* dpts,
                          int N, int P, void* stream);

/* ---- a4: CenterOfMass3d, keymorph/layers.py:78-134.  feat (N,K,D,H,W) -> pts (N,K,3) (z,y,x)
 *      in [-1,1]; sums (N,K,4) = {m, mz, my, mx} kept for the backward. */
int kmh_com3d_fwd(const float* feat, float* pts, float* sums, int N, int K, int D, int H, int W, void* ws,
                  void* stream);
int kmh_com3d_bwd(const float* dpts, const float* feat, const float* sums, float* dfeat, int N, int K,
                  int D, int H, int W, void* stream);


/* ==== a2/a3: backbone operators.  Activations are NDHWC (N, D, H, W, C) fp32 internally. ==== */

/* conv3d k=3 p=1 s=1 (keymorph/unet3d/buildingblocks.py:46-58, keymorph/layers.py:173-175) on the
 * fp32 matrix cores.  pack: torch (Cout,Cin,3,3,3) -> [27][Cin][Cout] (transposed=0, forward) or the
 * tap-mirrored [27][Cout][Cin] (transposed=1) that turns the SAME kernel into the data gradient.
 * fwd: y = act_out( conv( act_in(x*scale[n,c]+shift[n,c]) * [mask > 0] ) + bias ); scale/shift/mask/bias may be
 * NULL.  `mask` (same shape as x) fuses the ReLU backward into the data-gradient pass (x = dy, mask = y). */
int kmh_conv3d_pack_weight(const float* w, float* packed, int Cout, int Cin, int transposed, void* stream);
int kmh_conv3d_fwd(const float* x, const float* scale, const float* shift, const float* mask,
                   const float* packed_w, const float* bias, float* y, int N, int D, int H, int W, int Cin,
                   int Cout, int relu_in, int relu_out, void* stream);
/* Split-operand variant of the same convolution (csrc/conv_bf.hip): fp32 operands are split into 16-bit pieces and
 * multiplied on the 16-bit matrix cores with fp32 accumulation.  terms = 2 ("f16x3", the host default): operands
 * range-scaled by powers of two (ascale / wscale = {S, 1/S} device pairs, see kmh_absmax_scale) and split into fp16
 * hi + lo, 3 MFMAs per product block; terms = 3 ("bf16x6"): bf16 hi + mid + lo, 6 MFMAs, no scales.  Both are
 * fp32-class (5e-7 vs fp64).  Same semantics / arguments as kmh_conv3d_pack_weight + kmh_conv3d_fwd, with its own
 * packed layout. */
size_t kmh_conv3d_pack_bf_bytes(int Cout, int Cin, int transposed, int terms);
int kmh_conv3d_pack_weight_bf(const float* w, void* packed, int Cout, int Cin, int transposed, int terms,
                              const float* wscale, void* stream);
/* stats_out (N,Cout,2) doubles | NULL: per-channel (sum y, sum y^2) of the OUTPUT from the epilogue -- what
 * kmh_channel_stats(y) returns, i.e. the next GroupNorm's statistics without another pass over y; stats_ws holds the
 * per-brick partials (kmh_conv3d_fwd_bf_stats_ws_bytes). */
size_t kmh_conv3d_fwd_bf_stats_ws_bytes(int N, int D, int H, int W, int Cout, int rows_per_wave);
int kmh_conv3d_fwd_bf(const float* x, const float* scale, const float* shift, const float* mask, const void* packed,
                      const float* bias, float* y, int N, int D, int H, int W, int Cin, int Cout, int relu_in,
                      int relu_out, int terms, int rows_per_wave, const float* ascale, const float* wscale,
                      void* stats_ws, double* stats_out, int in_blocked, const float* addend, void* stream);
/* Kernel selection of kmh_conv3d_fwd_bf (the reference has one conv3d: keymorph/unet3d/buildingblocks.py:46-58; this
 * only chooses between two implementations with bit-identical results).  mode 0: conv3_fwd_bf_kernel always; 1: the
 * LDS-DMA kernel conv3_fwd_g_kernel for launches of >= 512 bricks (default; KEYMORPH_FWD_G sets the initial value);
 * 2: conv3_fwd_g_kernel whenever its preconditions hold (parity tests on small / ragged volumes).  Returns the
 * previous mode, -22 for a bad argument. */
int kmh_conv3d_fwd_bf_set_dispatch(int mode);
/* use_amp (keymorph/model.py:176-191 autocasts the keypoint extractor to fp16) is PER CALL: every launching entry point below
 * that takes `terms` -- kmh_conv3d_fwd_bf, kmh_conv3d_fwd_bf_pool, kmh_conv3d_wgrad_bf, kmh_conv3d_up2_fwd, kmh_conv3d_up2_dgrad,
 * kmh_up2_wgrad_gemm, kmh_up2_wgrad_fold, kmh_headcom_fwd_bf, kmh_headcom_bwd_bf -- accepts terms == 1 = "the fp16 kernels of
 * terms == 2 with only the hi x hi product": fp16 inputs (11 significant bits), fp32 accumulation, one MFMA per product block
 * instead of three.  Packing, scales, layouts, workspaces and the *_ok / *_bytes queries are those of terms == 2 (pass 2
 * there).  There is no process-wide switch: the caller that ran a forward with terms == 1 passes 1 to its backward calls. */
/* in_blocked of kmh_conv3d_fwd_bf: 0 = x is (N,D,H,W,Cin); 1 = channel-blocked (N,Cin/8,D,H,W,8) fp32; 2 = PRE-SPLIT
 * channel-blocked: (N, Cin/8, D*H*W + 1) records of 32 bytes, the 8 fp16 "hi" then the 8 fp16 "lo" terms of fmaf(value, S, 0)
 * with S = ascale[0], record D*H*W of every plane zero -- what kmh_maxpool3d_bwd_split writes; the kernel then copies
 * fragments (LDS-DMA) instead of converting them, with bit-identical results (conv3_fwd_s_kernel<1,true,true>).  Mode 2 serves
 * gradient operands (no scale / shift / mask / relu_in / addend) of the z-paired tile (Cout <= 16): kmh_conv3d_fwd_bf_split_ok
 * returns 1 for the shapes it takes, kmh_conv3d_fwd_bf returns -22 for the others.  Replaces the data gradient of
 * keymorph/unet3d/buildingblocks.py:46-58 (autograd of conv3d) behind a pooling layer (:321-380). */
int kmh_conv3d_fwd_bf_split_ok(int N, int D, int H, int W, int Cin, int Cout, int terms);
/* the kernel kmh_conv3d_fwd_bf would launch for this call now: 0 conv3_fwd_bf_kernel, 1 / 2 conv3_fwd_g_kernel with a
 * 32- / 64-wide output-channel tile, 3 its z-paired variant (Cout <= 16) */
int kmh_conv3d_fwd_bf_variant(int N, int D, int H, int W, int Cin, int Cout, int terms, int has_mask, int has_addend);
/* GroupNorm -> Conv3d -> ReLU -> MaxPool3d(2) in one launch (keymorph/unet3d/buildingblocks.py:46-78 followed by the next
 * Encoder's `self.pooling`, :321-380), for a block whose output feeds only that pooling: yp (N, D/2, H/2, W/2, Cout) the
 * pooled output, arg (same shape, bytes) the winners' window indices in kmh_maxpool3d_fwd's format (ATen's first-max
 * rule), stats_out (N, Cout, 2) the pooled tensor's (sum, sum^2); the full-resolution output is not written.
 * kmh_conv3d_fwd_bf_pool_ok: 1 when the shape is served (16 < Cout <= 32, Cin % 8 == 0, terms == 2, LDS-DMA kernel). */
int kmh_conv3d_fwd_bf_pool_ok(int N, int D, int H, int W, int Cin, int Cout, int terms);
int kmh_conv3d_fwd_bf_pool(const float* x, const float* scale, const float* shift, const void* packed, float* yp,
                           unsigned char* arg, int N, int D, int H, int W, int Cin, int Cout, int relu_in, int terms,
                           const float* ascale, const float* wscale, void* stats_ws, double* stats_out, int in_blocked,
                           void* stream);
/* addend (like y) | NULL: added to the result before the activation and the statistics (Cout > 16).
 * Decoder's first convolution without the upsampled tensor: the 27 taps over a nearest-x2 upsampled channel fall on
 * 2 x 2 x 2 low-resolution voxels per output parity, so kmh_conv3d_up2_fwd computes the upsampled channels'
 * contribution from the LOW-resolution tensor with 8 (pre-summed) taps instead of 27, and kmh_conv3d_fwd_bf over the
 * skip channels adds it (keymorph/unet3d/buildingblocks.py:471-475 nearest interpolate + cat, then :46-78). */
size_t kmh_conv3d_up2_pack_bytes(int Cout, int Cl, int terms);
int kmh_conv3d_up2_pack_weight(const float* w, void* packed, int Cout, int Ctot, int cofs, int Cl, int terms,
                               const float* wscale, void* stream);
/* data gradient of the same operator: ds (N,Dl,Hl,Wl,Cl) = per low voxel, the sum over its 8 children of the gradient
 * with respect to the upsampled tensor (what interpolate's backward would sum), from dz (N,2Dl,2Hl,2Wl,Cout): 64
 * pre-summed taps at low resolution instead of 8 x 27 at high resolution */
size_t kmh_conv3d_up2_dgrad_pack_bytes(int Cout, int Cl, int terms);
int kmh_conv3d_up2_dgrad_pack_weight(const float* w, void* packed, int Cout, int Ctot, int cofs, int Cl, int terms,
                                     const float* wscale, void* stream);
/* stats_out (N,Cl,2) doubles | NULL: per-channel (sum ds, sum ds^2) from the epilogue (what GroupNorm's backward of the
 * decoder block needs from this gradient: no separate pass over ds); stats_ws: ..._stats_ws_bytes (may be NULL with it) */
size_t kmh_conv3d_up2_dgrad_stats_ws_bytes(int N, int Dl, int Hl, int Wl, int Cl);
int kmh_conv3d_up2_dgrad(const float* dz, const void* packed, float* ds, int N, int Dl, int Hl, int Wl, int Cl, int Cout,
                         int terms, const float* dscale, const float* wscale, void* stats_ws, double* stats_out,
                         int in_blocked /* dz channel-blocked (N, Cout/8, 2Dl, 2Hl, 2Wl, 8): whole lines per request */,
                         void* stream);
/* weight gradient of the same operator (csrc/norm.hip): G (N, Dl*Hl*Wl, 27, Cout) = 2x2x2 box sums of dz such that
 * dW[tap][ci][co] = sum_m x_low[m][ci] G[m][tap][co] -- one plain matrix product over the low-resolution voxels */
int kmh_up2_boxsum(const float* dz, float* G, int N, int Dl, int Hl, int Wl, int Cout, int in_blocked /* as above */,
                   void* stream);
/* C (N, Cl, J) = A^T B per sample over the V low-resolution voxels: A (N, V, Cl) the normalised low tensor, B (N, V, J)
 * the box sums with J = 27 Cout (split-operand MFMA; ascale / bscale = {S, 1/S} of A and B for terms == 2).
 * a_scale / a_shift (N, Cl) | both NULL: A is the RAW low tensor and GroupNorm's affine a_scale[n][c] A + a_shift[n][c]
 * (keymorph/unet3d/buildingblocks.py:46-78, "g" of "gcr") is applied while it is staged; ascale is then the range scale
 * of the normalised values */
size_t kmh_up2_wgrad_gemm_ws_bytes(int N, int V, int Cl, int J);
int kmh_up2_wgrad_gemm(const float* A, const float* B, float* C, int N, int V, int Cl, int J, int terms,
                       const float* ascale, const float* bscale, const float* a_scale, const float* a_shift, void* ws,
                       void* stream);
/* Round 5: the two calls above in ONE product, the box sums formed on the fly from dz (never stored): C (N, Cl, 27 Cout) for
 * xl (N, Dl, Hl, Wl, Cl) the RAW low tensor (a_scale / a_shift (N, Cl): GroupNorm's affine, or both NULL), dz (N, 2Dl, 2Hl,
 * 2Wl, Cout) or channel-blocked (dz_blocked != 0), ascale / dscale = {S, 1/S} range scales of the normalised low tensor and
 * of dz (the box sums are scaled by S / 8).  fp16 split only (terms == 2), Cl % 4 == 0, Cout % 8 == 0: kmh_up2_wgrad_fold_ok.
 * Replaces autograd's weight gradient of interpolate(nearest x2) + cat + conv3d for the upsampled channels
 * (keymorph/unet3d/buildingblocks.py:471-475, :46-78). */
int kmh_up2_wgrad_fold_ok(int Cl, int Cout, int terms);
size_t kmh_up2_wgrad_fold_ws_bytes(int N, int Dl, int Hl, int Wl, int Cl, int Cout);
int kmh_up2_wgrad_fold(const float* xl, const float* dz, float* C, int N, int Dl, int Hl, int Wl, int Cl, int Cout, int terms,
                       const float* ascale, const float* dscale, const float* a_scale, const float* a_shift, int dz_blocked,
                       void* ws, void* stream);
int kmh_conv3d_up2_fwd(const float* xl, const float* scale, const float* shift, int Ctot, int cofs, const void* packed,
                       float* y, int N, int Dl, int Hl, int Wl, int Cl, int Cout, int terms, const float* ascale,
                       const float* wscale, void* stream);
/* in_blocked != 0 (here) / dz_blocked != 0 (weight gradient) / out_blocked != 0 (kmh_gn_bwd_apply): that tensor is
 * stored channel-blocked, (N, C/8, D, H, W, 8) instead of (N, D, H, W, C): the 8 channels of one chunk of one voxel
 * are a 32-byte record and a chunk's voxels are contiguous, so the conv loader uses whole cache lines.  Internal
 * layout of gradients handed from one SingleConv to the one before it; C % 8 == 0, no ReLU mask operand, results
 * bit-identical to the (N, D, H, W, C) path. */
/* split-bf16 weight gradient (same semantics as kmh_conv3d_wgrad; terms = 2 | 3) */
size_t kmh_conv3d_wgrad_bf_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int terms);
int kmh_conv3d_wgrad_bf_blocked_ok(int N, int D, int H, int W, int Cin, int Cout, int terms); /* 1: accepts dz_blocked */
int kmh_conv3d_wgrad_bf(const float* x, const float* scale, const float* shift, const float* dz,
                        const float* dzmask, float* dw, int N, int D, int H, int W, int Cin, int Cout, int relu_in,
                        int accumulate, int terms, int append_ones, const float* xscale, const float* dscale, int dz_blocked, const float* w_fold, double* bhat,
                        void* ws, void* stream);
/* First U-Net convolution, forward (csrc/firstlayer.hip): x (N,D,H,W) raw 1-channel input, scale / shift (N)
 * GroupNorm coefficients of that channel (NULL: identity), w (Cout,1,3,3,3), Cout <= 16 ->
 * y (N,D,H,W,Cout) = relu(conv3(x * scale + shift)) in exact fp32 (keymorph/unet3d/buildingblocks.py:46-78 for
 * encoders[0].SingleConv1); stats_out (N,Cout,2) doubles | NULL = (sum y, sum y^2) as in kmh_conv3d_fwd_bf. */
size_t kmh_conv3d_first_layer_fwd_ws_bytes(int N, int D, int H, int W, int Cout);
int kmh_conv3d_first_layer_fwd(const float* x, const float* scale, const float* shift, const float* w, float* y, int N,
                               int D, int H, int W, int Cout, void* ws, double* stats_out, void* stream);
/* first U-Net layer (Cin = 1), backward of keymorph/unet3d/buildingblocks.py:46-78 for encoders[0] without the
 * 1-channel data gradient: rs (N,Cout,2,27) (doubles; rs_f64 = 1 in the fold) = correlations of dz with the RAW input (R) and with the indicator of the
 * volume (S), by a dedicated exact-fp32 kernel (Cout <= 16); then ... */
size_t kmh_conv3d_first_layer_wgrad_ws_bytes(int N, int D, int H, int W, int Cout);
/* c123 (N,Cout,3) | NULL (needs dzmask): the gradient entering the correlations is [dzmask > 0] (c1 dz + c2 dzmask + c3)
 * -- the NEXT layer's GroupNorm backward (what kmh_gn_bwd_apply would have written) applied while dz is staged */
int kmh_conv3d_first_layer_wgrad(const float* x, const float* dz, const float* dzmask, const float* c123, double* rs, int N,
                                 int D, int H, int W, int Cout, void* ws, void* stream);
/* ... fold the (x, 1) correlations of one sample into dw and GroupNorm's (A, B) sums */
int kmh_conv3d_first_layer_fold(const void* rs, int rs_f64, const float* w, const float* scale_n, const float* shift_n, int Cout,
                                float* dw, double* ab_n, int accumulate, void* stream);
/* dw (Cout,Cin,3,3,3) (+)= sum_v act_in(x*scale+shift)[v+tap] dz[v]*[dzmask[v] > 0]  (dzmask may be NULL) */
size_t kmh_conv3d_wgrad_ws_bytes(int N, int D, int H, int W, int Cin, int Cout);
int kmh_conv3d_wgrad(const float* x, const float* scale, const float* shift, const float* dz,
                     const float* dzmask, float* dw, int N, int D, int H, int W, int Cin, int Cout, int relu_in,
                     int accumulate, void* ws, void* stream);

/* per-(n,c) sums over V voxels of an (N,V,C) tensor: mode 0 -> (sum a, sum a^2), mode 1 -> (sum a, sum a*b);
 * out (N,C,2) doubles.  Feeds GroupNorm (buildingblocks.py:59-78) / InstanceNorm (layers.py:165). */
size_t kmh_channel_stats_ws_bytes(int N, int C);
int kmh_channel_stats(const float* a, const float* b, int mode, int N, long long V, int C, double* out, void* ws,
                      const int* only_if, void* stream);
/* stats -> scale = rstd*gamma, shift = beta - mean*rstd*gamma per (n,c); mean_rstd (N,G,2).
 * count = elements per channel (voxels).  gamma/beta NULL = instance norm without affine. */
int kmh_gn_fwd_coeffs(const double* stats, const float* gamma, const float* beta, int N, int C, int G, double count,
                      float eps, float* scale, float* shift, float* mean_rstd, float* ascale, void* stream);
/* Range scales for the fp16-split ("f16x3", terms = 2) convolutions: device float[2] = {S, 1/S}, S a power of two.
 * kmh_gn_fwd_coeffs(ascale != NULL) writes the scale of the NORMALISED tensor from the guaranteed bound
 * max|gamma| sqrt(elements per group) + max|beta|; kmh_absmax_scale measures max(max|x|, min_abs) of a tensor. */
int kmh_absmax_scale(const float* x, long long n, float min_abs, float* out2, void* stream);
/* ab (N,C,2) = (sum dxn, sum dxn*x) -> c123 (N,C,3) with dx = c1*dxn + c2*x + c3; dgamma/dbeta (C) += */
int kmh_gn_bwd_coeffs(const double* ab, const float* gamma, const float* mean_rstd, int N, int C, int G,
                      double count, float* c123, float* dgamma, float* dbeta, const int* only_if, void* stream);
/* GroupNorm backward coefficients from statistics that need no pass over the tensors (csrc/norm.hip): dstats (N,C,2)
 * with [.][0] = sum_v dxn (the epilogue statistics `stats_out` of the data-gradient launch), bhat (N,C) = sum_v dxn *
 * xhat from kmh_conv3d_wgrad_bf(w_fold = the layer's weights (Cout,Cin,27), bhat zeroed by the caller): the weight
 * gradient of ONE sample contracted with the weights IS that sum.  only_if (kmh_channel_stats, kmh_gn_bwd_coeffs) gates
 * the direct path on the device: with only_if != NULL and *only_if == 0 those calls do nothing; *fallback is set to 1
 * (and nothing else written) when some gamma is exactly 0, where dgamma cannot be recovered from xhat. */
int kmh_gn_bwd_coeffs_fold(const double* dstats, const double* bhat, const float* gamma, const float* beta,
                           const float* mean_rstd, int N, int C, int G, double count, float* c123, float* dgamma,
                           float* dbeta, int* fallback, void* stream);
/* ---- a3: ConvNet blocks with InstanceNorm kept lazy (keymorph/layers.py:137-187, norm_type "instance"; keymorph/net.py:7-36):
 *      the next convolution's loader applies IN + ReLU to the RAW (max-pooled raw) output z, so InstanceNorm3d -> ReLU ->
 *      [MaxPool3d(2)] has no forward pass of its own; its autograd, for du at the pooled resolution, zhat = z*scale + shift,
 *      g = scatter(du) [zhat > 0]:  dz = c1 g + c2 z + c3  (c123 from kmh_gn_bwd_coeffs with G = C).
 *      kmh_in_bwd_stats: out (N,C,2) doubles = (sum g, sum g z) taken at du's resolution (z = the pooled raw tensor there). */
int kmh_in_bwd_stats(const float* du, const float* z, const float* scale, const float* shift, int N, long long V, int C,
                     double* out, void* ws, void* stream);
/*      no pooling between z and u: dz (N,V,C) = c1 du [zhat > 0] + c2 z + c3; dz may alias du; dz_scale2 | NULL = {S, 1/S} */
int kmh_in_bwd_apply(const float* du, const float* z, const float* scale, const float* shift, const float* c123, int N,
                     long long V, int C, float* dz, float* dz_scale2, void* stream);
/*      through MaxPool3d(2): du (N,D/2,H/2,W/2,C), winners of kmh_maxpool3d_fwd on the raw z (N,D,H,W,C) -> dz (N,D,H,W,C) */
int kmh_in_bwd_apply_pool(const unsigned char* argmax, const float* du, const float* z, const float* scale,
                          const float* shift, const float* c123, int N, int D, int H, int W, int C, float* dz,
                          float* dz_scale2, void* stream);

/* dx_scale2 (float[2], ZERO on entry)|NULL: also emits {S, 1/S}, the f16x3 range scale of dx (what
 * kmh_absmax_scale(dx) would return), so the consumer convolution's backward needs no extra pass over its incoming
 * gradient. */
int kmh_gn_bwd_apply(const float* dxn, const float* x, const float* c123, int N, long long V, int C,
                     int relu_mask, int accumulate, float* dx, float* dx_scale2, int out_blocked, void* stream);
int kmh_relu_mask(const float* dy, const float* y, long long n, float* dz, void* stream);
/* y = act(x*scale[n,c] + shift[n,c]) on (N,V,C): InstanceNorm3d(+ReLU) apply of keymorph/layers.py:165,183-185 */
int kmh_norm_apply(const float* x, const float* scale, const float* shift, int N, long long V, int C, int relu,
                   float* y, void* stream);

/* MaxPool3d(2) (buildingblocks.py:363, layers.py:176), NDHWC */
/* argmax (N,D/2,H/2,W/2,C) bytes | NULL: index 0..7 of each window's first maximum (scan order z, y, x), for the
 * backward */
int kmh_maxpool3d_fwd(const float* x, float* y, unsigned char* argmax, int N, int D, int H, int W, int C, void* stream);
/* dx = scatter(dy) [+ add]: the winner of each window from `argmax` (x may then be NULL: a full-resolution read less)
 * or by rescanning x; add (N,D,H,W,add_cstride >= C)|NULL is a second gradient of x summed in the same pass
 * (U-Net skip connection; may alias dx).  Odd D/H/W: the caller pre-fills the window-less trailing planes. */
int kmh_maxpool3d_bwd(const float* x, const unsigned char* argmax, const float* dy, const float* add, int add_cstride,
                      float* dx, int N, int D, int H, int W, int C, int out_blocked, void* stream);
/* MaxPool3d(2)'s backward written PRE-SPLIT for kmh_conv3d_fwd_bf(in_blocked = 2): dxs is (N, C/8, D*H*W + 1) records of 32
 * bytes (kmh_maxpool3d_bwd_split_bytes), dy_scale2 = {S, 1/S} the range scale of dy (a scatter keeps it); even D, H, W,
 * C % 8 == 0; autograd of max_pool3d (keymorph/unet3d/buildingblocks.py:321-380) in the consumer's operand format. */
size_t kmh_maxpool3d_bwd_split_bytes(int N, int D, int H, int W, int C);
int kmh_maxpool3d_bwd_split(const unsigned char* argmax, const float* dy, const float* dy_scale2, float* dxs, int N, int D,
                            int H, int W, int C, void* stream);
/* dx = scatter(dy) + [x > 0] (c1 dxn + c2 x + c3): the pooling backward summed with the skip connection's gradient whose
 * GroupNorm backward (c123 (N,C,3), kmh_gn_bwd_apply's coefficients) is applied on the fly -- autograd of max_pool3d plus
 * the decoder's skip (keymorph/unet3d/buildingblocks.py:363, 471-475) in one pass; even D, H, W, C % 4 == 0, dense
 * tensors; dx_scale2 | NULL receives {S, 1/S} for max |dx| */
int kmh_maxpool3d_bwd_lazy(const unsigned char* argmax, const float* dy, const float* dxn, const float* x, const float* c123,
                           float* dx, int N, int D, int H, int W, int C, float* dx_scale2, void* stream);
/* decoder join: out = cat(skip, nearest_upsample(low -> skip size)) (buildingblocks.py:471-475,568-582) */
int kmh_upcat_fwd(const float* skip, const float* low, float* out, int N, int D, int H, int W, int Cs, int Dl,
                  int Hl, int Wl, int Cl, void* stream);
/* dskip == NULL: only dlow is produced (the caller reads dout[..., :Cs] in place, see kmh_maxpool3d_bwd's add) */
int kmh_upcat_bwd(const float* dout, float* dskip, float* dlow, int N, int D, int H, int W, int Cs, int Dl, int Hl,
                  int Wl, int Cl, int accumulate_skip, void* stream);
/* (N,C,V) <-> (N,V,C) */
int kmh_layout_convert(const float* in, float* out, int N, long long V, int C, int to_ncdhw, void* stream);

/* final 1x1x1 conv (keymorph/unet3d/model.py:96-99,387-391): x NDHWC (N,V,Cin) -> y NCDHW (N,Cout,V) */
int kmh_pointwise_pack(const float* w, float* wt, int Cout, int Cin, void* stream);
int kmh_pointwise_fwd(const float* x, const float* wt, const float* bias, float* y, int N, long long V, int Cin,
                      int Cout, void* stream);
int kmh_pointwise_dgrad(const float* dy, const float* w, float* dx, int N, long long V, int Cin, int Cout,
                        void* stream);
size_t kmh_pointwise_wgrad_ws_bytes(int N, long long V, int Cin, int Cout);
int kmh_pointwise_wgrad(const float* dy, const float* x, float* dw, float* dbias, int N, long long V, int Cin,
                        int Cout, int accumulate, void* ws, void* stream);

/* Fused keypoint head: final 1x1x1 conv + bias -> ReLU -> center of mass (keymorph/unet3d/model.py:387-391 +
 * keymorph/layers.py:92-134) without materialising the (N,K,D,H,W) heat-map; the backward recomputes the logits.
 * feat (N,D,H,W,Cin) NDHWC with Cin <= 64, w (Cout,Cin), bias (Cout)|NULL, pts (N,Cout,3) ij order,
 * sums (N,Cout,4) = {m, mz, my, mx} kept for the backward. */
size_t kmh_headcom_fwd_ws_bytes(int N, long long V, int Cout);
size_t kmh_headcom_bwd_ws_bytes(int N, long long V, int Cin, int Cout);
int kmh_headcom_fwd(const float* feat, const float* w, const float* bias, float* pts, float* sums, float* sq, int N, int D,
                    int H, int W, int Cin, int Cout, void* ws, void* stream);
/* dpower (N,Cout)|NULL: gradient of the loss w.r.t. sums[..., 0] = sum relu(h) (keypoint weighting by power,
 * keymorph/model.py:96-109), added to the center-of-mass gradient inside the same pass. */
/* mask_dfeat != 0: dfeat is zeroed where feat <= 0 (feat is a ReLU output, so its producer's backward receives the
 * gradient already masked and skips its own mask pass). */
int kmh_headcom_bwd(const float* dpts, const float* dpower, const float* feat, const float* w, const float* bias,
                    const float* sums, float* dfeat, float* dw, float* dbias, int N, int D, int H, int W, int Cin,
                    int Cout, int mask_dfeat, void* ws, void* stream);
/* split-operand MFMA arithmetic (terms = 2: scaled f16x3, the default of the Python host; terms = 3: bf16x6), same
 * contracts; Cin % 4 == 0 */
size_t kmh_headcom_fwd_bf_ws_bytes(int N, long long V, int Cout, int terms);
size_t kmh_headcom_bwd_bf_ws_bytes(int N, long long V, int Cin, int Cout, int terms);
/* scales_out (float[4]) | NULL: the {S, 1/S} range scales of feat and w measured by this call (terms = 2); handed back
 * as scales_in, the backward skips its two measuring passes.  dfeat_scale2 (float[2], ZERO on entry) | NULL: also emits
 * the range scale of dfeat (what kmh_absmax_scale(dfeat) would return) for the consumer convolution's backward.
 * mask (kmh_headcom_mask_words(N, D, H, W, Cout) 32-bit words; 0 words = this geometry has no mask path) | NULL: the
 * forward also stores [h > 0], 1 bit per (voxel, keypoint channel) -- the ReLU of keymorph/layers.py:99 -- and a backward
 * handed the same buffer skips recomputing the logits (half its matrix work); with NULL the backward recomputes them. */
size_t kmh_headcom_mask_words(int N, int D, int H, int W, int Cout);
int kmh_headcom_fwd_bf(const float* feat, const float* w, const float* bias, float* pts, float* sums, float* sq,
                       float* scales_out, int N, int D, int H, int W, int Cin, int Cout, int terms, unsigned* mask, void* ws,
                       void* stream);
int kmh_headcom_bwd_bf(const float* dpts, const float* dpower, const float* feat, const float* w, const float* bias,
                       const float* sums, float* dfeat, float* dw, float* dbias, int N, int D, int H, int W, int Cin,
                       int Cout, int terms, int mask_dfeat, const float* scales_in, float* dfeat_scale2,
                       const unsigned* mask, void* ws, void* stream);

/* caller-side optimizer (scripts/run.py:439 torch.optim.Adam): one fused launch over a flat buffer.
 * g is multiplied by grad_scale first (1/world_size after the RCCL sum all-reduce). */
int kmh_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                  float eps, int step, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KEYMORPH_HIP_H */
