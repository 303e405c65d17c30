/*
 * libdis_hip.so -- C ABI of the MI355X (gfx950) DIS-SF / DIS-MF training-step kernels.
 *
 * This is the drop-in boundary.  The reference's only native seam is the
 * pybind module pair `ext_cpu` / `ext_cuda` of connecting_the_dots' torchext, bound at
 *   /root/reference/model/ext_functions.py:32-39   (import by bare name from CTD_DIR/torchext)
 *   /root/reference/model/ext_functions.py:124,126 (photometric_loss_forward)
 *   /root/reference/model/ext_functions.py:137,139 (photometric_loss_backward)
 * `dis_photometric_fwd/bwd` replace exactly those two entry points.  Every other function below
 * replaces a group of ATen/cuDNN operator calls inside model/networks.py and
 * model/multi_frame_networks.py (cited per function); the reference has no native code for them.
 *
 * Conventions
 *  - plain C symbols, raw DEVICE pointers, sizes as int, no torch types.
 *  - the caller owns every buffer (inputs, outputs, workspaces); the library never allocates,
 *    frees or synchronises.  All calls are asynchronous on `stream` (a hipStream_t passed as
 *    void*) and graph-capturable.  Process-wide state is limited to: (1) the operand split of the
 *    `*_bf16x3*` entry points (dis_set_conv_split / DIS_CONV_SPLIT: the entry points keep their
 *    round-1 names, the arithmetic they run is two-term fp16 by default, three-term bf16 on request
 *    - dis_get_conv_split() tells which); (2) one-time hipFuncSetAttribute flags per kernel;
 *    (3) the diagnostic tag of dis_last_kernel().  None of it is per-call data: calls from several
 *    host threads on different streams are safe as long as no thread changes the split meanwhile.
 *  - return value: 0 on success, DIS_ERR_* (<0) on a rejected argument, or the positive hipError_t
 *    of a failed launch.  Nothing throws across the boundary.
 *  - "planar" tensors are (N,C,H,W) contiguous fp32 (the reference layout, used for the 1-3 channel
 *    images/disparities/flows of the module API).  "nhwc" tensors are (N,H,W,C) contiguous fp32 and
 *    are used for every feature map inside the networks (one pixel of 32 channels = one 128-B line).
 *  - reductions to a scalar accumulate in fp64 in a caller-provided, ZEROED `double` workspace.
 */
#ifndef DIS_HIP_H
#define DIS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define DIS_OK 0
#define DIS_ERR_BAD_SHAPE (-1)
#define DIS_ERR_UNSUPPORTED (-2)
#define DIS_ERR_NULL (-3)

#define DIS_ACT_NONE 0
#define DIS_ACT_SELU 1
#define DIS_ACT_RELU 2
/* OR-ed into the `act` argument of dis_conv2d_fwd: y = act(y_previous + conv(x) + bias) (sum of convolutions over
 * several input tensors = a convolution over their channel concatenation; accumulation of input gradients). */
#define DIS_CONV_ACCUM 0x100

/* ABI version of this header; bumped on any signature change. */
int dis_abi_version(void);

/* ---------------------------------------------------------------- per-pixel operators ------- */

/* Local contrast normalisation, reference model/networks.py:663-689 (LCN.tforward).
 * x, out_lcn, out_std: (n,1,h,w).  reflect pad `radius`, (2r+1)^2 box sums. */
int dis_lcn_fwd(const float* x, float* out_lcn, float* out_std, int n, int h, int w, int radius, float eps,
                void* stream);

/* Photometric window loss, replaces ext_{cpu,cuda}.photometric_loss_forward
 * (reference model/ext_functions.py:124,126; arithmetic per :156-183).
 * es, ta: (n,c,h,w); out: (n,1,h,w).  type: 0 mse, 1 sad, 2 census_mse, 3 census_sad. block odd <= 15. */
int dis_photometric_fwd(const float* es, const float* ta, float* out, int n, int c, int h, int w, int block,
                        int type, float eps, void* stream);
/* replaces photometric_loss_backward (reference model/ext_functions.py:137,139): gradient wrt `es` only. */
int dis_photometric_bwd(const float* es, const float* ta, const float* grad_out, float* grad_es, int n, int c,
                        int h, int w, int block, int type, float eps, void* stream);
/* The census forms (type 2 / 3, 9 x 9 window, one channel) for S <= 4 estimates against ONE target in a single launch: DIS-SF
 * compares its four output scales with the same LCN image (reference model/single_frame_worker.py:110-118), and the soft sign of the
 * target differences is the same for all of them.  es, out, grad_out, grad_es: (s, n, 1, h, w) stacked; ta (n, 1, h, w).
 * Equal to s calls of dis_photometric_fwd / _bwd to rounding.  Other types / windows: DIS_ERR_UNSUPPORTED. */
int dis_photometric_fwd_multi(const float* es, const float* ta, float* out, int s, int n, int h, int w, int block, int type,
                              float eps, void* stream);
int dis_photometric_bwd_multi(const float* es, const float* ta, const float* grad_out, float* grad_es, int s, int n, int h,
                              int w, int block, int type, float eps, void* stream);


/* Pattern projection of RectifiedPatternSimilarityLoss, reference model/networks.py:358-367:
 * proj[n,y,x] = bilinear(pattern, x - disp[n,y,x], y), align_corners=True, padding 'border'.
 * pattern: (1,1,h,w) shared by all n.  disp, proj: (n,1,h,w). */
int dis_pattern_warp_fwd(const float* pattern, const float* disp, float* proj, int n, int h, int w, void* stream);
/* grad_disp = -grad_proj * d(sample)/dx   (the only path by which the photometric loss reaches the net). */
int dis_pattern_warp_bwd(const float* pattern, const float* disp, const float* grad_proj, float* grad_disp, int n,
                         int h, int w, void* stream);

/* val = sum(w*x)/sum(w) (reference model/networks.py:374), or mean(|x-y|) style reductions.
 * acc: 2 zeroed doubles {sum(w*x), sum(w)}; out: 1 float.  w may be NULL (=> plain mean over count). */
int dis_weighted_mean_fwd(const float* x, const float* w, double* acc, float* out, long count, void* stream);
/* grad_x = gscale[0] * w / sum(w)  (acc as left by the forward). */
int dis_weighted_mean_bwd(const float* w, const double* acc, const float* gscale, float* grad_x, long count,
                          void* stream);

/* mean(|a-b|)  (warm-up / pseudo-GT L1, reference model/multi_frame_worker.py:164, single_frame_worker.py:154) */
int dis_l1_mean_fwd(const float* a, const float* b, double* acc, float* out, long count, void* stream);
/* grad_a = gscale[0] * sign(a-b) / count */
int dis_l1_mean_bwd(const float* a, const float* b, const float* gscale, float* grad_a, long count, void* stream);

/* Real-data warm-up term (reference model/multi_frame_worker.py:168-173, single_frame_worker.py:158-163):
 * valid = sgm_disp > thresh (30); out = sum(|out_disp - sgm_disp + noise| * valid) / sum(valid).  noise is the caller's
 * 1.5 * N(0,1) draw (the reference draws it on the host generator and copies it over).  acc: 2 zeroed doubles. */
int dis_sgm_l1_fwd(const float* out_disp, const float* sgm_disp, const float* noise, float thresh, double* acc,
                   float* out, long count, void* stream);
int dis_sgm_l1_bwd(const float* out_disp, const float* sgm_disp, const float* noise, float thresh, const double* acc,
                   const float* gscale, float* grad_out_disp, long count, void* stream);

/* DisparitySmoothLoss, reference model/networks.py:411-431 with SobelFilter(ksize=5) :693-731.
 * disp, amb: (n,1,h,w).  acc: 1 zeroed double; out: 1 float = mean over (n,2,h,w). */
int dis_smooth_loss_fwd(const float* disp, const float* amb, double* acc, float* out, int n, int h, int w,
                        void* stream);
/* grad_disp is OVERWRITTEN (the scatter through the 5x5 taps is done as a gather, no atomics).
 * workspace: 2*n*h*w floats. */
int dis_smooth_loss_bwd(const float* disp, const float* amb, const float* gscale, float* grad_disp,
                        float* workspace, int n, int h, int w, void* stream);

/* DispToDepth, reference model/networks.py:311-319: depth = bf / (relu(disp) + 1e-12) */
int dis_disp_to_depth_fwd(const float* disp, float* depth, float bf, long count, void* stream);
int dis_disp_to_depth_bwd(const float* disp, const float* grad_depth, float* grad_disp, float bf, long count,
                          void* stream);

/* One direction of the flow-consistency (geometric) loss:
 *   multi-frame  reference model/networks.py:564-601  (primary_depth1 != NULL: adds the rf mask)
 *   single-frame reference model/networks.py:619-655  (clamp > 0: diff clamped to [0,clamp])
 * depth0, depth1, amb0, amb1, primary_depth1: (bs,1,h,w); flow0, flow1: (bs,2,h,w) pixel units;
 * R0,R1: (bs,3,3); t0,t1: (bs,3) device pointers; K: 9 floats, Kinv: 9 floats (host values, by pointer to host).
 * mask_out: (bs,1,h,w) {0,1} loss mask (saved for backward); acc: dis_geo_loss_acc_doubles() doubles of scratch (no zeroing):
 * acc[0], acc[1] = {sum(diff*mask), sum(mask)} after the call (dis_geo_loss_bwd reads them), behind them the per-block partial
 * sums they are formed from in a fixed order; out: 1 float = acc0/(acc1+1e-8). */
long dis_geo_loss_acc_doubles(void);
int dis_geo_loss_fwd(const float* depth0, const float* depth1, const float* flow0, const float* flow1,
                     const float* amb0, const float* amb1, const float* primary_depth1, const float* R0,
                     const float* t0, const float* R1, const float* t1, const float* K_host,
                     const float* Kinv_host, float clamp, float* mask_out, double* acc, float* out, int bs, int h,
                     int w, void* stream);
/* grad_depth0 (bs,1,h,w) is WRITTEN (+=); grad_depth1 (bs,1,h,w) is scatter-added (atomics).  Both must be
 * valid accumulators (zeroed or holding earlier contributions). */
int dis_geo_loss_bwd(const float* depth0, const float* depth1, const float* flow0, const float* R0,
                     const float* t0, const float* R1, const float* t1, const float* K_host,
                     const float* Kinv_host, float clamp, const float* mask, const double* acc,
                     const float* gscale, float* grad_depth0, float* grad_depth1, int bs, int h, int w,
                     void* stream);

/* Round 5: ALL directional terms of a step - tl (tl - 1) = 12 (reference model/multi_frame_worker.py:139-158, single_frame_worker.py
 * :126-149; model/networks.py:554-661) - in one forward and one backward launch.  terms: HOST array of nterms <= 16 tables of DEVICE
 * pointers, argument for argument those of dis_geo_loss_fwd / _bwd (pdepth1 may be NULL: no primary-depth mask; flow1 / amb0 / amb1
 * are not read by the backward; gdepth0 / gdepth1 not by the forward).  acc: dis_geo_loss_multi_acc_doubles(nterms) doubles (kept for
 * the backward); out (nterms) floats: the values of nterms single calls, bit for bit.  Backward: gscale (nterms) device floats;
 * both depth gradients are ADDED with float atomics into gdepth0 / gdepth1 (zeroed or pre-filled by the caller; terms may share
 * them): equal to the single calls to rounding. */
typedef struct DisGeoTerm {
  const float *depth0, *depth1, *flow0, *flow1, *amb0, *amb1, *pdepth1, *R0, *t0, *R1, *t1;
  float *mask, *gdepth0, *gdepth1;
} DisGeoTerm;
long dis_geo_loss_multi_acc_doubles(int nterms);
int dis_geo_loss_fwd_multi(const DisGeoTerm* terms, int nterms, const float* K_host, const float* Kinv_host, float clampv, double* acc,
                           float* out, int bs, int h, int w, void* stream);
int dis_geo_loss_bwd_multi(const DisGeoTerm* terms, int nterms, const float* K_host, const float* Kinv_host, float clampv,
                           const double* acc, const float* gscale, int bs, int h, int w, void* stream);

/* ---------------------------------------------------------------- layout / resize / warp ---- */

/* Pack up to 4 planar single-channel sources (n,1,h,w) into nhwc C=4 (NULL source => zeros).
 * Used for the FuseNet stem input cat(ir(2), amb, d) (reference multi_frame_networks.py:217,273). */
int dis_pack4_nhwc(const float* s0, const float* s1, const float* s2, const float* s3, float* out, int n, int h,
                   int w, void* stream);
/* same with a per-source sample stride (floats), e.g. the two planes of an (n,2,h,w) tensor: stride 2*h*w */
int dis_pack4_nhwc_strided(const float* s0, long st0, const float* s1, long st1, const float* s2, long st2,
                           const float* s3, long st3, float* out, int n, int h, int w, void* stream);
/* planar (n,c,h,w) <-> nhwc (n,h,w,c) */
int dis_planar_to_nhwc(const float* x, float* y, int n, int c, int h, int w, void* stream);
int dis_nhwc_to_planar(const float* x, float* y, int n, int c, int h, int w, void* stream);

/* Bilinear resize (F.interpolate mode='bilinear'), reference multi_frame_networks.py:47,63,246,420,422
 * (align_corners=1) and networks.py:273-293 (align_corners=0).  nhwc, any c.
 * scale_x/scale_y multiply channel 0 / channel 1 of the result when c==2 && flow_scale!=0
 * (resize_flow_like, reference multi_frame_networks.py:64-65). */
int dis_resize_bilinear_nhwc_fwd(const float* x, float* y, int n, int hin, int win, int hout, int wout, int c,
                                 int align_corners, void* stream);
int dis_resize_bilinear_nhwc_bwd(const float* gy, float* gx /*zeroed*/, int n, int hin, int win, int hout, int wout,
                                 int c, int align_corners, void* stream);
int dis_resize_bilinear_planar_fwd(const float* x, float* y, int nc, int hin, int win, int hout, int wout,
                                   int align_corners, float scale0, float scale1, int c_for_scale, void* stream);
int dis_resize_bilinear_planar_bwd(const float* gy, float* gx, int nc, int hin, int win, int hout, int wout,
                                   int align_corners, void* stream);

/* gather_warped_feat for all targets, reference multi_frame_networks.py:347-360 + warp :83-99.
 * feat: (tl,bs,h,w,c) nhwc; flows: (tl*tl, bs, h, w, 2) nhwc, entry [i*tl+j] = flow_ij (diagonal unused).
 * out: (tl, bs, h, w, tl, c): slot 0 = own frame, slots 1.. = other frames in increasing index, warped. */
int dis_gather_warped_feat_fwd(const float* feat, const float* flows, float* out, int tl, int bs, int h, int w,
                               int c, void* stream);
/* grad_feat is OVERWRITTEN: own-frame slot by plain stores, then the warped slots scatter-add (float atomics). */
int dis_gather_warped_feat_bwd(const float* grad_out, const float* flows, float* grad_feat, int tl, int bs, int h,
                               int w, int c, void* stream);

/* Deterministic, atomic-free form of the same backward.  dis_gather_csr_build turns `flows` into a CSR index by
 * destination pixel (lists sorted by source row) once per step and resolution: csr = dis_gather_csr_workspace(...)
 * int32 words; every dis_gather_warped_feat_bwd_csr call with the same flows then gathers whole rows of grad_out with
 * plain loads (grad_feat OVERWRITTEN, bitwise reproducible).  `init` (optional, may alias grad_feat) is a second
 * gradient of `feat` (e.g. its residual branch) that is added in the same pass. */
long dis_gather_csr_workspace(int tl, int bs, int h, int w);
int dis_gather_csr_build(const float* flows, int* csr, int tl, int bs, int h, int w, void* stream);
int dis_gather_warped_feat_bwd_csr(const float* grad_out, const int* csr, const float* init, float* grad_feat, int tl,
                                   int bs, int h, int w, int c, void* stream);
/* Round 6 - dis_gather_warped_feat_bwd_csr when `feat` IS y = act(GroupNorm(x2) + residual) (a Block2D3D / ResNetBlock output,
 * reference model/multi_frame_networks.py:428-430, 540-542) and this launch completes the gradient wrt y: grad_feat receives
 * gres = g act'(y) (the pre-activation gradient = the residual gradient), ab_out (tl * bs, slots, 2, c) doubles - ZEROED by the caller,
 * slots = dis_conv2d_gnsums_slots() - the channel sums dis_gn_bwd_coef takes.  Replaces the dis_gn_bwd_res_sums pass over g, y, x2.
 * DIS_ERR_UNSUPPORTED: no tiled instance (the caller runs the two launches). */
int dis_gather_warped_feat_bwd_csr_gnres(const float* grad_out, const int* csr, const float* init, float* grad_feat, const float* y,
                                         const float* x2, double* ab_out, int slots, int act, int tl, int bs, int h, int w, int c,
                                         void* stream);

/* unproject + change_view_angle + gather_warped_xyz + forward/backward flow mask for every target,
 * reference multi_frame_networks.py:172-214,283-294.  No gradient.
 * depth_core: (tl,bs,h,w); R (tl,bs,3,3), t (tl,bs,3) device; Kinv_host 9 floats;
 * ray uses the pixel coordinates (u_step*x, v_step*y) (even full-res pixels for the core grid).
 * out: (tl, bs, h, w, tl, 4) = xyz + mask per slot. */
int dis_mf_geometry(const float* depth_core, const float* R, const float* t, const float* flows,
                    const float* Kinv_host, int u_step, int v_step, float* out, int tl, int bs, int h, int w,
                    void* stream);
/* quarter-resolution version: bilinear (align_corners) resize of `geom` and mask re-binarised with >0.5
 * (reference multi_frame_networks.py:393-394). */
int dis_mf_geometry_resize(const float* geom, float* out, int tl, int bs, int hin, int win, int hout, int wout,
                           void* stream);

/* ---------------------------------------------------------------- dense layers -------------- */

/* Repack OIHW weights (cout,cin_real,k,k) into the LDS fragment order of dis_conv2d_fwd.
 * mode 0 = forward: the packed conv has cin_pad >= cin_real input channels (extra ones get zero weights,
 *          e.g. the 1-channel ambient conv runs as a 4-channel conv on a zero-padded nhwc4 input).
 * mode 1 = stride-1 input gradient: spatially flipped, cin<->cout swapped (cin_pad must equal cin_real).
 * packed has k*k*cin_pad*cout floats. */
int dis_conv2d_pack_weights(const float* w_oihw, float* packed, int cout, int cin_real, int cin_pad, int k,
                            int mode, void* stream);

/* Implicit-GEMM convolution on the matrix cores (v_mfma_f32_16x16x4_f32: fp32 in, fp32 accumulate), nhwc.
 * Replaces ZeroPad2d + Conv2d (+SELU/ReLU) of reference multi_frame_networks.py:159-164,330-345,514-542
 * and Conv2d of networks.py:222-234.
 * x: (n,hin,win,cin); y: (n,hout,wout,cout), hout = (hin + 2*pad - k)/stride + 1.
 * bias may be NULL.  act: DIS_ACT_*.  stats (optional, may be NULL): (n,2) zeroed doubles receiving
 * sum / sum of squares of y per sample (GroupNorm(1 group) statistics of the conv output).
 * The stride-1 input gradient is the same call on gy with mode-1 packed weights (cin/cout swapped). */
int dis_conv2d_fwd(const float* x, const float* w_packed, const float* bias, float* y, double* stats, int n,
                   int hin, int win, int cin, int cout, int k, int stride, int pad, int act, void* stream);
/* The same convolution for 3x3 stride-1 layers with 16 or 32 input and output channels (the FuseNet ResNet blocks; also
 * their input gradients with mode-1 weights) computed on the bf16 matrix cores at fp32 accuracy: every operand is split
 * into three bf16 terms (24 significant bits) and each product is accumulated as six bf16 x bf16 terms in the fp32 MFMA
 * accumulator; the dropped terms are <= 2^-24 relative, one fp32 rounding.
 * dis_conv2d_pack_bf16x3_size(cin, cout): 16-bit words of `packed` (negative: unsupported shape).
 * dis_conv2d_pack_weights_bf16x3: w_oihw is (cout, cin, 3, 3) for mode 0 and (cin, cout, 3, 3) for mode 1 (the weights
 * of the convolution whose input gradient is computed: channels swapped, taps flipped). */
long dis_conv2d_pack_bf16x3_size(int cin, int cout);
int dis_conv2d_pack_weights_bf16x3(const float* w_oihw, void* packed, int cout, int cin, int k, int mode, void* stream);
int dis_conv2d_fwd_bf16x3(const float* x, const void* w_packed, const float* bias, float* y, double* stats, int n, int hin,
                          int win, int cin, int cout, int k, int stride, int pad, int act, void* stream);
/* The same with the module's OIHW fp32 weights (w_o, w_i, 3, 3) handed over as they are: the kernel splits them into its
 * LDS-resident bf16 planes itself, which saves the packing launch per convolution call.  mode 0: forward (cout == w_o,
 * cin >= w_i: x may carry zero-padded extra channels); mode 1: input gradient of that convolution (cin == w_o,
 * cout >= w_i).  w_row_stride: floats between consecutive w_o rows (0 = dense = w_i*9; larger when w_oihw points at
 * w[:, a:b] inside a wider weight, as for the parts of a convolution over concatenated inputs). */
int dis_conv2d_fwd_bf16x3_oihw(const float* x, const float* w_oihw, int mode, int w_o, int w_i, int w_row_stride,
                               const float* bias, float* y, double* stats, int n, int hin, int win, int cin, int cout,
                               int k, int stride, int pad, int act, void* stream);
/* Input gradient of such a convolution when it was followed by an activation, with the activation's gradient fused in:
 * gx (+)= conv_T(gy * act'(y), w), y (same shape as gy) = the activation's output, w_oihw (w_o, w_i, 3, 3) the forward
 * weight (gy has cin == w_o channels, gx cout >= w_i).  Replaces dis_act_bwd + the mode-1 call above. */
int dis_conv2d_dgrad_bf16x3_act(const float* gy, const float* y, int act, const float* w_oihw, int w_o, int w_i,
                                int w_row_stride, float* gx, int n, int hin, int win, int cin, int cout, int pad,
                                int accumulate, void* stream);

/* dis_conv2d_fwd with per-pixel multipliers fused into the kernel (both optional, may be NULL):
 *   xscale (n,hin,win,NCHUNK): x[pixel][chunk c] is multiplied by xscale[pixel][c] while it is staged (NCHUNK = cin/32
 *          for cin % 32 == 0): conv(x * s) without materialising x * s;
 *   yscale (n,hout,wout,cout/32): the result (before bias / accumulation) of 32-channel group g is multiplied by
 *          yscale[pixel][g]: the input gradient of such a conv.
 * Used for the multi-frame 1x1 conv over the mask-weighted slots (reference multi_frame_networks.py:410-413) with
 * xscale = yscale = dis_slot_weights(geom). */
int dis_conv2d_fwd_scaled(const float* x, const float* xscale, const float* w_packed, const float* bias, float* y,
                          const float* yscale, double* stats, int n, int hin, int win, int cin, int cout, int k,
                          int stride, int pad, int act, void* stream);
/* wgt (pixels,tl) = mask / mean(mask) from geom (pixels,tl,4) */
int dis_slot_weights(const float* geom, float* wgt, long pixels, int tl, void* stream);

/* Weight/bias gradient.  gy: (n,hout,wout,cout) gradient wrt the PRE-activation output; x has cin_pad channels.
 * workspace: dis_conv2d_wgrad_workspace(cin_pad,cout,k,stride) floats (-1 if the shape is unsupported).
 * grad_w: (cout,cin_real,k,k) OIHW, grad_b: (cout) or NULL, both OVERWRITTEN.  Deterministic (partial slabs
 * per workgroup summed in a fixed order, no float atomics). */
long dis_conv2d_wgrad_workspace(int cin_pad, int cout, int k, int stride);
/* Round 6 - dis_conv2d_wgrad with gy * act'(y) formed while gy is staged (y = the conv's activated output): the layers whose input
 * gradient is never needed (FuseNet's stems conv1 4 -> 16 k4 s2 and amb_conv 4 -> 16 k3 s1, reference
 * model/multi_frame_networks.py:216-233) no longer need the dis_act_bwd pass.  workspace: dis_conv2d_wgrad_workspace(cin_pad, cout, k,
 * stride) floats.  DIS_ERR_UNSUPPORTED for any other shape. */
int dis_conv2d_wgrad_act(const float* x, const float* gy, const float* y, int act, float* grad_w, float* grad_b, float* workspace,
                         int n, int hin, int win, int cin_pad, int cin_real, int cout, int k, int stride, int pad, void* stream);
/* dis_conv2d_wgrad of conv(x * xscale) (see dis_conv2d_fwd_scaled); xscale may be NULL */
int dis_conv2d_wgrad_scaled(const float* x, const float* xscale, const float* gy, float* grad_w, float* grad_b,
                            float* workspace, int n, int hin, int win, int cin_pad, int cin_real, int cout, int k,
                            int stride, int pad, void* stream);
/* bf16x3 form (fp32 accuracy on the bf16 matrix cores, transposing LDS reads) for k = 3, stride 1 and 16 / 32 channels on
 * either side; same arguments, workspace and determinism as dis_conv2d_wgrad. */
int dis_conv2d_wgrad_bf16x3(const float* x, const float* gy, float* grad_w, float* grad_b, float* workspace, int n,
                            int hin, int win, int cin_pad, int cin_real, int cout, int k, int stride, int pad,
                            void* stream);
/* The same for a convolution that was followed by an activation (act = DIS_ACT_SELU / RELU): gy is the gradient wrt the
 * activation's OUTPUT y and the kernel stages gy * act'(y).  Replaces dis_act_bwd + dis_conv2d_wgrad_bf16x3. */
int dis_conv2d_wgrad_bf16x3_act(const float* x, const float* gy, const float* y, int act, float* grad_w, float* grad_b,
                                float* workspace, int n, int hin, int win, int cin_pad, int cin_real, int cout, int k,
                                int stride, int pad, void* stream);
int dis_conv2d_wgrad(const float* x, const float* gy, float* grad_w, float* grad_b, float* workspace, int n,
                     int hin, int win, int cin_pad, int cin_real, int cout, int k, int stride, int pad,
                     void* stream);
/* Operand split of the dis_conv2d_*_bf16x3* entry points that take fp32 weights (round 3): 1 = two fp16 terms, 3 products per
 * MAC (22-bit operands, power-of-two block scaling; default), 0 = three bf16 terms, 6 products (>= 24 bits).  Process-wide;
 * DIS_CONV_SPLIT=bf16x3 | f16x2 sets the initial value.  Same arguments, layouts and workspaces either way. */
int dis_set_conv_split(int mode);
int dis_get_conv_split(void);
/* Diagnostics: writes the name of the kernel template family the most recent convolution entry point of this process launched
 * ("conv_f16x2_kernel<32,32>", "conv_bf16x3_kernel<32,32>", "conv_fwd_kernel (fp32 MFMA)", ...; "" if none since the last call
 * with clear != 0) into the HOST buffer `name` of `cap` bytes, NUL-terminated.  bench.py labels its per-call HIP-event times
 * with it, so that a `*_bf16x3*` call served by the two-term fp16 kernel is reported as such.  Not thread-safe, never read by a
 * compute path. */
int dis_last_kernel(char* name, int cap, int clear);
/* GroupNorm applied ON LOAD by the consuming convolution (round 3).  The reference chains Conv2d -> SELU -> GroupNorm(1, C) ->
 * Conv2d (ResNetBlock model/multi_frame_networks.py:514-542; Block2D3D conv1_1 -> conv1_2, conv2_1 -> conv2_2 :338-345) and
 * writes the normalised tensor between them; here x is the PRE-normalisation tensor and the consumer stages
 * x * (rstd * gamma_c) + (beta_c - rstd * gamma_c * mean) per sample (gn_stats (n, 2): fp64 sum / sum of squares as
 * dis_conv2d_fwd leaves them; gn_gamma / gn_beta (cin)), padding stays zero: bit for bit dis_gn_apply followed by
 * dis_conv2d_fwd_bf16x3_oihw / dis_conv2d_wgrad_bf16x3, without the read + write of the normalised tensor.
 * 3x3, stride 1, cin == cout in {16, 32}, w_oihw (w_o, w_i, 3, 3) dense or a row-strided slice; act NONE / SELU. */
int dis_conv2d_fwd_bf16x3_gn(const float* x, const double* gn_stats, const float* gn_gamma, const float* gn_beta,
                             float gn_eps, const float* w_oihw, int w_o, int w_i, int w_row_stride, const float* bias,
                             float* y, double* stats, int n, int hin, int win, int cin, int cout, int k, int stride,
                             int pad, int act, void* stream);
/* ... and its backward without the GroupNorm reduce pass: the input gradient g = conv_T(gy, w) of that conv also leaves, per
 * (sample, channel), the sums of g and of g * x over the pixels (x = gn_x: the GroupNorm's input, shaped like g) - one fp64 slot per
 * workgroup, ab_out (n, dis_conv2d_gnsums_slots(), 2, cin), ZEROED by the caller - and dis_gn_bwd_from_sums turns g, x and the sums
 * into the gradient wrt x (times act'(x) for in_act != 0), grad_gamma and grad_beta in ONE elementwise pass (reference backward of
 * torch.nn.GroupNorm(1, C), model/multi_frame_networks.py:338-345).  coef: n * (c + 2) + 4 n c + 2 floats of workspace.  Two-term fp16
 * kernels only: DIS_ERR_UNSUPPORTED under dis_set_conv_split(0) (dis_gn_apply_bwd remains the general form). */
long dis_conv2d_gnsums_slots(void);
int dis_conv2d_dgrad_bf16x3_gnsums(const float* gy, const float* w_oihw, int w_o, int w_i, int w_row_stride, float* g,
                                   const float* gn_x, double* ab_out, int n, int hin, int win, int cin, int cout, int pad,
                                   void* stream);
/* The same for the residual pattern out = SELU(GroupNorm(x2) + res) whose gradient arrives through the NEXT ResNetBlock's first conv
 * (reference model/multi_frame_networks.py:514-542): that conv's accumulating input-gradient launch computes, in place,
 * g = (g + conv_T(gy, w)) * selu'(act_y) (g arrives holding the next block's residual-branch gradient, act_y = out) and the sums of
 * g and g * gn_x (gn_x = x2); g then is the residual gradient of THIS block and, with dis_gn_bwd_from_sums(in_act = 0), gives the
 * gradient wrt x2 - no reduce pass, no separate residual-gradient write.  act_y == NULL: a plain GroupNorm output with TWO consumers
 * (Block2D3D's conv_mf GroupNorm, :338-345): g = g + conv_T(gy, w) in place, the sums of g and g * gn_x as above. */
int dis_conv2d_dgrad_bf16x3_gnsums_res(const float* gy, const float* w_oihw, int w_o, int w_i, int w_row_stride, float* g,
                                       const float* act_y, const float* gn_x, double* ab_out, int n, int hin, int win, int cin,
                                       int cout, int pad, void* stream);
/* ... and when the consumer of `out` is a conv with an activation of its own (final_conv, 32 -> 16 + SELU, :262-266):
 * g = conv_T(gy * selu'(y), w) * selu'(act_y), written; cin = 16 (gy / y channels), cout = 32 - or (round 5) cin = 32, cout = 16: the
 * 16-channel slice w[:, 32:48] of ref_conv (48 -> 32 + SELU, :248-256) behind amb_res2, w_row_stride = 48 * 9. */
int dis_conv2d_dgrad_bf16x3_act_gnsums_res(const float* gy, const float* y, const float* w_oihw, int w_o, int w_i,
                                           int w_row_stride, float* g, const float* act_y, const float* gn_x, double* ab_out,
                                           int n, int hin, int win, int cin, int cout, int pad, void* stream);
int dis_gn_bwd_from_sums(const float* g, const float* x, const double* stats, const float* gamma, const double* ab, int slots,
                         float* gx, float* grad_gamma, float* grad_beta, float* coef, int n, long hw, int c, float eps,
                         int in_act, void* stream);
/* Round 5: the two halves of dis_gn_bwd_from_sums as entry points of their own, and the input-gradient launch that applies the second
 * half on load (csrc/norm_act.hip, csrc/conv2d.hip; reference backward of GroupNorm(1, C) in front of a 3x3 conv,
 * model/multi_frame_networks.py:338-345, :514-542).
 *   dis_gn_bwd_coef: ab (n, slots, 2, c) -> coef: (n, c + 2) floats [k1_c..., kx, k0] followed (8-byte aligned) by n * 2c doubles
 *     (n * (c + 2) + 4 n c + 2 floats in all), grad_gamma / grad_beta (c); counter: one unsigned, ZERO on entry (zero again on exit).
 *   dis_gn_bwd_apply_coef: gx = act'(x) (g k1_c + x kx + k0) - the elementwise pass alone.
 *   dis_conv2d_dgrad_f16x2_gnb: gx (+)= conv_T(gpre, w) with gpre = act'(q) (g k1_c + q kx + k0) formed while g and q are staged,
 *     gpre also stored to gpre_out for the weight-gradient launch; optional channel-sum epilogue as dis_conv2d_dgrad_bf16x3_gnsums /
 *     _gnsums_res (ab_gn_x, ab_act_y, ab_out; all NULL: none).  3x3, stride 1, pad 1, c -> c, c in {16, 32}; two-term fp16 kernels only. */
/* Round 5: y = act(conv3x3(out) + bias), out = SELU(GroupNorm(x2) + res) formed while x2 and res are staged (dis_gn_apply's arithmetic,
 * bit for bit) and stored to `out` by the tiles that own each pixel: a ResNetBlock's output is written by the conv that consumes it
 * first (reference model/multi_frame_networks.py:514-542).  3x3 stride 1 pad 1; c -> c (c in {16, 32}; stats optional) or 32 -> 16 (no
 * stats); act SELU; two-term fp16 kernels only. */
int dis_conv2d_fwd_f16x2_gnres(const float* x2, const double* gn_stats, const float* gn_gamma, const float* gn_beta, float gn_eps,
                               const float* res, float* out, const float* w_oihw, int w_o, int w_i, int w_row_stride,
                               const float* bias, float* y, double* stats, int n, int hin, int win, int cin, int cout, int act,
                               void* stream);
/* ... and the same for the 1 x 1 multi-frame conv (128 -> 32 with slot weights, model/multi_frame_networks.py:406-413), whose input
 * gradient runs on the fp32 MFMA kernel: g (n, h, w, 32) the gradient wrt the GroupNorm's output, q the GroupNorm's input, coef
 * (n, 34); gx (n, h, w, 128) (+)= (gpre W^T) * yscale[pixel][32-channel group] (yscale may be NULL); gpre stored to gpre_out for
 * dis_conv2d_wgrad_scaled.  w_packed = dis_conv2d_pack_weights(mode 1). */
int dis_conv2d_dgrad1x1_scaled_gnb(const float* g, const float* q, const float* coef, int in_act, float* gpre_out,
                                   const float* w_packed, float* gx, const float* yscale, int n, int hin, int win, int cin,
                                   int cout, int accumulate, void* stream);
/* ... and for the 4 x 4 stride-2 pad-1 conv (32 -> 32, Block2D3D.conv2_1, :338-345): here the WEIGHT-gradient launch applies the pass
 * while it stages gy (every gy pixel belongs to one tile) and stores gpre for dis_conv2d_dgrad_strided.  g / q (n, hout, wout, 32),
 * coef (n, 34), x (n, hin, win, 32); workspace: dis_conv2d_wgrad_workspace(32, 32, 4, 2) floats.  Two-term fp16 kernel only. */
int dis_conv2d_wgrad_k4s2_f16x2_gnb(const float* x, const float* g, const float* q, const float* coef, int in_act, float* gpre_out,
                                    float* grad_w, float* grad_b, float* workspace, int n, int hin, int win, void* stream);
/* Round 5: FuseNet's 4 x 4 stride-2 pad-1 down convolution (32 -> 32, Block2D3D.conv2_1, model/multi_frame_networks.py:338-345)
 * forward on the two-term fp16 kernels (csrc/conv_k4s2.hip: wave-resident weights, de-interleaved halo columns): y = act(conv(x, w) +
 * bias), w OIHW (32, 32, 4, 4) unpacked, stats (n, 2) accumulated or NULL.  DIS_ERR_UNSUPPORTED under dis_set_conv_split(0). */
int dis_conv2d_fwd_k4s2_f16x2(const float* x, const float* w_oihw, const float* bias, float* y, double* stats, int n, int hin, int win,
                              int act, void* stream);
/* ... and its input gradient: gx (n, hin, win, 32) (+)= conv_transpose(gy (n, hin / 2, win / 2, 32), w), the four parity classes from
 * one gy halo tile in one launch (replaces dis_conv2d_dgrad_strided's four launches + four packing launches); hin, win even. */
int dis_conv2d_dgrad_k4s2_f16x2(const float* gy, const float* w_oihw, float* gx, int n, int hin, int win, int accumulate, void* stream);
/* Round 5: the channel sums dis_gn_bwd_coef starts from when NO convolution epilogue left them (the gradient wrt the GroupNorm's output
 * arrives from a join, a resize or a feature warp: Block2D3D's conv_fuse GroupNorm with its residual, model/multi_frame_networks.py
 * :338-345, FuseNet.res3's last GroupNorm, :514-542): g = gy * act'(y) (y / act: the GroupNorm output's activation; NULL / 0: g = gy),
 * g stored to gres when given (the residual gradient), ab_out (n, slots, 2, c) doubles = per-block sums of g and g * x, every slot
 * written.  With dis_gn_bwd_coef and an on-load elementwise pass this replaces dis_gn_apply_bwd's reduce + apply launches. */
int dis_gn_bwd_res_sums(const float* gy, const float* y, const float* x, float* gres, double* ab_out, int slots, int n, long hw, int c,
                        int act, void* stream);
int dis_gn_bwd_coef(const double* stats, const float* gamma, const double* ab, int slots, float* coef, float* grad_gamma,
                    float* grad_beta, unsigned* counter, int n, long hw, int c, float eps, void* stream);
int dis_gn_bwd_apply_coef(const float* g, const float* x, const float* coef, float* gx, int n, long hw, int c, int in_act,
                          void* stream);
int dis_conv2d_dgrad_f16x2_gnb(const float* g, const float* q, const float* coef, int in_act, float* gpre_out, const float* w_oihw,
                               int w_o, int w_i, int w_row_stride, float* gx, int accumulate, const float* ab_gn_x,
                               const float* ab_act_y, double* ab_out, int n, int hin, int win, int c, void* stream);
/* Round 6 - input gradient AND weight gradient of a 3x3 stride-1 pad-1 conv c -> c (c = 32) in ONE launch (csrc/conv_bwd_fused.hip):
 * replaces the pair dis_conv2d_dgrad_f16x2_gnb / dis_conv2d_fwd_bf16x3_oihw(mode 1) + dis_conv2d_wgrad_bf16x3[_gn | _act] of one Conv2d
 * node (reference: torch.nn.Conv2d's backward inside ResNetBlock / Block2D3D, model/multi_frame_networks.py:338-345,514-542).
 *   operand of both products: coef != NULL: gpre = act'(q) (g k1_c + q kx + k0), as dis_conv2d_dgrad_f16x2_gnb forms it (also stored to
 *     gpre_out when that is non-NULL); coef == NULL: g (in_act == 0) or g act'(q) (q = the conv's activated output).
 *   input gradient gx: dis_conv2d_dgrad_f16x2_gnb's forms (accumulate, ab_gn_x / ab_act_y / ab_out) - bit-identical results.
 *   weight gradient: x (n, hin, win, c) = the conv's input; x_gn_stats != NULL: staged as GroupNorm(x) (dis_conv2d_wgrad_bf16x3_gn);
 *     x may be the tensor ab_gn_x or ab_act_y (fetched once).  grad_w (c, c, 3, 3), rows grad_w_row_stride floats apart (0 = contiguous;
 *     a multiple of 9 >= 9 c: the slice of a wider OIHW gradient, e.g. of a conv over a channel concatenation), grad_b (c) or NULL;
 *     workspace: dis_conv2d_bwd_fused_workspace(c) floats (-1: no kernel for this channel count).
 * DIS_ERR_UNSUPPORTED: no instance for the combination / the three-term mode / DIS_BWD_FUSED=0 (the caller keeps the two launches). */
long dis_conv2d_bwd_fused_workspace(int c);
int dis_conv2d_bwd_fused_f16x2(const float* g, const float* q, const float* coef, int in_act, float* gpre_out, const float* w_oihw,
                               int w_o, int w_i, int w_row_stride, float* gx, int accumulate, const float* ab_gn_x,
                               const float* ab_act_y, double* ab_out, const float* x, const double* x_gn_stats,
                               const float* x_gn_gamma, const float* x_gn_beta, float x_gn_eps, float* grad_w, float* grad_b,
                               float* workspace, int n, int hin, int win, int c, int grad_w_row_stride, void* stream);
int dis_conv2d_wgrad_bf16x3_gn(const float* x, const double* gn_stats, const float* gn_gamma, const float* gn_beta,
                               float gn_eps, const float* gy, float* grad_w, float* grad_b, float* workspace, int n,
                               int hin, int win, int cin_pad, int cin_real, int cout, int k, int stride, int pad,
                               void* stream);
/* Split-neutral names (round 5) of the 3x3 entry points above whose `_bf16x3` suffix is history: which operand split multiplies is a
 * process-wide mode (dis_set_conv_split: 1 = two-term fp16, 3 products per MAC, the default; 0 = three-term bf16, 6 products, >= 24-bit
 * operands), not a property of the entry point.  Same symbols' code (ELF aliases), same arguments. */
int dis_conv2d_fwd_split_oihw(const float* x, const float* w_oihw, int mode, int w_o, int w_i, int w_row_stride, const float* bias, float* y, double* stats, int n, int hin, int win, int cin, int cout, int k, int stride, int pad, int act, void* stream);
int dis_conv2d_fwd_split_gn(const float* x, const double* gn_stats, const float* gn_gamma, const float* gn_beta, float gn_eps, const float* w_oihw, int w_o, int w_i, int w_row_stride, const float* bias, float* y, double* stats, int n, int hin, int win, int cin, int cout, int k, int stride, int pad, int act, void* stream);
int dis_conv2d_dgrad_split_gnsums(const float* gy, const float* w_oihw, int w_o, int w_i, int w_row_stride, float* g, const float* gn_x, double* ab_out, int n, int hin, int win, int cin, int cout, int pad, void* stream);
int dis_conv2d_dgrad_split_gnsums_res(const float* gy, const float* w_oihw, int w_o, int w_i, int w_row_stride, float* g, const float* act_y, const float* gn_x, double* ab_out, int n, int hin, int win, int cin, int cout, int pad, void* stream);
int dis_conv2d_dgrad_split_act_gnsums_res(const float* gy, const float* y, const float* w_oihw, int w_o, int w_i, int w_row_stride, float* g, const float* act_y, const float* gn_x, double* ab_out, int n, int hin, int win, int cin, int cout, int pad, void* stream);
int dis_conv2d_dgrad_split_act(const float* gy, const float* y, int act, const float* w_oihw, int w_o, int w_i, int w_row_stride, float* gx, int n, int hin, int win, int cin, int cout, int pad, int accumulate, void* stream);
int dis_conv2d_wgrad_split(const float* x, const float* gy, float* grad_w, float* grad_b, float* workspace, int n, int hin, int win, int cin_pad, int cin_real, int cout, int k, int stride, int pad, void* stream);
int dis_conv2d_wgrad_split_act(const float* x, const float* gy, const float* y, int act, float* grad_w, float* grad_b, float* workspace, int n, int hin, int win, int cin_pad, int cin_real, int cout, int k, int stride, int pad, void* stream);
int dis_conv2d_wgrad_split_gn(const float* x, const double* gn_stats, const float* gn_gamma, const float* gn_beta, float gn_eps, const float* gy, float* grad_w, float* grad_b, float* workspace, int n, int hin, int win, int cin_pad, int cin_real, int cout, int k, int stride, int pad, void* stream);

/* Input gradient of a k=4, stride=2, pad=1 convolution (transposed convolution) as four 2x2 phase convolutions on
 * the matrix cores.  w_oihw is the unpacked weight (cout,cin,4,4).  workspace: 16*cin*cout floats.
 * gx: (n,hin,win,cin) overwritten, or added to when `accumulate` != 0. */
int dis_conv2d_dgrad_strided(const float* gy, const float* w_oihw, float* gx, float* workspace, int n, int hin,
                             int win, int cin, int cout, int k, int stride, int pad, int accumulate, void* stream);

/* Head: Conv2d(cin,1,3,pad 1) + alpha*sigmoid(x - offset) (reference networks.py:121-125,140-149;
 * multi_frame_networks.py:157,265), cin = 16 or 32.  x nhwc (n,h,w,cin); w (1,cin,3,3); y planar (n,1,h,w). */
int dis_disp_head_fwd(const float* x, const float* w, const float* b, float* y, int n, int h, int wd, int cin,
                      float alpha, float offset, void* stream);
/* gy (n,1,h,w) gradient wrt y; y is the forward output.  gx (n,h,w,cin) overwritten; grad_w/grad_b overwritten.
 * workspace: dis_disp_head_bwd_workspace(n,h,wd,cin) floats (8-byte aligned: pre-sigmoid gradient + fp64 block slabs). */
long dis_disp_head_bwd_workspace(int n, int h, int wd, int cin);
int dis_disp_head_bwd(const float* x, const float* w, const float* y, const float* gy, float* gx, float* grad_w,
                      float* grad_b, float* workspace, int n, int h, int wd, int cin, float alpha, void* stream);

/* dy *= act'(y) given the activation OUTPUT y (SELU and ReLU are invertible enough for that). In place ok. */
int dis_act_bwd(const float* gy, const float* y, float* gpre, int act, long count, void* stream);
/* The same on channel ranges of wider nhwc buffers: gy / y pixels are ldg / ldy floats apart (c % 4 == 0, 16-byte
 * aligned), gpre is dense (npix, c); act == DIS_ACT_NONE gathers gy.  Host side: ops._ConvG.backward when a layer's
 * output lives inside a decoder concatenation buffer (the reference concatenates with torch.cat, networks.py:262-288). */
int dis_act_bwd_ld(const float* gy, int ldg, const float* y, int ldy, float* gpre, int act, long npix, int c,
                   void* stream);
/* dis_act_bwd_ld + the bias gradient bias_grad[ch] = sum_pixels gpre[pixel][ch] from the same pass (fp32 block partials, fp64
   totals, fixed order); workspace: dis_act_bwd_ld_bias_workspace(c) floats; c / 4 must divide 256, else DIS_ERR_UNSUPPORTED */
long dis_act_bwd_ld_bias_workspace(int c);
int dis_act_bwd_ld_bias(const float* gy, int ldg, const float* y, int ldy, float* gpre, int act, long npix, int c,
                        float* bias_grad, float* workspace, void* stream);
/* dst[pixel*ldd + j] = src[pixel*lds + j] for j < c and 0 for c <= j < c + czero: a tensor written into a channel range
 * of a wider nhwc buffer, followed by czero zero lanes (the padding of a concatenation to a multiple of 4 channels). */
int dis_copy_channels(const float* src, int lds, float* dst, int ldd, long npix, int c, int czero, void* stream);

/* GroupNorm(num_groups=1) statistics: stats (n,2) zeroed doubles <- sum, sumsq over (h,w,c) per sample. */
int dis_gn_stats(const float* x, double* stats, int n, long per_sample, void* stream);
/* y = act( gamma_c*(x-mean)*rstd + beta_c (+ residual) );  nhwc, eps 1e-5 (torch default).
 * Covers GroupNorm of reference multi_frame_networks.py:336,344,451,522,526 and the residual tail :539-540. */
int dis_gn_apply(const float* x, const double* stats, const float* gamma, const float* beta, const float* residual,
                 float* y, int n, long hw, int c, int act, float eps, void* stream);
/* Backward of dis_gn_apply.  gy: grad wrt y; y: forward output (needed when act != none); x: forward input.
 * red / gparam_acc: workspaces for per-block partial sums (no zeroing needed, no atomics; summed in a fixed order):
 * with W = dis_gn_bwd_workspace(n,c) doubles in total, red = W*2/(2+2c) doubles and gparam_acc = W*2c/(2+2c) doubles
 * (i.e. n*64*2 and n*64*2c).
 * gx overwritten; gres (optional) receives the residual-branch gradient (= gradient after act');
 * grad_gamma/grad_beta (c floats) overwritten.
 * in_act != DIS_ACT_NONE: x is the activation OUTPUT of the producing layer (conv -> SELU -> GroupNorm, reference
 * multi_frame_networks.py:338-345,533-535) and gx is additionally multiplied by act'(x), i.e. it is the gradient wrt
 * that layer's PRE-activation output (saves the separate dis_act_bwd pass). */
long dis_gn_bwd_workspace(int n, int c);
int dis_gn_apply_bwd(const float* gy, const float* y, const float* x, const double* stats, const float* gamma,
                     float* gx, float* gres, float* grad_gamma, float* grad_beta, double* red, double* gparam_acc,
                     int n, long hw, int c, int act, float eps, int in_act, void* stream);

/* y = act(a + b) elementwise and its backward (residual SELU of Block2D3D, reference :428). */
int dis_add_act_fwd(const float* a, const float* b, float* y, int act, long count, void* stream);

/* Scale the tl slots of a gathered feature tensor by mask/mean(mask) (reference multi_frame_networks.py:410):
 * wf (tl,bs,h,w,tl,c); geom (tl,bs,h,w,tl,4) (mask in .w); out same shape as wf.  Backward is the same call on
 * the gradient; accumulate != 0 adds the result to `out` instead of overwriting it. */
int dis_mask_weight_slots(const float* wf, const float* geom, float* out, long pixels, int tl, int c, int accumulate,
                          void* stream);

/* ---------------------------------------------------------------- Conv3D (k-NN continuous conv) */

/* reference multi_frame_networks.py:469-512 for ALL target frames at once, in two stages.
 * geom: (tl,bs,h,w,tl,4) xyz+mask per slot; wf: (tl,bs,h,w,tl,c) gathered features (c == 32).
 *
 * stage 1, neighbour selection (reference :489-498): per output pixel the 9 candidates with the smallest masked
 * planar distance among the 3x3 x tl window (candidate id = (ky*3+kx)*tl+slot, zero-padded border).
 * idx_out: (tl,bs,ho,wo,9) uint8.  Ties / masked fills resolve to the lowest candidate id (the reference leaves
 * them to torch.topk(sorted=False), i.e. implementation-defined).  The selection depends on the geometry
 * only, so one call serves every Conv3D layer that shares `geom`. */
int dis_conv3d_knn_select(const float* geom, unsigned char* idx_out, int tl, int bs, int h, int wd, int stride,
                          void* stream);
/* stage 2 (reference :499-508): gather, MLP 3->16->32 on the local coordinates, weighted feature sum, 32x32 mix,
 * SELU.  dense1_w (16,3), dense1_b (16), dense2_w (32,16), dense2_b (32), w (32,32).
 * y: (tl,bs,ho,wo,32) = SELU(agg @ w)  (GroupNorm follows as dis_gn_*). */
int dis_conv3d_knn_fwd(const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                       const float* dense2_w, const float* dense2_b, const float* w, const unsigned char* idx,
                       float* y, int tl, int bs, int h, int wd, int stride, void* stream);
/* The forward that also keeps agg (tl,bs,ho,wo,32), the weighted feature sums in front of the 32x32 mix, for
 * dis_conv3d_knn_bwd_det (agg may be NULL: dis_conv3d_knn_fwd). */
int dis_conv3d_knn_fwd_agg(const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                           const float* dense2_w, const float* dense2_b, const float* w, const unsigned char* idx,
                           float* y, float* agg, int tl, int bs, int h, int wd, int stride, void* stream);
/* gy: gradient wrt y (post-SELU).  grad_wf (zeroed) scatter-added (float atomics, 128-B rows).  gparams:
 * 32*32+16*3+16+32*16+32 floats overwritten in that order (w, dense1_w, dense1_b, dense2_w, dense2_b: the order of
 * the module's parameters()).
 * workspace: dis_conv3d_knn_bwd_workspace() floats (per-block partial slabs, summed in a fixed order). */
long dis_conv3d_knn_bwd_workspace(void);
int dis_conv3d_knn_bwd(const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                       const float* dense2_w, const float* dense2_b, const float* w, const unsigned char* idx,
                       const float* y, const float* gy, float* grad_wf, float* gparams, float* workspace, int tl,
                       int bs, int h, int wd, int stride, void* stream);
/* The same with a DETERMINISTIC feature gradient (no float atomics: bitwise reproducible).  csr = dis_conv3d_csr_build(idx):
 * the (output pixel, neighbour) entries of the neighbour sets grouped by the source row they selected, lists sorted by entry,
 * dis_conv3d_csr_workspace(...) ints, built once per geometry and shared by every Conv3D layer that uses the sets.  stage:
 * dis_conv3d_knn_bwd_stage(...) floats of scratch (the per-entry gradient rows).  accumulate = 0: grad_wf is written, rows
 * nobody selected get zeros (no zero fill by the caller); 1: the selected rows are added to grad_wf's contents. */
long dis_conv3d_csr_workspace(int tl, int bs, int h, int w, int stride);
int dis_conv3d_csr_build(const unsigned char* idx, int* csr, int tl, int bs, int h, int w, int stride, void* stream);
long dis_conv3d_knn_bwd_stage(int tl, int bs, int h, int wd, int stride);
int dis_conv3d_knn_bwd_csr(const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                           const float* dense2_w, const float* dense2_b, const float* w, const unsigned char* idx,
                           const float* y, const float* gy, float* grad_wf, float* gparams, float* workspace,
                           const int* csr, float* stage, int accumulate, int tl, int bs, int h, int wd, int stride,
                           void* stream);

/* The backward of Conv3D (reference :469-512 under autograd) with a DETERMINISTIC feature gradient and no index structure:
 * the output pixels are processed in 9 (stride 1) / 4 (stride 2) classes of pixels with pairwise disjoint 3x3 windows, one
 * launch per class, so the rows of grad_wf are updated with plain read-modify-writes in a fixed order.  grad_wf is ADDED to
 * (zero it, or hand in the gradient another consumer of wf has written).  agg: from dis_conv3d_knn_fwd_agg.  gparams as above.
 * workspace: dis_conv3d_knn_bwd_det_workspace(...) floats. */
long dis_conv3d_knn_bwd_det_workspace(int tl, int bs, int h, int wd, int stride);
int dis_conv3d_knn_bwd_det(const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                           const float* dense2_w, const float* dense2_b, const float* w, const unsigned char* idx,
                           const float* y, const float* agg, const float* gy, float* grad_wf, float* gparams,
                           float* workspace, int tl, int bs, int h, int wd, int stride, void* stream);
/* The same kernel in one launch over all pixels with the float-atomic scatter (not reproducible bit for bit); same arguments
 * and workspace. */
int dis_conv3d_knn_bwd_agg(const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                           const float* dense2_w, const float* dense2_b, const float* w, const unsigned char* idx,
                           const float* y, const float* agg, const float* gy, float* grad_wf, float* gparams,
                           float* workspace, int tl, int bs, int h, int wd, int stride, void* stream);

/* ---------------------------------------------------------------- general convolution family (DIS-SF) */

/* Streaming implicit-GEMM convolution on the matrix cores for the DispNetS encoder-decoder (reference
 * model/networks.py:170-295: Conv2d k7/k5/k3 stride 1/2 with ReLU :222-234, ConvTranspose2d(k3,s2,p1,op1) :236-240
 * with crop_like :242-244), channel counts 2..1024.  All tensors nhwc; a tensor argument is (pointer, ld, off):
 * a pixel occupies `ld` floats and the channels [off, off+c) are used, so channel slices of a concatenated buffer
 * can be read / written in place.  `cin`/`cout` are the channel counts processed (cin % 4 == 0); `cin_w`/`cout_w`
 * (<= cin/cout) are the channels the weight tensor really has: the rest are zero-padded lanes (read as zero weights,
 * written as zeros).
 *   DIS_CONVG_CONV        y = act(conv(x, w[cout_w][cin_w][k][k]) + bias), hout = (hin+2*pad-k)/stride+1
 *   DIS_CONVG_CONV_DGRAD  x := gradient wrt the conv's pre-activation output (n,hin,win,cin = conv cout);
 *                         y := gradient wrt the conv's input (n,hout,wout,cout = conv cin); w is the conv's weight
 *                         [cin_w][cout_w][k][k]; stride 2 runs as 4 output-parity phases
 *   DIS_CONVG_TCONV       y = act(conv_transpose(x, w[cin_w][cout_w][k][k], stride 2) + bias) cropped to (hout,wout)
 *   DIS_CONVG_TCONV_DGRAD x := gradient wrt the (cropped) transposed-conv output; y := gradient wrt its input;
 *                         w is the transposed conv's weight [cout_w][cin_w][k][k]
 * wpack: workspace of dis_convg_pack_workspace(cin,cout,k) floats (x4 for the two phase-decomposed cases:
 * CONV_DGRAD and TCONV with stride 2) PLUS dis_convg_splitk_workspace(...) floats (round 4; 0 for most calls: partial sums of
 * the split-K form small maps take under the two-term fp16 split).
 * Routing (same results, same contract): 3x3 stride-1 CONV / CONV_DGRAD calls with <= 10 32x32 channel-slice launches and
 * >= 400k pixels run on the halo-resident kernels (7x7 too, one launch per tap row, under the three-term split); with >= 32 input
 * channels and a virtual output grid of >= 1024 positions (DIS_CONVG_HALO_MIN) the call takes the LDS-halo form convh2_kernel
 * (round 4: the input halo of a tile staged once per 32-channel chunk, all taps from LDS; the four parity phases of the stride-2
 * transposed forms in one launch when their weights stay resident); everything else streams (256 x 128 tiles for >= 128 channels
 * on both sides).
 * Arithmetic (dis_set_conv_split, default 1): layers with >= 32 input channels multiply two-term fp16 operands (3 products per
 * MAC, fp32 accumulate; one power-of-two scale per image of x in the streaming kernel, per halo tile in the slice launches and in
 * the halo form, and one per weight tensor); dis_set_conv_split(0) selects the three-term bf16 split (6 products, >= 24-bit operands) everywhere. */
#define DIS_CONVG_CONV 0
#define DIS_CONVG_CONV_DGRAD 1
#define DIS_CONVG_TCONV 2
#define DIS_CONVG_TCONV_DGRAD 3
long dis_convg_pack_workspace(int cin, int cout, int k);
long dis_convg_splitk_workspace(int mode, int n, int hin, int win, int hout, int wout, int cin, int cout, int k, int stride,
                                int pad);
int dis_convg_run(int mode, const float* x, int ldx, int xoff, const float* w, const float* bias, float* y, int ldy,
                  int yoff, float* wpack, int n, int hin, int win, int cin, int cin_w, int hout, int wout, int cout,
                  int cout_w, int k, int stride, int pad, int act, void* stream);

/* Weight gradient of the family: grad_w[g][x][ky][kx] = sum_{n,gy,gx} X[n][gy*stride-pad+ky][gx*stride-pad+kx][x] *
 * G[n][gy][gx][g]  (g < cG_w, x < cX_w; OVERWRITTEN; deterministic split-K slabs).
 *   conv:            X = layer input,              G = gradient wrt the pre-activation output  -> (cout,cin,k,k)
 *   transposed conv: X = gradient wrt its output,  G = layer input                             -> (cin,cout,k,k)
 * workspace: dis_convg_wgrad_workspace(n,hG,wG,cX,cG,k) floats.
 * Routing: k in {3,5} with stride in {1,2} and k = 7 with stride 1, cX >= 16, cG >= 32 run as 32x32 channel-slice pairs of the one-pass
 * halo kernels (x and G staged once for all taps; two-term fp16 operands by default, the three-term bf16 split under
 * dis_set_conv_split(0)); other shapes on the fp32 split-K kernel (one pass per tap). */
long dis_convg_wgrad_workspace(int n, int hG, int wG, int cX, int cG, int k);
int dis_convg_wgrad(const float* X, int ldX, int xoff, int hX, int wX, int cX, int cX_w, const float* G, int ldG,
                    int goff, int hG, int wG, int cG, int cG_w, float* grad_w, float* workspace, int n, int k,
                    int stride, int pad, void* stream);

/* out[c] = sum over npix pixels of G[pixel*ldG + goff + c]  (bias gradients).  workspace: dis_colsum_workspace(c). */
long dis_colsum_workspace(int c);
int dis_colsum(const float* G, int ldG, int goff, long npix, int c, float* out, float* workspace, void* stream);

/* SigmoidAffine of the DIS-SF disparity heads, reference model/networks.py:140-149: y = alpha*sigmoid(x - offset).
 * The backward writes the pre-sigmoid gradient as a zero-padded 4-channel nhwc tensor (count,4) for dis_convg_*. */
int dis_sigmoid_affine_fwd(const float* x, float* y, float alpha, float offset, long count, void* stream);
int dis_sigmoid_affine_bwd(const float* y, const float* gy, float* gpre4, float alpha, long count, void* stream);

/* ---------------------------------------------------------------- DispNetS, bf16 activation storage -------- */

/* BASELINE config 2 ("DIS-SF bs=8 ... bf16"): the encoder-decoder's nhwc feature maps are stored as bf16, convolutions
 * are ONE bf16 x bf16 product per MAC with fp32 accumulation; parameters, their gradients, disparities and losses stay
 * fp32.  Same modes / argument meaning as dis_convg_run and dis_convg_wgrad (reference model/networks.py:170-295:
 * Conv2d k7/k5/k3 s1/s2 + ReLU, ConvTranspose2d(k3,s2,p1,op1) + crop_like), with a storage flag per tensor
 * (x_bf16 / y_bf16 / g_bf16: 1 = bf16, 0 = fp32) and ld / channel offsets counted in ELEMENTS of the tensor's own
 * type (bf16: multiples of 8, fp32: multiples of 4).  dis_convb_pack_workspace counts 16-bit words per phase (x4);
 * wpack holds 4 x that many words PLUS dis_convb_splitk_workspace(...) floats (round 4; 0 for most calls: partial sums of the
 * split-K form the small maps of the deep layers take). */
long dis_convb_pack_workspace(int cin, int cout, int k);
long dis_convb_splitk_workspace(int mode, int x_bf16, int n, int hin, int win, int hout, int wout, int cin, int cout, int k,
                                int stride, int pad);
/* One packing launch per step (round 4).  A call's packed weights depend on its weights alone, which change once per step:
 *   dis_convb_pack_record(host_descs, capacity, NULL)  every packing launch of the dis_convb_run calls that follow also leaves a
 *                                                      descriptor (dis_convb_pack_desc_bytes() bytes) in the HOST array;
 *   dis_convb_pack_record(NULL, 0, out)                stops; out[0] = descriptors seen (> capacity: the record is incomplete,
 *                                                      do not use it), out[1] = workgroups of the batch launch (its plan is
 *                                                      written into the descriptors: upload them AFTER the stop);
 *   dis_convb_pack_batch(dev_descs, count, workgroups, stream)  packs all recorded calls' weights (into the wpack buffers they were
 *                                                      recorded with: the caller keeps those alive and unchanged) in ONE launch;
 *   dis_convb_run(mode | DIS_CONVB_PREPACKED, ...)     runs a recorded call on its packed weights (w may be NULL).
 * The record is process-global host state (like dis_last_kernel), not thread-safe. */
#define DIS_CONVB_PREPACKED 0x100
long dis_convb_pack_desc_bytes(void);
int dis_convb_pack_record(void* host_descs, int capacity, int* count_out);
int dis_convb_pack_batch(const void* dev_descs, int count, int workgroups, void* stream);
int dis_convb_run(int mode, const void* x, int x_bf16, int ldx, int xoff, const float* w, const float* bias, void* y,
                  int y_bf16, int ldy, int yoff, void* wpack, int n, int hin, int win, int cin, int cin_w, int hout,
                  int wout, int cout, int cout_w, int k, int stride, int pad, int act, void* stream);
long dis_convb_wgrad_workspace(int n, int hG, int wG, int cX, int cG, int k);
int dis_convb_wgrad(const void* X, int x_bf16, int ldX, int xoff, int hX, int wX, int cX, int cX_w, const void* G,
                    int g_bf16, int ldG, int goff, int hG, int wG, int cG, int cG_w, float* grad_w, float* workspace, int n,
                    int k, int stride, int pad, void* stream);
/* gpre = gy * act'(y) on channel ranges (ldg / ldy elements per pixel) of bf16 nhwc buffers; gpre dense (npix, c) bf16 */
int dis_act_bwd_bf16(const void* gy, int ldg, const void* y, int ldy, void* gpre, int act, long npix, int c,
                     void* stream);
/* dis_act_bwd_bf16 + the bias gradient bias_grad[ch] = sum_pixels gpre[pixel][ch] from the same pass (summed in fp32 before gpre's
   rounding); workspace: dis_colsum_bf16_workspace(c) floats; c / 4 must divide 256, else DIS_ERR_UNSUPPORTED */
int dis_act_bwd_bf16_bias(const void* gy, int ldg, const void* y, int ldy, void* gpre, int act, long npix, int c,
                          float* bias_grad, float* workspace, void* stream);
/* the same with a dense fp32 result (the network's first layer: x is fp32, its weight gradient runs through dis_conv2d_wgrad) */
int dis_act_bwd_bf16_f32(const void* gy, int ldg, const void* y, int ldy, float* gpre, int act, long npix, int c,
                         void* stream);
/* dis_copy_channels into a bf16 buffer from an fp32 (src_bf16 = 0) or bf16 source */
int dis_copy_channels_bf16(const void* src, int src_bf16, int lds, void* dst, int ldd, long npix, int c, int czero,
                           void* stream);
/* dis_colsum of a bf16 tensor (bias gradients), fp32 result */
long dis_colsum_bf16_workspace(int c);
int dis_colsum_bf16(const void* G, int ldG, int goff, long npix, int c, float* out, float* workspace, void* stream);

/* ---------------------------------------------------------------- augmentation -------------- */

/* Training-time augmentation of the IR and ambient images on the device (reference data/data_manipulation.py:114-195 with
 * data/dataset.py:67-70: no affine part, 5x5 Gaussian blur sigma in [0.2,0.5], Gaussian noise <= 3/255, salt-and-pepper
 * <= 5e-4, clip to [0,1]).  im / amb / outputs: (n, h, w) planes.  params: n x 6 floats on the DEVICE
 * {blur flag, sigma_im, sigma_amb, noise_im, noise_amb, sp_ratio or < 0}; seed: one int64 on the device (both may be
 * static buffers of a captured step); minmax_ws: 2n uint32 workspace.  Not bit-comparable with numpy's generator by
 * construction; tests check the distributions. */
int dis_augment(const float* im, const float* amb, const float* params, const long long* seed, unsigned* minmax_ws,
                float* out_im, float* out_amb, int n, int h, int w, void* stream);

/* ---------------------------------------------------------------- optimiser ----------------- */

/* torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) on a flat fp32 buffer (reference train_val.py:55-56).
 * step_count is the 1-based step index; grads are multiplied by grad_scale first (1/world_size for DP).  The betas are
 * doubles: torch.optim.Adam evaluates (1 - beta) and beta^t on python floats and rounds once. */
int dis_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long count, float lr,
                  double beta1, double beta2, float eps, int step_count, float grad_scale, void* stream);
/* The same update with the step counter on the device: state = 4 x 32 bit {int steps taken, float 1-beta1^t,
 * float sqrt(1-beta2^t), unused}, advanced by the call itself.  Safe to capture in a hipGraph: replay k applies
 * step k's bias correction (dis_adam_step would replay the capture-time correction). */
int dis_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long count, float lr,
                      double beta1, double beta2, float eps, int* state, float grad_scale, void* stream);

/* ---------------------------------------------------------------- gradient exchange -------- */

/* The data-parallel step's collective behind the C ABI (ONE all-reduce of the flat fp32 gradient per
 * step, mean of the per-rank gradients; the reference itself is single-GPU, train_val.py:55-56, and has no counterpart).  RCCL is
 * bound at the first call (dlopen by soname: a process that already holds torch's copy keeps ONE RCCL); DIS_ERR_UNSUPPORTED when no
 * librccl can be found.  RCCL's own errors come back as 10000 + ncclResult_t.
 *   dis_allreduce_unique_id: rank 0 fills 128 bytes (ncclUniqueId) and hands them to the other ranks by its own means;
 *   dis_allreduce_init:      every rank, collectively; *comm is an opaque handle owned by the caller (dis_allreduce_destroy);
 *   dis_allreduce_sum_f32:   in place on `count` floats of device memory, asynchronous on `stream`; average != 0: the sum / nranks
 *                            (ncclAvg) - the DP mean, so that dis_adam_step runs with grad_scale 1; count 0 is a no-op.
 * One process per GPU, the device current at dis_allreduce_init is the communicator's. */
int dis_allreduce_unique_id(void* id128);
int dis_allreduce_init(void** comm, const void* id128, int nranks, int rank);
int dis_allreduce_sum_f32(void* comm, float* buf, long count, int average, void* stream);
int dis_allreduce_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif
