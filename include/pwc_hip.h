/*
 * pwc_hip.h -- C ABI of libpwc_hip.so: the MI355X (gfx950) PWC-Net inference ops.
 *
 * The reference (daigo0927/pwcnet) has no FFI layer: its op-level API is the Python
 * of model.py / modules.py running stock TensorFlow-1.8 primitives.  Each entry point
 * below replaces the TF sub-graph of one reference callable (file:line cited per
 * function); pwcnet_amd/modules.py binds them with ctypes and re-exposes the
 * reference's class names and call signatures.  INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions (all functions):
 *   - tensors are NHWC float32 in device (HBM) memory, caller-owned; nothing is
 *     allocated, freed or synchronised inside the library;
 *   - a tensor argument is (pointer to its first channel, channel stride `*_cs` in
 *     floats): pixel p, channel c lives at ptr[p * cs + c].  cs == C is a dense
 *     tensor; cs > C addresses a channel slice of a wider tensor, which is how
 *     tf.concat (modules.py:264,270,305) is made free;
 *   - work is enqueued on `stream` (a hipStream_t; NULL = the default stream) and the
 *     call returns immediately;
 *   - return value: PWC_OK (0), a negative PWC_E* argument error, or a positive
 *     hipError_t from the launch.  No exceptions, no global mutable state: the
 *     functions are re-entrant and thread-safe on distinct streams.  What a launch
 *     does is decided by its arguments alone (tile variants are explicit `_variant_`
 *     entry points); the library exports NO process-wide setting.  The ablation /
 *     tile-pinning knobs the A/B scripts under scripts/ use (pwc_debug_*) exist only
 *     in libpwc_hip_harness.so, a second build of the same sources with -DPWC_HARNESS
 *     (pwcnet_amd/_lib.py, PWC_HARNESS=1) that no product path and no test loads.
 */
#ifndef PWC_HIP_H
#define PWC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pwc_stream_t; /* hipStream_t */

enum {
    PWC_OK = 0,
    PWC_EINVAL = -1,   /* null pointer / non-positive size */
    PWC_EALIGN = -2,   /* pointer or channel stride not aligned as the kernel needs */
    PWC_ERANGE = -3,   /* size exceeds what the kernel indexes (see each function) */
    PWC_EUNSUPPORTED = -4
};

/* Library version (major*10000 + minor*100 + patch). */
int pwc_version(void);
/* Text for a return code of this library (negative codes) or of HIP (positive). */
const char* pwc_error_string(int code);

/* ---- a1: CostVolumeLayer.__call__ + get_cost/pad2d/crop2d, modules.py:158-204 ----
 * out[n,y,x,(v+R)*(2R+1)+(h+R)] =
 *     lrelu_slope( (1/C) * sum_c f0[n,y,x,c] * f1w[n,y+v,x+h,c] ),  0 outside the image;
 * v = vertical shift (outer loop, modules.py:197), h = horizontal (inner, :198).
 * Needs C % 4 == 0, f0/f1w 16-byte aligned with cs % 4 == 0; search_range in [1,4]. */
int pwc_cost_volume_f32(const float* f0, int f0_cs, const float* f1w, int f1w_cs,
                        float* out, int out_cs, int N, int H, int W, int C,
                        int search_range, float slope, pwc_stream_t stream);
/* 1 if pwc_cost_volume_f32 runs its rolling-window kernel for this geometry (C = 32, search_range 4,
 * 16-byte aligned operands, out_cs % 4 == 0, at least 4096 pixels per image), 0 for the tile kernel.
 * Same results either way; exported so that profilers can name the launch. */
int pwc_cost_volume_uses_rolling_kernel(int H, int W, int C, int search_range, int f0_cs, int f1_cs,
                                        int out_cs);

/* ---- a2: WarpingLayer(warp_type='bilinear') = bilinear_warp, modules.py:99-137 ----
 * out[n,y,x,:] = sum_{i,j} w_ij * x[n, clip(y+floor(fy)+i), clip(x+floor(fx)+j), :]
 * with (fx,fy) = flow_scale * flow[n,y,x,0:2]; flow_scale restates the caller's
 * `flows_up*self.scales[l]` (model.py:109).  Needs C % 4 == 0 and 16-byte alignment of
 * x/out with cs % 4 == 0. */
int pwc_warp_bilinear_f32(const float* x, int x_cs, const float* flow, int flow_cs,
                          float flow_scale, float* out, int out_cs,
                          int N, int H, int W, int C, pwc_stream_t stream);

/* ---- a3: WarpingLayer(warp_type='nearest') = nearest_warp, modules.py:83-97 ----
 * flow truncated toward zero (tf.cast int32, :85), added to the grid, clipped. */
int pwc_warp_nearest_f32(const float* x, int x_cs, const float* flow, int flow_cs,
                         float flow_scale, float* out, int out_cs,
                         int N, int H, int W, int C, pwc_stream_t stream);

/* ---- a2/a3 + the f0 part of tf.concat (modules.py:264) in one launch ----
 * pwc_warp_bilinear_f32 (bilinear != 0) or pwc_warp_nearest_f32 (bilinear == 0), and in the same
 * launch a copy of copy_C channels of every pixel of copy_src into copy_dst (copy_C == 0: none).
 * copy_C % 4 == 0, copy strides % 4 == 0, 16-byte aligned. */
int pwc_warp_copy_f32(int bilinear, const float* x, int x_cs, const float* flow, int flow_cs,
                      float flow_scale, float* out, int out_cs, int N, int H, int W, int C,
                      const float* copy_src, int copy_src_cs, float* copy_dst, int copy_dst_cs,
                      int copy_C, pwc_stream_t stream);

/* ---- a2+a1 (+ the f0 part of tf.concat, modules.py:264) for the COARSE pyramid levels ----
 * One launch for model.py:105-112 on small feature maps (7x16 ... 28x64 pixels per image), where
 * separate warp / cost-volume / copy launches are latency-bound: out as pwc_cost_volume_f32 of
 * (f0, warp(f1, flow_scale*flow)); flow == NULL: no warp (pyramid level 0, model.py:105-106);
 * f0_copy != NULL: f0's C channels are also copied to f0_copy (channel stride f0_copy_cs).
 * search_range must be 4 (PWC_EUNSUPPORTED otherwise); alignment rules of pwc_cost_volume_f32. */
int pwc_cost_volume_coarse_f32(const float* f0, int f0_cs, const float* f1, int f1_cs,
                               const float* flow, int flow_cs, float flow_scale,
                               float* out, int out_cs, float* f0_copy, int f0_copy_cs,
                               int N, int H, int W, int C, int search_range, float slope,
                               pwc_stream_t stream);

/* ---- a2+a1 fused: model.py:109-112 (warp then cost volume) without materialising
 * the warped feature map.  f1 is the UN-warped second feature map.  Same result as
 * pwc_warp_bilinear_f32 followed by pwc_cost_volume_f32.  Runs the matrix-pipe kernel of
 * pwc_warp_cost_volume_concat_f32 where that one is supported, the tile kernel with a gathering
 * loader otherwise. */
int pwc_warp_cost_volume_f32(const float* f0, int f0_cs, const float* f1, int f1_cs,
                             const float* flow, int flow_cs, float flow_scale,
                             float* out, int out_cs, int N, int H, int W, int C,
                             int search_range, float slope, pwc_stream_t stream);

/* ---- a2+a1 + the `features_0` part of tf.concat (modules.py:264) in ONE launch, correlation on the matrix
 * pipe (csrc/cost_volume_mfma.hip) -- model.py:105-112 for a pyramid level:
 *   out      = pwc_cost_volume_f32(f0, bilinear_warp(f1, flow_scale * flow))     (flow == NULL: f1 as is)
 *   f0_copy  = f0                                                                (f0_copy == NULL: no copy)
 * without ever writing the warped map.  out_pad_writable != 0 declares channels 81..83 of every `out` record
 * padding that the callee may overwrite with zeros (the estimator buffers of pwcnet_amd/model.py: segments start
 * on multiples of 4 channels) -- the record then leaves as 21 full 16-byte stores.
 * Supported: search_range 4, C in {32, 64, 96}, 16-byte aligned f0 / f1 / out / f0_copy with channel strides
 * % 4 == 0, flow 4-byte aligned, per-image extents below 2^31 bytes; anything else returns PWC_EUNSUPPORTED /
 * PWC_EALIGN and the caller uses the separate entry points.  pwc_warp_cost_volume_concat_supported tells in
 * advance (pointers taken as aligned). */
int pwc_warp_cost_volume_concat_f32(const float* f0, int f0_cs, const float* f1, int f1_cs,
                                    const float* flow, int flow_cs, float flow_scale,
                                    float* out, int out_cs, int out_pad_writable,
                                    float* f0_copy, int f0_copy_cs,
                                    int N, int H, int W, int C, int search_range, float slope,
                                    pwc_stream_t stream);
int pwc_warp_cost_volume_concat_supported(int H, int W, int C, int search_range, int f0_cs, int f1_cs,
                                          int flow_cs, int out_cs, int f0_copy_cs);
/* Round 5: the same launch with the correlation on the F16 matrix pipe: every operand (f0, and the warped f1 after the
 * fp32 bilinear blend of modules.py:132-135) is used as the two-term fp16 split x = h + 2^-11 m' of pwc_conv3x3_h2_f32,
 * three v_mfma_f32_16x16x32_f16 per (4x4-pixel block pair, 32 channels), fp32 accumulation -- 27 matrix instructions per
 * block where the fp32 form has 72 twice as long, and none of them stalls the vector instructions of its SIMD.  Error
 * against float64: not larger than the fp32 form's (tests).  RANGE: features below 65504 in magnitude (beyond: NaN
 * outputs, see PWC_STATUS_NONFINITE).  Same support set as the fp32 form.  Round 6: out_pad_writable = 2 (flow != NULL) puts
 * the pixel's flow (x, y), as read, into channels 81, 82 of its record and zero into channel 83: the record is then
 * [cv | flows_up_prev | 0] of the estimator's input (see pwc_conv3x3_h2_ex3_f32). */
int pwc_warp_cost_volume_concat_h2_f32(const float* f0, int f0_cs, const float* f1, int f1_cs,
                                       const float* flow, int flow_cs, float flow_scale,
                                       float* out, int out_cs, int out_pad_writable,
                                       float* f0_copy, int f0_copy_cs,
                                       int N, int H, int W, int C, int search_range, float slope,
                                       pwc_stream_t stream);
/* Round 5: the same operation for the SMALL pyramid levels -- C in {96, 128, 192} (7 x 16 ... 28 x 64 pixels at 448 x 1024) --
 * on the F16 matrix pipe (csrc/cost_volume_blk.hip): one 4 x 4-pixel block of f0 per workgroup, its 12 x 12-pixel window
 * of f1 (all channels, the four bilinear corners) requested at once -- or, where a launch has fewer blocks than the device has
 * CUs, one of the window's three block rows per workgroup -- so a launch is one chain flow -> corners -> blend -> matrix
 * instructions -> stores instead of a channel-stage loop of memory round trips (pwc_cost_volume_coarse_f32) or a walk over
 * block rows (the _h2 form: faster from about 8192 pixels per launch on, where this form's nine-fold re-gather of f1 costs
 * more than the walk's prologue).  Also takes C = 64 (small batches).  Arithmetic and range as
 * pwc_warp_cost_volume_concat_h2_f32.  Alignment requirements as pwc_warp_cost_volume_concat_f32;
 * pwc_warp_cost_volume_concat_blk_supported tells in advance. */
int pwc_warp_cost_volume_concat_blk_f32(const float* f0, int f0_cs, const float* f1, int f1_cs,
                                       const float* flow, int flow_cs, float flow_scale,
                                       float* out, int out_cs, int out_pad_writable,
                                       float* f0_copy, int f0_copy_cs,
                                       int N, int H, int W, int C, int search_range, float slope,
                                       pwc_stream_t stream);
int pwc_warp_cost_volume_concat_blk_supported(int H, int W, int C, int search_range, int f0_cs, int f1_cs,
                                              int flow_cs, int out_cs, int f0_copy_cs);

/* ---- a4/a5/a6: tf.layers.Conv2D(Cout,(3,3),(s,s),'same',dilation_rate=d) [+
 * tf.nn.leaky_relu(slope)] -- modules.py:62-67,267-268,274,306-324 ----
 *
 * Weight packing for the fp32-MFMA implicit-GEMM kernel.  `w_hwio` is the TF variable
 * (3,3,Cin,Cout).  `cin_map` (length Cin_phys, device int32, may be NULL = identity on
 * the first Cin entries, padding after) gives for every PHYSICAL input channel of the
 * activation buffer the logical TF channel it holds, or -1 for a padding channel
 * (its weights are packed as zero).  Cin_phys % 16 == 0.  The packed buffer holds
 * pwc_conv3x3_packed_floats(Cin_phys, Cout) floats. */
size_t pwc_conv3x3_packed_floats(int Cin_phys, int Cout);
int pwc_conv3x3_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys,
                         int Cout, float* packed, pwc_stream_t stream);

/* y[n,oy,ox,co] = act(bias[co] + sum_{ty,tx,ci} x[n,oy*s-pt+ty*d,ox*s-pl+tx*d,ci] * w[ty,tx,ci,co])
 * with TF 'SAME' padding (pad_total = max((out-1)*s + 2d+1 - in, 0), pad_before =
 * pad_total/2: stride 2 on an even size pads bottom/right only).  apply_act=0 gives
 * the activation-less flow heads.  Output is (N,ceil(H/s),ceil(W/s),Cout) at channel
 * stride y_cs.
 * Launch plan: `tile` = -1 lets the library choose (large M: big tiles for the full
 * rounds of workgroups + a smaller-tile tail launch; small M: the 9 taps are split over
 * 3 or 9 workgroup groups whose partial sums go through `workspace` and are summed in a
 * fixed order by a reduce kernel -- deterministic).  tile >= 0 forces one tile
 * configuration; `split` = 0 (library decides), 1 (never), 3 or 9.  `workspace` is
 * caller-owned scratch of `workspace_floats` floats (pwc_conv3x3_workspace_floats gives
 * the worst case); NULL disables the tap split.
 * Needs Cout % 16 == 0, x 16-byte aligned, x_cs % 4 == 0, x_cs >= Cin_phys, and every
 * padding channel of x finite (they meet zero weights). */
int pwc_conv3x3_f32(const float* x, int x_cs, const float* packed, const float* bias,
                    float* y, int y_cs, int N, int H, int W, int Cin_phys, int Cout,
                    int stride, int dilation, int apply_act, float slope, int tile, int split,
                    float* workspace, size_t workspace_floats, pwc_stream_t stream);
size_t pwc_conv3x3_workspace_floats(int M, int Cout);

/* Introspection: the plan pwc_conv3x3_f32(tile = -1) uses for M output pixels:
 * plan4 = {main tile id, tail tile id or -1, pixels covered by the main launch, tap split};
 * pwc_conv3x3_tile_shape gives a tile id's workgroup tile BM x BN. */
int pwc_conv3x3_plan(int M, int Cout, int Cin_phys, int* plan4);
/* 1 when pwc_conv3x3_f32(tile = -1, split = 0) routes this shape to the resident-weights /
 * halo-patch kernel (stride 1, dilation 1, Cin_phys == Cout in {16, 32}, M >= 65536). */
int pwc_conv3x3_uses_halo_kernel(int M, int Cin_phys, int Cout, int stride, int dilation);
int pwc_conv3x3_tile_shape(int tile, int* bm, int* bn);

/* Winograd F(2x2,3x3) form of the same convolution for stride 1 and Cout % 16 == 0:
 * 2.25x fewer multiplies (16 per 2x2 outputs instead of 36), fp32 result within ~1e-6
 * relative of the direct sum.  A dilation-d convolution is run as d*d ordinary ones on the
 * pixel sub-lattices (y mod d, x mod d).  pwc_conv3x3_wino_workgroups gives the number of
 * 16x16-pixel x 32-channel workgroups a launch would have (the caller's profitability test:
 * small or sparsely filled launches are faster on pwc_conv3x3_f32).  `packed_u` holds the transformed weights
 * U = G g G^T produced by pwc_conv3x3_wino_pack_f32 (same cin_map semantics as
 * pwc_conv3x3_pack_f32; pwc_conv3x3_wino_packed_floats floats).  Other requirements as
 * pwc_conv3x3_f32. */
size_t pwc_conv3x3_wino_packed_floats(int Cin_phys, int Cout);
int pwc_conv3x3_wino_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys,
                              int Cout, float* packed_u, pwc_stream_t stream);
int pwc_conv3x3_wino_f32(const float* x, int x_cs, const float* packed_u, const float* bias,
                         float* y, int y_cs, int N, int H, int W, int Cin_phys, int Cout,
                         int dilation, int apply_act, float slope, pwc_stream_t stream);
long pwc_conv3x3_wino_workgroups(int N, int H, int W, int Cout, int dilation);
/* Winograd F(4x4,3x3) form of the same convolution (csrc/conv3x3_wino4.hip): 36 multiplies per 4x4 outputs instead
 * of 64, for the big full-resolution layers (modules.py:267-268 `optflow_4/conv2d..conv2d_2`, modules.py:308-316
 * `context/conv2d_1..conv2d_3`).  fp32 throughout; its rounding error is ~10x that of F(2x2) per layer (2e-6 of the
 * activation scale) and leaves the end-to-end flow error where a direct fp32 convolution puts it (profiles/
 * r03_f4x4_numerics.txt).  packed_u comes from pwc_conv3x3_wino4_pack_f32 ([36][Cin_phys/16][Cout_pad][16] floats).
 * Needs Cout % 16 == 0, Cin_phys % 16 == 0, y 16-byte aligned with y_cs % 4 == 0.  pwc_conv3x3_wino4_supported: 1
 * where it is the faster kernel for the shape (48 <= Cin_phys <= 256, Cout >= 32, at least 256 workgroups of 16 x 32
 * pixels x 16 couts, sub-lattices of at least 14 x 28 pixels), 0 otherwise. */
size_t pwc_conv3x3_wino4_packed_floats(int Cin_phys, int Cout);
int pwc_conv3x3_wino4_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys,
                               int Cout, float* packed_u, pwc_stream_t stream);
int pwc_conv3x3_wino4_f32(const float* x, int x_cs, const float* packed_u, const float* bias,
                          float* y, int y_cs, int N, int H, int W, int Cin_phys, int Cout,
                          int dilation, int apply_act, float slope, pwc_stream_t stream);
int pwc_conv3x3_wino4_supported(int N, int H, int W, int Cin_phys, int Cout, int dilation);
/* The same convolution DIRECTLY (no Winograd transform) on the F16 matrix pipe with exact-to-22-bit operand splits
 * (csrc/conv3x3_h2.hip; reference modules.py:266-268 `optflow_l/conv2d*`, modules.py:306-323 `context/conv2d*`,
 * modules.py:221-236 `fp_extractor/conv2d*` where they are stride 1).  fp32 in, fp32 out, fp32 accumulation.  Every
 * operand is used as x = h + 2^-11 m' with h = fp16(x), m' = fp16((x - h) 2^11); of the four cross products the three
 * above 2^-22 are formed (uh vh, uh vm', um' vh; v_mfma_f32_32x32x16_f16) in two fp32 accumulators.  Measured error
 * against a float64 convolution: 0.1x that of pwc_conv3x3_wino4_f32 and 0.7x that of pwc_conv3x3_wino_f32 on every
 * layer shape (profiles/r04_exp_h2.txt), 0.4x that of a v_mfma_f32_16x16x4_f32 chain (profiles/
 * r04_exp_f16x2_numerics.txt).  RANGE: inputs and weights must be below 65504 in magnitude (fp16's largest finite
 * value); a larger input makes the outputs that depend on it NaN (inf - inf in the split), never a silently wrong
 * number.  LOWER END: the split is exact to 22 bits for |x| >= 2^-14 (6e-5); below that h and m' reach fp16's subnormals and
 * the operand carries an ABSOLUTE error of up to 1.5e-11 (3e-4 relative at |x| = 1e-7) -- harmless for activations and
 * weights that are summed with terms of ordinary size, not for a tensor that is small as a whole: the training path keeps its
 * data gradients on the fp32 kernels for that reason (pwcnet_amd/train.py, Trainer(f16x2_dgrad=False)).  packed_w comes from pwc_conv3x3_h2_pack_f32 (split weights, pwc_conv3x3_h2_packed_floats floats; same
 * cin_map semantics as pwc_conv3x3_pack_f32).  Needs Cout % 32 == 0, Cout <= 512, Cin_phys % 16 == 0, x and y 16-byte aligned with
 * x_cs % 4 == 0 and y_cs % 4 == 0.  pwc_conv3x3_h2_supported: 1 where it is the fastest kernel of this library for the
 * shape (Cin_phys >= 32, sub-lattices of at least 7 x 24 pixels -- narrower ones of an even dilation are taken two at a time
 * where Cout % 64 == 0 --, at least 192 tiles -- or at least 64 with a channel loop
 * long enough for the workspace form below to spread it over the CUs), 0 otherwise; the entry point
 * itself accepts every shape that meets the requirements above. */
size_t pwc_conv3x3_h2_packed_floats(int Cin_phys, int Cout);
int pwc_conv3x3_h2_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys,
                            int Cout, float* packed_w, pwc_stream_t stream);
int pwc_conv3x3_h2_f32(const float* x, int x_cs, const float* packed_w, const float* bias,
                       float* y, int y_cs, int N, int H, int W, int Cin_phys, int Cout,
                       int dilation, int apply_act, float slope, float* workspace, size_t workspace_floats,
                       pwc_stream_t stream);
/* `workspace` (caller-owned, 16-byte aligned, pwc_conv3x3_h2_workspace_floats floats, every byte 0xFF before the first
 * launch that uses the buffer -- every launch leaves it so -- and not shared by launches that may run concurrently)
 * turns a launch into one workgroup per CU, each computing an equal share of the launch's (tile, 16-channel stage)
 * sequence, wherever that is the faster form (more tiles than CUs: no partly filled last round, one prologue per workgroup
 * instead of one per tile; far fewer tiles than CUs and a long channel loop: every CU gets a part of it).  A tile cut into
 * pieces is finished by the workgroup holding its first piece, which adds the others' published fp32 sums to its own in the
 * order of their stages (finished sums, fixed order: the result does not depend on timing, launches repeat bitwise) --
 * within fp32 rounding of the uncut sum.  workspace = NULL (or pwc_conv3x3_h2_workspace_floats = 0): one workgroup per
 * tile. */
size_t pwc_conv3x3_h2_workspace_floats(int N, int H, int W, int Cin_phys, int Cout, int dilation);
int pwc_conv3x3_h2_supported(int N, int H, int W, int Cin_phys, int Cout, int dilation);
/* Round 5.  Status words: `status` points at TWO caller-owned uint32 in device memory (8-byte aligned; the caller zeroes
 * them, launches only OR bits into status[0] / atomic-max status[1]; NULL: no report).
 *   status[0] & PWC_STATUS_NONFINITE        pwc_resize_bilinear_status_f32 wrote a value that is not finite.  The F16-pipe
 *                                           kernels (pwc_conv3x3_h2*, pwc_conv3x3_c16pair*, pwc_warp_cost_volume_concat_h2_f32)
 *                                           turn an operand at or beyond 65504 in magnitude into NaN outputs (inf - inf in the
 *                                           split), NaN survives every later layer of the network, so the model's LAST launch
 *                                           sees it: repeat the forward on the fp32 kernels (the reference, plain fp32, has no
 *                                           such limit).  Watching the operands inside the kernels instead was measured at 2-3 %
 *                                           of every launch (profiles/r05_timeline_range_tracking_cost.txt).
 *   status[0] & PWC_STATUS_STREAMK_TIMEOUT  a workgroup of the workspace form of pwc_conv3x3_h2* gave up waiting for a published
 *                                           partial sum (cannot happen short of a fault); its outputs are NaN and the workspace
 *                                           may hold stale sums: refill it with 0xFF bytes before its next use
 *   status[1]                               pwc_absmax_f32: bits of the largest |value| seen (a non-negative float orders like
 *                                           its bit pattern) -- the debugging aid behind PWCDCNet(track_max=True)
 * pwc_conv3x3_h2_ex_f32 = pwc_conv3x3_h2_f32 with the status words and with the input channels given as TWO tensors over
 * the same pixels: physical channels [0, Cin_a_phys) are channels [0, Cin_a_phys) of x (Cin_a_phys % 16 == 0), channels
 * [Cin_a_phys, Cin_phys) are channels [0, Cin_phys - Cin_a_phys) of x2 (channel stride x2_cs, 16-byte aligned).  This is
 * how `tf.concat([cv, features_0, flows_up_prev, features_up_prev])` (reference modules.py:261-264) costs nothing for
 * features_0: the estimator's first layer reads it from the pyramid tensor.  x2 = NULL: all channels from x. */
#define PWC_STATUS_NONFINITE 1u
#define PWC_STATUS_STREAMK_TIMEOUT 2u
int pwc_conv3x3_h2_ex_f32(const float* x, int x_cs, int Cin_a_phys, const float* x2, int x2_cs,
                          const float* packed_w, const float* bias, float* y, int y_cs, int N, int H, int W,
                          int Cin_phys, int Cout, int dilation, int apply_act, float slope, float* workspace,
                          size_t workspace_floats, uint32_t* status, pwc_stream_t stream);
/* Round 6: the input channels as THREE tensors over the same pixels (dilation 1): physical channels [0, Cin_a_phys) from x, the
 * next Cin_b_phys from x2, the remaining Cin_phys - Cin_a_phys - Cin_b_phys from x3 (Cin_a_phys, Cin_b_phys % 16 == 0; x2, x3
 * 16-byte aligned, strides % 4 == 0).  An operand's channel stride may be up to 12 channels SHORT of its 16-channel stage count
 * (x_cs = 84 with Cin_a_phys = 96): its last stage then reads the first channels of the NEXT pixel's record (zeros behind an
 * image's last pixel), which must hold finite values, and the caller packs zero weights for those channels (cin_map = -1).
 * With it `tf.concat([cv, features_0, flows_up_prev, features_up_prev])` (reference modules.py:261-264) is three tensors that
 * their producers write DENSELY: [cv 81 | flows_up_prev 2 | 0] in 336-byte records (pwc_warp_cost_volume_concat_h2_f32 with
 * out_pad_writable = 2), features_0 in the pyramid tensor, features_up_prev as a 32-channel tensor -- no producer writes a slice
 * of a wider record.  The logical channel order of the concat lives in the packed weights. */
int pwc_conv3x3_h2_ex3_f32(const float* x, int x_cs, int Cin_a_phys, const float* x2, int x2_cs, int Cin_b_phys,
                           const float* x3, int x3_cs, const float* packed_w, const float* bias, float* y, int y_cs,
                           int N, int H, int W, int Cin_phys, int Cout, int apply_act, float slope, float* workspace,
                           size_t workspace_floats, uint32_t* status, pwc_stream_t stream);
/* TWO chained 3x3 stride-1 'SAME' convolutions of 16 channels each (16 -> 16 -> 16) with a leaky-relu of slope `slope`
 * behind each, in one launch (csrc/conv3x3_c16pair.hip; reference modules.py:62-67, the `fp_extractor/conv2d_1`,
 * `conv2d_2` pair of pyramid level 1): the intermediate stays in LDS instead of making a round trip through memory.
 * Arithmetic as pwc_conv3x3_h2_f32 (fp32 in / out / accumulation, exact-to-22-bit fp16 operand pairs, three products;
 * the intermediate is rounded to fp32 exactly as a tensor in memory would be; inputs below 65504).  packed comes from
 * pwc_conv3x3_c16pair_pack_f32 (both HWIO (3,3,16,16) kernels; pwc_conv3x3_c16pair_packed_floats floats).  x, y: NHWC
 * with channel strides x_cs, y_cs >= 16 (multiples of 4), 16-byte aligned.  _supported: 1 where the fused launch is
 * the faster way to run the pair (at least one 16 x 32-pixel tile per CU). */
size_t pwc_conv3x3_c16pair_packed_floats(void);
int pwc_conv3x3_c16pair_pack_f32(const float* w1_hwio, const float* w2_hwio, float* packed, pwc_stream_t stream);
int pwc_conv3x3_c16pair_f32(const float* x, int x_cs, const float* packed, const float* bias1, const float* bias2,
                            float* y, int y_cs, int N, int H, int W, float slope, pwc_stream_t stream);
int pwc_conv3x3_c16pair_supported(int N, int H, int W);
/* The same launch with the stride-2 'SAME' convolution of the RAW images in front (3 -> 16 -> 16 -> 16, leaky-relu behind
 * each; reference modules.py:57-67, `fp_extractor/conv2d`, `conv2d_1`, `conv2d_2`: all of pyramid level 1): a workgroup
 * fetches the 41 x 73-pixel raw patch of its 16 x 32-pixel output tile and neither intermediate leaves the CU.  The first
 * convolution runs on the same split arithmetic (27 of a 32-deep matrix instruction).  The images are two batches that
 * share the weights (the two frames of a pair): N_a images at x_a, N_b at x_b (x_b = NULL with N_b = 0), NHWC with
 * EXACTLY 3 channels at channel stride 3, H0 x W0 with W0 % 4 == 0 (PWC_EUNSUPPORTED otherwise), 16-byte aligned; y:
 * (N_a + N_b) x ceil(H0 / 2) x W0 / 2 x 16 at channel stride y_cs.  packed: pwc_conv3x3_c3c16pair_pack_f32 of the HWIO
 * (3,3,3,16), (3,3,16,16), (3,3,16,16) kernels (pwc_conv3x3_c3c16pair_packed_floats floats).  _supported: W0 % 4 == 0
 * and the level-1 image is one pwc_conv3x3_c16pair_supported names. */
size_t pwc_conv3x3_c3c16pair_packed_floats(void);
int pwc_conv3x3_c3c16pair_pack_f32(const float* w0_hwio, const float* w1_hwio, const float* w2_hwio, float* packed,
                                   pwc_stream_t stream);
int pwc_conv3x3_c3c16pair_f32(const float* x_a, int N_a, const float* x_b, int N_b, const float* packed,
                              const float* bias0, const float* bias1, const float* bias2, float* y, int y_cs,
                              int H0, int W0, float slope, pwc_stream_t stream);
int pwc_conv3x3_c3c16pair_supported(int N, int H0, int W0);
/* Stride 2 ('SAME', dilation 1, EVEN H and W; the extractor's down-sampling layers, reference modules.py:57-60) through the
 * same kernel (round 5).  With the input seen as its four parity planes x_ab[y', x'] = x[2 y' + a, 2 x' + b] the strided
 * convolution is a stride-1 one with taps at 0 / +1 on each plane: a channel stage of the kernel is (16 channels, parity), fetched
 * from the pixels of its plane, and four of a stage's nine tap slots carry matrix instructions -- 16 Cin products per output
 * (a strided convolution needs 9 Cin; round 4's "stride-1 launch storing every second sum" executed 36 Cin and lost to the fp32
 * kernel).  y is (N, H / 2, W / 2) at channel stride y_cs.  packed_w: pwc_conv3x3_h2_stride2_pack_f32 (same arguments as
 * pwc_conv3x3_h2_pack_f32: cin_map over the input's Cin_phys physical channels; pwc_conv3x3_h2_stride2_packed_floats floats);
 * workspace: pwc_conv3x3_h2_stride2_workspace_floats (0: none), rules of pwc_conv3x3_h2_f32; status: the caller's status words
 * (or NULL), as pwc_conv3x3_h2_ex_f32's (PWC_STATUS_STREAMK_TIMEOUT).  Odd sizes: PWC_EUNSUPPORTED
 * (pwc_conv3x3_f32 takes them).  _supported: 1 where it is the faster kernel: inputs of up to 32 channels whose output is a
 * shape pwc_conv3x3_h2_supported takes at 4 Cin_phys channels (measured: 16 -> 32 and 32 -> 64 of the extractor; from 64 input
 * channels on the per-stage fixed cost of 4 x as many stages outweighs the matrix pipe's rate). */
size_t pwc_conv3x3_h2_stride2_packed_floats(int Cin_phys, int Cout);
int pwc_conv3x3_h2_stride2_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys,
                                    int Cout, float* packed_w, pwc_stream_t stream);
size_t pwc_conv3x3_h2_stride2_workspace_floats(int N, int H, int W, int Cin_phys, int Cout);
int pwc_conv3x3_h2_stride2_f32(const float* x, int x_cs, const float* packed_w, const float* bias,
                               float* y, int y_cs, int N, int H, int W, int Cin_phys, int Cout,
                               int apply_act, float slope, float* workspace, size_t workspace_floats,
                               uint32_t* status, pwc_stream_t stream);
int pwc_conv3x3_h2_stride2_supported(int N, int H, int W, int Cin_phys, int Cout);

/* Round 5: 3x3 convolution for SMALL launches (csrc/conv3x3_sk.hip) -- the 7 x 16 and 14 x 32 pyramid levels of a batch of
 * 8, every level of a single pair -- on the F16 matrix pipe, arithmetic and RANGE of pwc_conv3x3_h2_f32 (two-term fp16
 * operand splits, fp32 accumulation; |x|, |w| < 65504).  Same operation as pwc_conv3x3_f32 (TF 'SAME' padding, stride 1 | 2,
 * any dilation, + bias, + leaky_relu when apply_act): the K dimension (9 taps x Cin_phys) of a 16-pixel x 16-channel tile is
 * dealt to the eight waves of ONE workgroup, every wave requests all its operands at once, the eight partial sums are added
 * in wave order (launches repeat bitwise) -- one dispatch where the tiled kernels need a tap / channel split over workgroups
 * plus a reduce launch to fill the device.  Needs Cin_phys % 32 == 0, Cout % 16 == 0, x / y / packed_w / bias 16-byte
 * aligned, x_cs % 4 == 0, y_cs % 4 == 0, N*H*W*x_cs*4 < 2^31.  packed_w: pwc_conv3x3_sk_pack_f32 (the split halves in
 * fragment order, pwc_conv3x3_sk_packed_floats floats; cin_map as in pwc_conv3x3_pack_f32).  pwc_conv3x3_sk_supported: 1
 * where it is the fastest kernel of this library for the shape (the form that stages the workgroup's input patch in the
 * LDS: stride 1, no dilation, 96 ... 288 input channels, up to 2.4e8 multiply-adds -- output pixels x Cin_phys x Cout -- and 8 K
 * output pixels; stride 2, 64 ... 128 input channels, up to 2e8 multiply-adds; otherwise up to 1e8 multiply-adds and 4096
 * output pixels, beyond that pixel count stride-2 and thin layers only), else 0.
 * pwc_conv3x3_sk_variant_f32: the same convolution with the workgroup tile and form GIVEN -- tile in {11, 21, 22} (fragments from
 * global memory), {31, 41, 42} (patch in the LDS) -- for tests and tuning (every tile gives the same result on every shape it
 * admits; PWC_EUNSUPPORTED where it does not: x2 tiles need Cout % 32 == 0, 3x / 4x no dilation and at most 288 (stride 2: 128)
 * input channels).  The tile is an argument of the call: the library keeps no process-wide setting. */
size_t pwc_conv3x3_sk_packed_floats(int Cin_phys, int Cout);
int pwc_conv3x3_sk_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys, int Cout,
                            float* packed_w, pwc_stream_t stream);
int pwc_conv3x3_sk_f32(const float* x, int x_cs, const float* packed_w, const float* bias, float* y, int y_cs,
                       int N, int H, int W, int Cin_phys, int Cout, int stride, int dilation, int apply_act,
                       float slope, pwc_stream_t stream);
int pwc_conv3x3_sk_supported(int N, int H, int W, int Cin_phys, int Cout, int stride, int dilation);
int pwc_conv3x3_sk_variant_f32(const float* x, int x_cs, const float* packed_w, const float* bias, float* y, int y_cs,
                               int N, int H, int W, int Cin_phys, int Cout, int stride, int dilation, int apply_act,
                               float slope, int tile, pwc_stream_t stream);

/* Round 5: the same convolution for THIN inputs to 32 output channels (csrc/conv3x3_t32.hip) -- Cin_phys 16 (stride 1 | 2) or
 * 32 (stride 1), Cout 32, no dilation: the feature extractor's full-resolution layers (reference modules.py:58-71,
 * fp_extractor/conv2d_3 ... conv2d_5).  Weights STATIONARY: a wave keeps the split halves of the whole weight tensor in
 * registers (9 or 18 K steps of a 32 x 32 x 16 matrix instruction whose row operand is all 32 output channels) and only pixel
 * fragments move: a workgroup walks over tiles of 8 rows x 32 pixels, the next tile's input patch arrives by LDS-DMA while the
 * four waves compute the current one; one barrier per tile.  Arithmetic, RANGE and alignment requirements of
 * pwc_conv3x3_sk_f32.  packed_w: pwc_conv3x3_t32_pack_f32 (pwc_conv3x3_t32_packed_floats floats; cin_map as in
 * pwc_conv3x3_pack_f32).  pwc_conv3x3_t32_supported: 1 where it is the fastest kernel of this library for the shape (a
 * supported shape with at least 256 tiles), else 0. */
size_t pwc_conv3x3_t32_packed_floats(int Cin_phys);
int pwc_conv3x3_t32_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys, float* packed_w,
                             pwc_stream_t stream);
int pwc_conv3x3_t32_f32(const float* x, int x_cs, const float* packed_w, const float* bias, float* y, int y_cs,
                        int N, int H, int W, int Cin_phys, int Cout, int stride, int apply_act, float slope,
                        pwc_stream_t stream);
int pwc_conv3x3_t32_supported(int N, int H, int W, int Cin_phys, int Cout, int stride);
/* Round 6: the same convolution for 32 OUTPUT channels from 32 or 64 input channels, stride 1, no dilation (csrc/conv3x3_w32.hip)
 * -- fp_extractor/conv2d_4, conv2d_5 (reference modules.py:58-71), optflow_l/conv2d_4 (modules.py:266-268: the 64 -> 32 layer of
 * every estimator) and context/conv2d_5 (modules.py:321-322).  The whole split weight tensor stays in the LDS for the life of a
 * persistent workgroup (36 / 72 KB), all 32 output channels are the row operand of one 32 x 32 x 16 matrix instruction, the input
 * patch of a tile (8 / 4 rows x 32 columns) is requested into registers under the previous tile's K loop and split once: no
 * per-stage staging (pwc_conv3x3_h2_f32 runs these layers at half the rate of its 128-cout layers).  64 input channels: the K
 * loop is dealt to two wave groups whose finished sums are added in a fixed order (launches repeat bitwise).  Arithmetic, RANGE
 * and alignment requirements of pwc_conv3x3_sk_f32; N*H*W*x_cs*4 and N*H*W*y_cs*4 < 2^31.  packed_w: pwc_conv3x3_w32_pack_f32
 * (pwc_conv3x3_w32_packed_floats floats; cin_map as in pwc_conv3x3_pack_f32).  pwc_conv3x3_w32_supported: 1 where it is the
 * fastest kernel of this library for the shape (a supported shape with at least 256 tiles), else 0. */
size_t pwc_conv3x3_w32_packed_floats(int Cin_phys);
int pwc_conv3x3_w32_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys, float* packed_w,
                             pwc_stream_t stream);
int pwc_conv3x3_w32_f32(const float* x, int x_cs, const float* packed_w, const float* bias, float* y, int y_cs,
                        int N, int H, int W, int Cin_phys, int Cout, int apply_act, float slope, pwc_stream_t stream);
int pwc_conv3x3_w32_supported(int N, int H, int W, int Cin_phys, int Cout, int stride, int dilation);
/* Tile variants of the kernel above (workgroup = couts x rows x 32 columns): 1 = 128 x 8, 2 = 64 x 16, 3 = 96 x 8,
 * 4 = 32 x 16, 5 = 64 x 8.  pwc_conv3x3_h2_plan: the one pwc_conv3x3_h2_f32 launches for a shape (fewest estimated
 * rounds of 256 workgroups x matrix instructions per tap; 0 = the shape is not accepted).  pwc_conv3x3_h2_variant_f32:
 * the same convolution with the variant given (Cout must be a multiple of its couts) -- for tests and tuning. */
int pwc_conv3x3_h2_plan(int N, int H, int W, int Cin_phys, int Cout, int dilation);
int pwc_conv3x3_h2_variant_f32(const float* x, int x_cs, const float* packed_w, const float* bias,
                               float* y, int y_cs, int N, int H, int W, int Cin_phys, int Cout,
                               int dilation, int apply_act, float slope, int variant, float* workspace,
                               size_t workspace_floats, pwc_stream_t stream);
/* The same convolution with the input-channel stages dealt to `csplit` workgroups per tile (launches that would leave
 * most of the GPU's workgroup slots empty: the 14x32 / 28x64 pyramid levels).  Partial outputs go to `workspace`
 * (pwc_conv3x3_wino_split_workspace_floats floats, 16-byte aligned) and are summed in a fixed order, with the bias and the
 * activation, by a second kernel: deterministic, within fp32 rounding of pwc_conv3x3_wino_f32.
 * pwc_conv3x3_wino_split_plan gives the csplit to use for a shape (1 = do not split; then no workspace is needed). */
int pwc_conv3x3_wino_split_plan(int N, int H, int W, int Cin_phys, int Cout, int dilation);
size_t pwc_conv3x3_wino_split_workspace_floats(int N, int H, int W, int Cout, int csplit);
int pwc_conv3x3_wino_split_f32(const float* x, int x_cs, const float* packed_u, const float* bias,
                               float* y, int y_cs, int N, int H, int W, int Cin_phys, int Cout,
                               int dilation, int apply_act, float slope, int csplit, float* workspace,
                               size_t workspace_floats, pwc_stream_t stream);

/* Same convolution straight from the HWIO variable, any Cin/Cout, plus the optional
 * residual add of modules.py:275-277 (`flows += flows_up_prev`) and modules.py:326
 * (`flows + x`): y = act(conv) + residual.  Used for Cin = 3 (first extractor layer),
 * Cout = 2 (flow heads) and as the in-library cross-check of the MFMA kernel. */
int pwc_conv3x3_direct_f32(const float* x, int x_cs, const float* w_hwio, const float* bias,
                           float* y, int y_cs, const float* residual, int res_cs,
                           int N, int H, int W, int Cin, int Cout, int stride, int dilation,
                           int apply_act, float slope, pwc_stream_t stream);

/* ---- a7: tf.image.resize_bilinear (TF 1.8 legacy: src = dst*in/out, no half-pixel,
 * no align-corners), modules.py:283-284 and model.py:127; y = resize(x) * mul. */
int pwc_resize_bilinear_f32(const float* x, int x_cs, float* y, int y_cs,
                            int N, int H, int W, int C, int OH, int OW, float mul,
                            pwc_stream_t stream);
/* The same launch; additionally status[0] |= PWC_STATUS_NONFINITE if any value it writes is not finite (status: two
 * caller-owned uint32, see pwc_conv3x3_h2_ex_f32; NULL = pwc_resize_bilinear_f32).  The model's last launch -- the x4
 * upsampling of model.py:127 -- runs through this entry: one compare per output of a launch that waits for memory. */
int pwc_resize_bilinear_status_f32(const float* x, int x_cs, float* y, int y_cs,
                                   int N, int H, int W, int C, int OH, int OW, float mul,
                                   uint32_t* status, pwc_stream_t stream);
/* status[1] = max(status[1], bits of max |x[p, 0:C]|) over npix pixels (channel stride x_cs); NaN entries are skipped.
 * Debugging aid (PWCDCNet(track_max=True) runs it on every conv input): how far a network's activations are from the
 * 65504 of the F16-pipe kernels. */
int pwc_absmax_f32(const float* x, int x_cs, long npix, int C, uint32_t* status, pwc_stream_t stream);

/* Two resizes of one geometry in one launch (modules.py:283-284: flows_up and features_up of the
 * same level): xa/ya carry 2 channels (8-byte aligned, even channel strides), xb/yb CB channels
 * (CB % 4 == 0, 16-byte aligned).  Same arithmetic as pwc_resize_bilinear_f32 with mul = 1. */
int pwc_resize_bilinear_pair_f32(const float* xa, int xa_cs, float* ya, int ya_cs,
                                 const float* xb, int xb_cs, float* yb, int yb_cs,
                                 int N, int H, int W, int CB, int OH, int OW, pwc_stream_t stream);

/* ---- stream placement probe (no reference counterpart: the reference times repeated sess.run calls, test.py:48-53, and
 * must not depend on the process's stream history).  pwc_device_spin keeps `stream` busy for about `ticks` cycles of the
 * device clock counter without touching memory; pwc_device_touch adds 1 to p[0].  pwcnet_amd/model.py brackets the pair
 * with timing events to learn whether two HIP streams share a hardware queue; never part of a forward. */
int pwc_device_spin(long long ticks, pwc_stream_t stream);
int pwc_device_touch(float* p, pwc_stream_t stream);

/* ---- tf.concat helper (modules.py:264,305): dst[p, 0:C] = src[p, 0:C] for npix pixels. */
int pwc_copy_channels_f32(const float* src, int src_cs, float* dst, int dst_cs,
                          long npix, int C, pwc_stream_t stream);

/* ---- losses (forward): reference losses.py:4-13 (L1loss, L2loss, EPE) and the per-level term of
 * multiscale_loss / multirobust_loss (losses.py:15-48) ----
 * out_sums[n] = sum over the H x W pixels of image n of
 *     || pred[n,y,x,0:2] - gt[n, floor(y*GH/H), floor(x*GW/W), 0:2] / gt_div ||_ord ,  ord = 1 or 2;
 * the nearest-neighbour downsampling of the ground truth (tf.image.resize_nearest_neighbor,
 * losses.py:27,43) is folded into the read (GH = H, GW = W for none).  Deterministic (fixed-order
 * partial sums in `workspace`, at least pwc_flow_norm_workspace_floats(N,H,W) floats). */
size_t pwc_flow_norm_workspace_floats(int N, int H, int W);
int pwc_flow_norm_sums_f32(const float* pred, int pred_cs, const float* gt, int gt_cs,
                           int N, int H, int W, int GH, int GW, float gt_div, int ord,
                           float* workspace, size_t workspace_floats, float* out_sums,
                           pwc_stream_t stream);

/* ==== f4: training path (reference train.py:66-92: tf.gradients of the forward + tf.train.AdamOptimizer) ====
 * Gradient tensors have the layout of the activation they belong to.  `accumulate` != 0: the kernel adds
 * into its output (an activation with several consumers collects its gradient from all of them). */

/* tf.nn.leaky_relu gradient, in place: dy[p,c] *= (y[p,c] > 0 ? 1 : slope), y = the activation's OUTPUT. */
int pwc_lrelu_grad_f32(const float* y, int y_cs, float* dy, int dy_cs, long npix, int C, float slope,
                       pwc_stream_t stream);
/* dst[p,0:C] = (accumulate ? dst : 0) + alpha * src[p,0:C] (channel-slice add / scaled copy). */
int pwc_add_f32(const float* src, int src_cs, float* dst, int dst_cs, long npix, int C, float alpha,
                int accumulate, pwc_stream_t stream);
/* out[c] (+)= sum over pixels of dy[p,c]: bias gradient of tf.layers.Conv2D.  Deterministic. */
size_t pwc_channel_sums_workspace_floats(long npix, int C);
int pwc_channel_sums_f32(const float* dy, int dy_cs, long npix, int C, float* workspace,
                         size_t workspace_floats, float* out, int accumulate, pwc_stream_t stream);
/* The two above in ONE pass over dy: dy *= (y > 0 ? 1 : slope) in place, out[c] (+)= sum_p dy[p,c]. */
int pwc_lrelu_grad_channel_sums_f32(const float* y, int y_cs, float* dy, int dy_cs, long npix, int C, float slope,
                                    float* workspace, size_t workspace_floats, float* out, int accumulate,
                                    pwc_stream_t stream);
/* Transpose of pwc_resize_bilinear_f32 (tf.image.resize_bilinear legacy, modules.py:283-284) for integer
 * factors OH/H = OW/W in 1..4: dx (+)= mul * R^T dy.  Gather form, deterministic. */
int pwc_resize_bilinear_grad_f32(const float* dy, int dy_cs, float* dx, int dx_cs, int N, int H, int W, int C,
                                 int OH, int OW, float mul, int accumulate, pwc_stream_t stream);
/* Gradient of pwc_warp_bilinear_f32 (bilinear_warp, modules.py:99-137; floor / clip carry no gradient):
 * dx += scatter of the corner weights (fp32 atomics; dx may be null), dflow (+)= flow_scale * d/d(flow*scale). */
int pwc_warp_bilinear_grad_f32(const float* x, int x_cs, const float* flow, int flow_cs, float flow_scale,
                               const float* dy, int dy_cs, float* dx, int dx_cs, float* dflow, int dflow_cs,
                               int dflow_accumulate, int N, int H, int W, int C, pwc_stream_t stream);
/* The same gradient with a bit-reproducible scatter: the corner contributions to dx are added as 64-bit fixed-point
 * integers (2^-36 steps) in `workspace` (pwc_warp_bilinear_grad_workspace_bytes bytes, 8-byte aligned; zeroed by the
 * call) and converted once -- integer sums do not depend on the order of the atomics.  dx == NULL: no workspace needed.
 * Range: contributions below 1.5e-11 vanish, sums are exact up to +-1.3e8; an upstream gradient that is not finite or
 * reaches 2^26 raises a poison word behind the sums and EVERY element of dx becomes NaN (a diverging step stays visible;
 * the fp32-atomic form keeps relative precision instead). */
size_t pwc_warp_bilinear_grad_workspace_bytes(int N, int H, int W, int C);
int pwc_warp_bilinear_grad_det_f32(const float* x, int x_cs, const float* flow, int flow_cs, float flow_scale,
                                   const float* dy, int dy_cs, float* dx, int dx_cs, float* dflow, int dflow_cs,
                                   int dflow_accumulate, int N, int H, int W, int C, void* workspace,
                                   size_t workspace_bytes, pwc_stream_t stream);
/* Gradient of pwc_cost_volume_f32 (modules.py:158-204) w.r.t. both feature maps, leaky-relu and mean
 * included: cv = the forward output, dcv = its gradient.  df0 / df1w may be null.  search_range 4. */
int pwc_cost_volume_grad_f32(const float* f0, int f0_cs, const float* f1w, int f1w_cs, const float* cv, int cv_cs,
                             const float* dcv, int dcv_cs, float* df0, int df0_cs, float* df1w, int df1w_cs,
                             int accumulate, int N, int H, int W, int C, int search_range, float slope,
                             pwc_stream_t stream);
/* Gradient w.r.t. pred of  scale * sum_p || pred[p] - gt[nearest(p)] / gt_div ||_ord  (losses.py:4-8,20-29). */
int pwc_flow_norm_grad_f32(const float* pred, int pred_cs, const float* gt, int gt_cs, int N, int H, int W,
                           int GH, int GW, float gt_div, int ord, float scale, float* dpred, int dpred_cs,
                           int accumulate, pwc_stream_t stream);
/* tf.train.AdamOptimizer update of a flat parameter buffer (TF 1.8 adam.py; train.py:90) with the gradient
 * of gamma * l2_loss(var) folded in (train.py:75): g = grad_scale * grads + l2_gamma * p; lr_t from the host. */
int pwc_adam_step_f32(float* params, const float* grads, float* m, float* v, long n, float lr_t, float beta1,
                      float beta2, float eps, float l2_gamma, float grad_scale, pwc_stream_t stream);
/* Weight gradient of tf.layers.Conv2D(...,(3,3),(s,s),'same',dilation_rate=d) on the fp32 MFMA units:
 * dw_hwio[ty][tx][ci][co] (+)= sum_pixels x_pad[...] * dy, ci in the variable's LOGICAL order (cin_map maps the
 * physical channels of x, -1 = padding; null = identity).  Deterministic (fixed-order split-k sums). */
size_t pwc_conv3x3_wgrad_workspace_floats(int N, int H, int W, int Cin_phys, int Cout, int stride);
int pwc_conv3x3_wgrad_f32(const float* x, int x_cs, const float* dy, int dy_cs, const int32_t* cin_map, int Cin,
                          int Cin_phys, int Cout, float* dw_hwio, int accumulate, int N, int H, int W, int stride,
                          int dilation, float* workspace, size_t workspace_floats, pwc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PWC_HIP_H */
