/*
 * dyolo.h — C-ABI of libdyolo.so, the MI355X (gfx950) device library for the
 * Drone-YOLO detection hot path.
 *
 * The reference (ultralytics fork, /root/reference) has no FFI boundary of its
 * own: its seam is Python nn.Modules looked up by name (nn/tasks.py:1012-1018)
 * that call torch ATen ops.  Every entry point below replaces the torch call
 * sites named in its comment; the Python host mirror (drone-yolo_amd/nn/...,
 * utils/ops.py) binds them with ctypes exactly as INTEGRATION.md shows.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types, no C++ types.
 *   - all pointers are DEVICE pointers unless a comment says "host".
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*),
 *     re-entrant, never allocates or frees, never synchronises.
 *   - return value: DY_OK (0) or a negative dy_status; the message for the
 *     last failure on the calling thread is dy_last_error_string().
 *   - activations are NHWC ("channels last"): element (n,h,w,c) of a view lives
 *     at base + ((n*H + h)*W + w)*ld + c, where ld >= C is the pixel pitch in
 *     ELEMENTS.  A channel slice of a wider buffer is therefore just another
 *     (base, ld) pair: Concat / chunk are done by construction, not by copies
 *     (reference: nn/modules/conv.py:323-333, block.py:237-242).
 *   - dtype of activations/weights: DY_BF16, DY_F16 or DY_F32.  Accumulation,
 *     bias, SiLU and residual adds are always fp32.
 */
#ifndef DYOLO_H_
#define DYOLO_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DYOLO_VERSION_MAJOR 0
#define DYOLO_VERSION_MINOR 1

typedef void* dy_stream_t; /* hipStream_t */

typedef enum dy_status {
  DY_OK = 0,
  DY_ERR_INVALID_ARG = -1,   /* null pointer, bad size, misaligned view */
  DY_ERR_UNSUPPORTED = -2,   /* shape / dtype combination not built */
  DY_ERR_LAUNCH = -3,        /* hipLaunchKernel / HIP runtime error */
  DY_ERR_WORKSPACE = -4      /* workspace too small */
} dy_status;

/* DY_FP8 = OCP e4m3fn (gfx950's fp8, NOT MI300's fnuz): 1 byte per element, 16 per 16-byte chunk.  An fp8 activation or weight
 * element q stands for the real value q * scale (dy_conv_desc.act_scale for activations, w_scale[co] / act_scale for the weights of
 * output channel co).  Entry points that are not built for it return DY_ERR_INVALID_ARG ("bad dtype"). */
typedef enum dy_dtype { DY_BF16 = 0, DY_F16 = 1, DY_F32 = 2, DY_FP8 = 3, DY_F16X2 = 4 } dy_dtype;
/* DY_F16X2 = SPLIT float16 storage (round 5): the bar-exact precision at 16-bit MFMA speed.  An element x is the pair
 *   hi = rn_f16(x) (0 when |x| < 2^-14),  lo = rn_f16((x - hi) * 2^11),      x ~= hi + lo * 2^-11   (22 mantissa bits)
 * and occupies 4 bytes like an fp32 element; within a pixel row every group of 8 channels is 32 bytes, the 8 hi halves then the 8 lo
 * halves ([hi c..c+7 | lo c..c+7]): a channel slice at a multiple of 8 channels is a contiguous byte range (Concat / chunk by view as
 * for the other types, pitches counted in 4-byte elements), and a 128-byte K-step of the implicit GEMM holds four (hi, lo) chunk
 * pairs, so one staged step feeds THREE v_mfma_f32_16x16x32_f16 per fragment pair — w_hi x_hi + w_lo x_hi + (w_hi 2^-11) x_lo, fp32
 * accumulate; the lo x lo term (2^-22 relative) is dropped.  Weights (DY_WLAYOUT_ROWS only) are packed the same way per (tap, 8
 * channels) after each output-channel row was scaled by a power of two into [2^13, 2^14) (w_lo unscaled); dy_conv_desc.w_scale[co]
 * holds the inverse power and multiplies the accumulator in the epilogue.  Built in dy_conv2d_nhwc (dense 1x1 / 3x3, stride 1 / 2,
 * residual, x2 / up2x, out_f32; and the GROUPED form of DWConv, conv.py:102-107 — cout == groups, 1 / 2 / 4 input channels per group,
 * cout % 8 == 0, plain call: there w is fp32 [cout][ksize * ksize * (cin / groups)], unscaled, no w_scale, and the kernel multiplies joined
 * fp32 inputs by it), dy_nchw_f32_to_nhwc, dy_nhwc_to_nchw_f32, dy_sppf_maxpool3 and the chunk copies; the reference has
 * no counterpart (its CPU path is fp32: this type reproduces it to ~4 fp32 ulps per layer, see tests/test_kernels_gpu.py). */
/* DY_ACT_SILU_L2E: SiLU in the log2(e)-SCALED activation domain.  The caller packs every bias multiplied by log2(e) (and the weights of a
 * layer that reads unscaled data, e.g. the image, likewise), so the accumulator holds t = log2(e) * z; the epilogue computes
 * t / (1 + 2^-t) = log2(e) * silu(z): every stored activation of the pass is log2(e) times the reference's (max pool, nearest upsample,
 * Concat and the Bottleneck sum commute with the positive scale) and the layers that leave the domain (the last 1x1 of each Detect
 * branch) carry weights divided by log2(e).  One VALU instruction less per output element than DY_ACT_SILU (sigmoid(z) = 1 / (1 + 2^-t)
 * needs no multiply in front of v_exp_f32).  The host mirror uses it for 16-bit / fp8 inference (drone-yolo_amd/hip_ops.py::scaled_activations). */
typedef enum dy_act { DY_ACT_NONE = 0, DY_ACT_SILU = 1, DY_ACT_SILU_L2E = 2 } dy_act;
/* Packed weight layouts of dy_conv2d_nhwc (see dy_conv_desc.w_layout). */
typedef enum dy_wlayout { DY_WLAYOUT_ROWS = 0, DY_WLAYOUT_HALO3X3 = 1, DY_WLAYOUT_FRAG1X1 = 2 } dy_wlayout;

/* ---- library -------------------------------------------------------------- */

/* (major << 16) | minor. */
int32_t dy_version(void);
/* Message of the last error raised on this thread ("" if none). Host string. */
const char* dy_last_error_string(void);
/* Introspection for measurement (bench.py's roofline.dominant_kernel): name of the device kernel the last launching call on this
 * thread dispatched to ("conv3x3_vgemm16_kernel", "conv_gemm_glds_kernel<256,256>", ...), so that live per-launch timings can be
 * grouped by the same symbols a rocprofv3 kernel trace reports.  Host string with static storage; "" before the first launch. */
const char* dy_last_kernel_name(void);
/* Size in bytes of one element of `dtype` (2, 2, 4, 1, 4) or 0 if unknown. */
int32_t dy_dtype_size(int32_t dtype);

/* ---- convolution ----------------------------------------------------------
 * Replaces: Conv.forward / forward_fuse = SiLU(BN(conv2d(x))) (nn/modules/conv.py:37-55),
 * RepVGGBlock.forward after get_equivalent_kernel_bias folding (nn/modules/block.py:1421-1490),
 * Bottleneck.forward's residual add (block.py:348-350), the plain nn.Conv2d 1x1 heads
 * of Detect (nn/modules/head.py:43-57) and DWConv (conv.py:102-107, groups > 1).
 *
 * y[n,ho,wo,co] = act( sum_{r,q,c} x[n, ho*stride - pad + r, wo*stride - pad + q, c]
 *                                   * w[co][(r*ksize + q)*cin + c]  + bias[co] ) (+ residual)
 *
 * Weights are PACKED by the caller:  row co holds the K = ksize*ksize*cin taps in
 * (r, q, c) order, rows are k_pad elements apart, k_pad = dy_conv_k_pad(...), rows
 * co >= cout up to cout_pad = dy_conv_cout_pad(cout) and columns k >= K must be ZERO.
 * BatchNorm (eps, running stats) and RepVGG branches are folded into w/bias by the
 * caller (utils/torch_utils.py:242-269 is the folding rule).
 * bias: fp32[cout_pad].  residual (optional, same dtype as x) is added AFTER the
 * activation.  Requirements: cin % (16/elem_size) == 0, ld_* and view bases aligned
 * to 16 bytes (vector path) — unaligned OUTPUT views fall back to scalar stores.
 */
typedef struct dy_conv_desc {
  const void* x;         /* input view base */
  const void* w;         /* packed weights [cout_pad][k_pad] */
  const float* bias;     /* [cout_pad] fp32 */
  const void* residual;  /* optional view (n,ho,wo,cout), pitch ld_res; NULL = none */
  void* y;               /* output view base */
  int32_t batch, h, w_in, cin, ld_x;
  int32_t ho, wo, cout, ld_y, ld_res;
  int32_t ksize, stride, pad;
  int32_t groups;        /* 1 = dense (MFMA path); >1 = grouped/depthwise direct kernel:
                            w is then [cout][ksize*ksize*(cin/groups)] unpadded */
  int32_t act;           /* dy_act */
  int32_t dtype;         /* dy_dtype of x, w, residual */
  int32_t out_f32;       /* 1: y is fp32 regardless of dtype (Detect logits) */
  int32_t k_pad, cout_pad;
  int32_t up2x;          /* 1: x is read through a fused 2x nearest upsample
                            (nn.Upsample(None,2,'nearest') folded into this conv's gather):
                            h,w_in are the UPSAMPLED dims, the buffer holds (h/2, w_in/2).
                            2: x is read ZERO-DILATED by 2 (logical (h, w_in) = 2x the buffer dims, value at even
                            (row, col) only, zero elsewhere): the gather of a stride-2 transposed convolution, used for
                            the input gradient of stride-2 layers.  DY_WLAYOUT_ROWS only, no x2. */
  /* Optional second source = Concat folded into the gather (conv.py:323-333): input
   * channels [0,cin_split) are read from x (through up2x if set), channels
   * [cin_split,cin) from x2 (never upsampled), pitch ld_x2.  x2 = NULL: single source. */
  const void* x2;
  int32_t ld_x2, cin_split;
  /* Weight layout.  DY_WLAYOUT_ROWS (0): the [cout_pad][k_pad] rows described above (any ksize/stride).
   * DY_WLAYOUT_HALO3X3 (1): dense 3x3, pad 1, stride 1 or 2, single source, cout % 4 == 0 — selects the
   * LDS-halo kernel.  Weights are then packed in MFMA-fragment order: with E = 16/elem_size,
   * KC = 4*E channels per chunk, BN = (cout > 32 ? 64 : 32), NF = BN/16, the element
   *   w[co = nt*BN + j*16 + lr][r][q][ci = c*KC + lq*E + e]
   * lives at ((((nt*ceil(cin/KC) + c)*9 + (r*3+q))*NF + j)*64 + lq*16 + lr)*E + e, zero where co >= cout
   * or ci >= cin.  k_pad / cout_pad are ignored; bias stays fp32[dy_conv_cout_pad(cout)].
   * DY_WLAYOUT_FRAG1X1 (2): dense 1x1, stride 1, no residual (x2 / up2x allowed) — selects the streaming
   * kernel.  Same fragment order with a single tap: BN = (cout > 64 ? 128 : cout > 16 ? 64 : 16), NF = BN/16,
   *   w[co = nt*BN + j*16 + lr][ci = c*KC + lq*E + e]  at  (((nt*ceil(cin/KC) + c)*NF + j)*64 + lq*16 + lr)*E + e.
   * Built for ceil(cin/KC) in {2,3,4,6,8,12}; cout <= 16 and fp32 output (out_f32 with a 16-bit dtype) only
   * for 2 chunks; anything else returns DY_ERR_UNSUPPORTED (pack such layers with DY_WLAYOUT_ROWS). */
  int32_t w_layout;
  /* DY_FP8 only (BASELINE config 5: fp8 weights / activations on the fp8 MFMA, fp32 accumulate).  Weights are quantised per OUTPUT
   * channel, activations with ONE scale for the whole network (e4m3 is a floating format: a per-tensor scale only has to keep
   * the values inside [2^-9, 448] * scale):
   *   y_real[co] = act( (sum q_x * q_w[co]) * w_scale[co] + bias[co] ) (+ q_res * act_scale);   y_q = sat_e4m3(y_real / act_scale)
   * w_scale: fp32[cout_pad], = act_scale * (weight scale of channel co); with out_f32 the result is y_real itself.
   * DY_WLAYOUT_ROWS only.  Ignored (may be NULL / 0) for the other dtypes. */
  const float* w_scale;
  float act_scale;
  /* Training forward in front of a train-mode BatchNorm (conv.py:49-51): optional pointer to that BatchNorm's dy_bn_desc.workspace.
   * A kernel built for it stores, from its epilogue, per-channel partial sums and sums of squares of the STORED output values there
   * (one slot per spatial workgroup, in the layout of dy_bn_train_fwd's own reduction pass, totals zeroed), so the batch statistics
   * need no pass of their own: dy_conv_stats_written() returns the number of slots the calling thread's last dy_conv2d_nhwc wrote
   * (0: the dispatched kernel has no such epilogue, nothing was touched) -- hand it to dy_bn_desc.partial_slabs.  NULL = off. */
  double* bn_stats;
  /* Storage type of y when it differs from `dtype` (mixed-precision plans of BASELINE config 5: an fp8 trunk in front of a 16-bit
   * Detect tail, a 16-bit layer handing over to the fp8 trunk).  0: y has type `dtype` (or fp32 with out_f32); k > 0: y has
   * dy_dtype k - 1.  Built in the flat-K kernel (csrc/conv_gemm_fk.hip, DY_WLAYOUT_ROWS) for DY_FP8 -> DY_F16 and DY_F16 -> DY_FP8
   * (act_scale > 0 then gives the output quantum: y_q = sat_e4m3(y_real / act_scale)); any other pair returns DY_ERR_UNSUPPORTED. */
  int32_t y_dtype1;
  /* Training backward (r05).  This convolution computes an INPUT gradient whose output IS the gradient dy that reaches the train-mode
   * BatchNorm + activation of the layer in front (conv.py:49-51 backwards; that layer's output has this convolution as its only consumer,
   * so nothing is added to dy afterwards).  With bnb_z != NULL a kernel built for it reads that layer's saved pre-BatchNorm map z beside
   * its own stores and leaves, in bn_stats (= that BatchNorm's dy_bn_desc.workspace), per-channel partial sums of
   *     du = dy * act'(u)  and  du * xhat,      xhat = (z - mean) * rstd,  u = gamma * xhat + beta,   dy as STORED,
   * in the slot layout of dy_bn_train_bwd's own reduction pass (totals zeroed): dy_conv_stats_written() > 0 then goes to
   * dy_bn_desc.partial_slabs of dy_bn_train_bwd, which skips its pass over dy and z.  0 slots: the dispatched kernel has no such epilogue
   * and touched nothing (built: the register-weight 3x3 kernel, 16-bit, stride 1, 64 or 128 channels in, cout % 64 == 0, no residual).
   * bnb_z: (n, ho, wo, cout) view of pitch bnb_ld_z and type `dtype`; bnb_act: dy_act of that layer; mean / rstd / gamma / beta: fp32[cout].
   * With bnb_z set, bn_stats is never used for the forward statistics. */
  const void* bnb_z;
  int32_t bnb_ld_z, bnb_act;
  const float* bnb_mean;
  const float* bnb_rstd;
  const float* bnb_gamma;
  const float* bnb_beta;
} dy_conv_desc;

/* Quantise a 16-bit / fp32 NHWC view to DY_FP8: dst_q = sat_e4m3(src / act_scale).  c % 16 == 0, views 16-byte aligned.
 * Replaces nothing in the reference (it has no fp8 path); it is the hand-over from the image stem (run in fp16) to the fp8 layers. */
int32_t dy_quantize_fp8_nhwc(const void* src, void* dst, int64_t rows, int32_t c, int32_t ld_src, int32_t ld_dst, int32_t src_dtype, float act_scale,
                             dy_stream_t stream);
/* Pack the fp32 master weights of a convolution into a dy_conv_desc weight layout, on the device, in ONE launch (training re-packs
 * every step; replaces the host-side pad / permute / flip / cast of hip_ops.PackedConv).  w: fp32, read through the element strides
 * (s_co, s_ci, s_r, s_q) of its (cout, cin, r, q) axes — any memory layout.  transpose_flip = 1 packs the weights of the convolution
 * that computes the INPUT gradient: W'[ci][co][r][q] = w[co][ci][k-1-r][k-1-q] (logical cout / cin swapped).  cin_logical > cin: the
 * packed convolution sees cin_logical input channels, the extra ones zero (the 3-channel image padded to one chunk).  dst: `dtype`
 * (not DY_FP8), exactly the element count of the layout (ROWS: dy_conv_cout_pad x dy_conv_k_pad; fragment layouts as documented
 * at dy_conv_desc.w_layout), padding written as zeros. */
int32_t dy_pack_conv_weights(const float* w, int64_t s_co, int64_t s_ci, int64_t s_r, int64_t s_q, int32_t cout, int32_t cin, int32_t ksize,
                             int32_t transpose_flip, int32_t cin_logical, void* dst, int64_t dst_elems, int32_t dtype, int32_t w_layout, dy_stream_t stream);
/* The same for MANY convolutions in ONE launch (a training step re-packs every layer's weights twice: forward and input-gradient
 * form, ~160 launches of ~5 us).  dy_pack_conv_weights_table validates the jobs (the arguments of dy_pack_conv_weights, one dtype)
 * and writes the launch table into HOST memory (dy_pack_conv_weights_table_bytes(n) bytes) plus the grid size; the caller copies it to
 * the device once; dy_pack_conv_weights_batched runs it (re-usable while the source / destination addresses stand; capturable). */
typedef struct dy_pack_job {
  const float* w;
  int64_t s_co, s_ci, s_r, s_q;
  int32_t cout, cin, ksize, transpose_flip, cin_logical, w_layout;
  void* dst;
  int64_t dst_elems;
} dy_pack_job;
int64_t dy_pack_conv_weights_table_bytes(int32_t n_jobs);
int32_t dy_pack_conv_weights_table(const dy_pack_job* jobs, int32_t n_jobs, int32_t dtype, void* table_host, int64_t table_bytes, int32_t* total_blocks);
int32_t dy_pack_conv_weights_batched(const void* table_dev, int32_t n_jobs, int32_t total_blocks, int32_t dtype, dy_stream_t stream);
int32_t dy_conv_k_pad(int32_t cin, int32_t ksize, int32_t dtype);
int32_t dy_conv_cout_pad(int32_t cout);
int32_t dy_conv2d_nhwc(const dy_conv_desc* d, dy_stream_t stream);
int32_t dy_conv_stats_written(void); /* > 0: slots of d->bn_stats the last dy_conv2d_nhwc of this thread filled */

/* ---- one Detect branch from its second 3x3 convolution to the decoded output ------------------------------
 * Replaces in ONE kernel per (level, branch), 16-bit storage: Detect.forward's cv2[i][1] -> cv2[i][2] (kind 1, box) or cv3[i][1] ->
 * cv3[i][2] (kind 2, class) of the legacy v8 head (nn/modules/head.py:43-57, 64-70), each Conv = SiLU(conv + folded BatchNorm
 * bias) (conv.py:53-55), and that branch's share of Detect._inference (head.py:100-131): kind 1 = DFL (block.py:58-76) + dist2bbox on the
 * anchor grid (tal.py:333-357) x stride -> rows 0..3 of pred; kind 2 = sigmoid -> rows 4.. of pred + the NMS candidate filter
 * (ops.py:250,290-295: best class score > conf_thres, optional class mask) appended to the dy_nms workspace.
 * x: NHWC (batch, h, w, c_in) pitch ld_x, the output of the branch's FIRST conv.  w3 / b3: the 3x3 c_in -> c_mid conv in
 * DY_WLAYOUT_HALO3X3, bias fp32[64].  w1 / b1: the plain 1x1 conv in DY_WLAYOUT_FRAG1X1 (kind 1: c_mid -> 4*reg_max, bias fp32[64];
 * kind 2: c_mid -> nc <= 16, bias fp32[16] zero padded).  out: pred (batch, 4 + nc, anchors) fp32; this level's anchors are
 * [anchor0, anchor0 + h*w) in row-major (y, x) order.  kind 2 with nms_workspace: candidates are APPENDED — zero the counts once per
 * pass with dy_nms_reset_counts before the first class branch; then dy_nms(prefiltered = 1).
 * Built for c_in = c_mid = 64, reg_max 16, DY_BF16 / DY_F16 (dy_detect_branch_fused_supported tells). */
typedef struct dy_branch_desc {
  const void* x;
  const void* w3;
  const float* b3;
  const void* w1;
  const float* b1;
  float* out;
  int32_t batch, h, w, ld_x, c_in, c_mid, nc, reg_max, kind, dtype, anchors, anchor0;
  float stride;
  void* nms_workspace;
  int64_t nms_workspace_bytes;
  float conf_thres;
  const uint8_t* classes_mask;
  int32_t act_l2e; /* 1: the trunk conv's SiLU is DY_ACT_SILU_L2E (b3 scaled by log2 e, w1 divided by it: the logits stay in true units) */
} dy_branch_desc;
int32_t dy_detect_branch_fused_supported(int32_t c_in, int32_t c_mid, int32_t c_out, int32_t kind, int32_t nc, int32_t reg_max, int32_t dtype);
int32_t dy_detect_branch_fused(const dy_branch_desc* d, dy_stream_t stream);
int32_t dy_nms_reset_counts(void* nms_workspace, int32_t batch, dy_stream_t stream);

/* ---- fused C2f block (n = 1, hidden 32) -------------------------------------------------------------
 * Replaces in ONE kernel: C2f.forward (nn/modules/block.py:237-242) = cv1 (Conv 1x1 cin -> 2*hidden) -> chunk(2) ->
 * Bottleneck(hidden, hidden, shortcut, k = (3,3), e = 1.0) (block.py:337-350) -> cat -> cv2 (Conv 1x1 3*hidden -> cout), each
 * Conv = SiLU(conv + folded BatchNorm bias) (conv.py:53-55), for the stride-4 C2f blocks of Drone-YOLO-s: the backbone's
 * (yolov8-p2-repvgg.yaml layer 2) and, with cin_lo > 0, the neck's (layer 21) together with the nn.Upsample(2, 'nearest') and
 * Concat (conv.py:323-335) in front of it: the block input is then [upsample2x(x_lo) (cin_lo channels) | x (cin - cin_lo)].
 * Intermediates stay on chip, rounded to `dtype` where the layer-by-layer path rounds.
 * x: NHWC (batch, h, w, cin - cin_lo) pitch ld_x; x_lo: NHWC (batch, h/2, w/2, cin_lo) pitch ld_x_lo, or NULL with cin_lo = 0;
 * y: NHWC (batch, h, w, cout) pitch ld_y.  Weights (BatchNorm folded):
 *   w_cv1   DY_WLAYOUT_FRAG1X1 of (2*hidden, cin);      w_cv2   DY_WLAYOUT_FRAG1X1 of (cout, 3*hidden);
 *   w_m_cv1, w_m_cv2   DY_WLAYOUT_HALO3X3 of (hidden, hidden, 3, 3);
 *   bias    fp32: cv1 [2*hidden] | m.cv1 [hidden] | m.cv2 [hidden] | cv2 [cout].
 * Built for 64 direct channels (+ 128 upsampled), hidden 32, cout 64, DY_BF16 / DY_F16 (dy_c2f_fused_supported tells); other
 * shapes: run the four dy_conv2d_nhwc calls. */
typedef struct dy_c2f_desc {
  const void* x;
  const void* x_lo;
  void* y;
  const void* w_cv1;
  const void* w_m_cv1;
  const void* w_m_cv2;
  const void* w_cv2;
  const float* bias;
  int32_t batch, h, w, cin, cin_lo, hidden, cout, ld_x, ld_x_lo, ld_y, shortcut, dtype;
  int32_t act_l2e; /* 1: all four SiLUs are DY_ACT_SILU_L2E (input, output and biases in the log2(e)-scaled domain) */
} dy_c2f_desc;
int32_t dy_c2f_fused_supported(int32_t cin, int32_t cin_lo, int32_t hidden, int32_t cout, int32_t n_bottlenecks, int32_t dtype);
int32_t dy_c2f_fused(const dy_c2f_desc* d, dy_stream_t stream);

/* Fused stem.  Replaces in one pass: the predictor's dtype/layout step for tensor sources
 * (engine/predictor.py:118-136) AND the model's first layer Conv(cin<=3, cout, 3, 2) (nn/modules/conv.py:37-55,
 * yolov8-p2-repvgg.yaml layer 0), so the image is never materialised in NHWC.
 * x: fp32 NCHW (n, cin, h, w) contiguous.  w: [ceil(cout/16)*16][32] of `dtype`, row co = the folded taps in
 * k = c*9 + r*3 + q order (i.e. OIHW flattened), zero padded to 32; bias fp32[ceil(cout/16)*16].
 * y: NHWC view (n, (h-1)/2+1, (w-1)/2+1, cout) of `dtype`, pitch ld_y.  cout <= 80.
 * DY_F16X2 (round 5): w = [cout_pad16][32] float16 hi halves, then as many lo halves, then fp32[cout_pad16] inverse row scales in ONE buffer
 * (rows scaled by a power of two into [2^13, 2^14) before the split, as for dy_conv2d_nhwc); the image is split on the fly; cout % 8 == 0, <= 64. */
int32_t dy_stem_conv3x3s2_nchw(const float* x, const void* w, const float* bias, void* y, int32_t n, int32_t cin,
                               int32_t h, int32_t w_in, int32_t cout, int32_t ld_y, int32_t act, int32_t dtype,
                               dy_stream_t stream);

/* The same from a uint8 NCHW image, value = x / divisor: the training input (DetectionTrainer.preprocess_batch's `img.float() / 255`,
 * models/yolo/detect/train.py:57-60) consumed by the stem directly.  16-bit storage, cout a multiple of 16 (<= 80); the operand values
 * are exactly those of dy_nchw_u8_to_nhwc followed by dy_conv2d_nhwc (same division, same rounding to `dtype`). */
int32_t dy_stem_conv3x3s2_nchw_u8(const uint8_t* x, float divisor, const void* w, const float* bias, void* y, int32_t n, int32_t cin,
                                  int32_t h, int32_t w_in, int32_t cout, int32_t ld_y, int32_t act, int32_t dtype, dy_stream_t stream);

/* Fused first TWO layers.  Replaces in one kernel: the layout step + Conv(3, 32, 3, 2) (yolov8-p2-repvgg.yaml layer 0,
 * nn/modules/conv.py:37-55) + RepVGGBlock(32, 64, stride 2) in deploy form (layer 1, nn/modules/block.py:1393-1490:
 * get_equivalent_kernel_bias folds the three branches into one 3x3 kernel), each followed by SiLU.  The 1/2-resolution
 * 32-channel intermediate stays in LDS, rounded to `dtype` exactly where the layer-by-layer path rounds it.
 * x: fp32 NCHW (n, 3, h, w) contiguous.  w0/b0: as dy_stem_conv3x3s2_nchw ([32][32], k = c*9 + r*3 + q; fp32[32]).
 * w1: [64][288] of `dtype`, k = (r*3 + q)*32 + c (BatchNorm and branches folded); b1: fp32[64].
 * y: NHWC view (n, h/4, w/4, 64) of `dtype`, pitch ld_y.  Built for DY_BF16 / DY_F16 and h, w multiples of 4
 * (dy_stem2_fused_supported tells); otherwise run dy_stem_conv3x3s2_nchw and dy_conv2d_nhwc.
 * DY_F16X2 (round 5): w0 = the split stem pack of dy_stem_conv3x3s2_nchw ([32][32] float16 hi halves, as many lo halves, fp32[32] inverse row
 * scales, one buffer); w1 = the split DY_WLAYOUT_ROWS pack of dy_conv2d_nhwc for a 3x3 32 -> 64 layer ([64][9 taps x 4 groups x (hi x 8 | lo x 8)]
 * float16) with its inverse row scales in w1_scale (fp32[64]); act0 = act1 = DY_ACT_SILU; y: split-float16 view, ld_y in 4-byte elements and
 * a multiple of 8.  The intermediate is rounded to (hi, lo) pairs where the layer-by-layer path rounds it. */
typedef struct dy_stem2_desc {
  const float* x;
  const void* w0;
  const float* b0;
  const void* w1;
  const float* b1;
  void* y;
  int32_t n, h, w, ld_y, act0, act1, dtype;
  const float* w1_scale; /* DY_F16X2 only (NULL otherwise) */
} dy_stem2_desc;
int32_t dy_stem2_fused_supported(int32_t cin, int32_t c0, int32_t c1, int32_t h, int32_t w, int32_t dtype);
int32_t dy_stem2_fused(const dy_stem2_desc* d, dy_stream_t stream);

/* ---- image sources: LetterBox + BGR->RGB + HWC->CHW + /255 ------------------------------------------
 * Replaces: LetterBox.__call__ (ultralytics/data/augment.py:1545-1608: cv2.resize INTER_LINEAR to (new_w, new_h), then
 * cv2.copyMakeBorder with 114) and the non-tensor branch of BasePredictor.preprocess (engine/predictor.py:125-135:
 * `im[..., ::-1].transpose(0,3,1,2)`, `.float()`, `/= 255`), in one pass.
 * src: DEVICE uint8 (n, h0, w0, 3), all frames one shape, channel order as decoded (BGR): swap_rb = 1 reverses it.
 * dst: fp32 (n, 3, hn, wn) contiguous = what dy_stem_conv3x3s2_nchw consumes.  The resized image occupies rows
 * [top, top+new_h) x columns [left, left+new_w); everything else is pad_value (114).  The geometry (new_w, new_h, top,
 * left, hn, wn) is LetterBox's own host arithmetic (augment.py:1566-1591).  8-bit bilinear arithmetic: OpenCV's
 * (11-bit coefficients), bit-exact against oracle/letterbox_oracle.py; new size == source size copies. */
int32_t dy_letterbox_u8_to_nchw_f32(const uint8_t* src, float* dst, int32_t n, int32_t h0, int32_t w0, int32_t new_w, int32_t new_h,
                                    int32_t top, int32_t left, int32_t hn, int32_t wn, int32_t swap_rb, float pad_value,
                                    dy_stream_t stream);

/* multi_scale of DetectionTrainer.preprocess_batch (models/yolo/detect/train.py:60-73): a uint8 NCHW batch -> fp32 NCHW (n, c, ho, wo) holding
 * nn.functional.interpolate(src.float() / 255, size=(ho, wo), mode="bilinear", align_corners=False) — torch's coordinate rule and blend order. */
int32_t dy_resize_bilinear_u8_nchw_f32(const uint8_t* src, float* dst, int32_t n, int32_t c, int32_t h, int32_t w, int32_t ho, int32_t wo, dy_stream_t stream);
/* scale_img of test-time augmentation (utils/torch_utils.py:436-445, called by DetectionModel._predict_augment, nn/tasks.py:347-383): the fp32 NCHW batch
 * (n, c, h, w) — read mirrored left-right when flip_lr — resized bilinearly (align_corners = False, torch's coordinate rule) to (hs, ws) in the top-left
 * corner of dst (n, c, ho, wo), the rest filled with `pad` (the reference pads with 0.447 to multiples of the model's largest stride). */
int32_t dy_scale_img_nchw_f32(const float* src, float* dst, int32_t n, int32_t c, int32_t h, int32_t w, int32_t hs, int32_t ws, int32_t ho, int32_t wo,
                              int32_t flip_lr, float pad, dy_stream_t stream);

/* ---- tiled inference on large frames -----------------------------------------------------------------
 * The reference slices through third-party packages that are not vendored (mix6.py:84-89 `sv.InferenceSlicer`,
 * examples/YOLOv8-SAHI-Inference-Video/yolov8_sahi.py:50-55): fixed-size tiles with a fractional overlap, one inference
 * per tile, detections shifted back and merged by class-aware NMS.  Parity is unpinned there; this build defines it.
 * dy_tiles_u8_to_nchw_f32: k crops (th x tw at offsets_yx[k] = (y, x), DEVICE int32) of ONE uint8 HWC frame (hf, wf, 3) ->
 *   fp32 (k, 3, th, tw) / 255 (swap_rb as in dy_letterbox_u8_to_nchw_f32); beyond the frame edge: pad_value.
 * dy_rows_to_pred: per-tile dy_nms outputs rows (k, max_det, 6) + counts (k) -> pred (1, 4+nc, k*max_det) fp32 in FRAME
 *   coordinates (xywh, the row's score in its class channel, zeros elsewhere and for rows >= counts) for a final dy_nms. */
int32_t dy_tiles_u8_to_nchw_f32(const uint8_t* frame, const int32_t* offsets_yx, float* dst, int32_t k, int32_t hf, int32_t wf, int32_t th,
                                int32_t tw, int32_t swap_rb, float pad_value, dy_stream_t stream);
int32_t dy_rows_to_pred(const float* rows, const int32_t* counts, const int32_t* offsets_yx, float* pred, int32_t k, int32_t max_det,
                        int32_t nc, dy_stream_t stream);

/* ---- layout / copy ops ------------------------------------------------------ */

/* Replaces: predictor preprocess `.half()/.float()` + the NCHW->device layout step
 * (engine/predictor.py:118-136).  src: fp32 NCHW (n,c,h,w) contiguous.  dst: NHWC
 * view of `dtype`, pitch ld_dst, channels [c, c_pad) are written as zero. */
int32_t dy_nchw_f32_to_nhwc(const float* src, void* dst, int32_t n, int32_t c, int32_t h, int32_t w,
                            int32_t c_pad, int32_t ld_dst, int32_t dtype, dy_stream_t stream);

/* Training input: uint8 NCHW batch -> NHWC of `dtype`, every value divided by `divisor` (255), channels zero-padded to c_pad.
 * Replaces DetectionTrainer.preprocess_batch's `batch["img"].float() / 255` (models/yolo/detect/train.py:57-60) + the
 * layout step, in one pass (1 byte read per value). */
int32_t dy_nchw_u8_to_nhwc(const uint8_t* src, void* dst, int32_t n, int32_t c, int32_t h, int32_t w, int32_t c_pad,
                           int32_t ld_dst, float divisor, int32_t dtype, dy_stream_t stream);

/* Inverse, for handing activations back to NCHW callers: src NHWC view -> fp32 NCHW. */
int32_t dy_nhwc_to_nchw_f32(const void* src, float* dst, int32_t n, int32_t c, int32_t h, int32_t w,
                            int32_t ld_src, int32_t src_dtype, dy_stream_t stream);

/* Replaces: nn.Upsample(None, 2, 'nearest') feeding Concat (yolov8-p2-repvgg.yaml:30,34,38).
 * src (n,h,w,c) pitch ld_src -> dst (n,2h,2w,c) pitch ld_dst (typically a channel
 * slice of the Concat buffer). c % (16/elem_size) == 0. */
int32_t dy_upsample2x_nhwc(const void* src, void* dst, int32_t n, int32_t h, int32_t w, int32_t c,
                           int32_t ld_src, int32_t ld_dst, int32_t dtype, dy_stream_t stream);

/* Strided NHWC copy (Concat fallback when a producer could not write in place;
 * conv.py:323-333). */
int32_t dy_copy_nhwc(const void* src, void* dst, int32_t n, int32_t h, int32_t w, int32_t c,
                     int32_t ld_src, int32_t ld_dst, int32_t dtype, dy_stream_t stream);

/* Replaces: SPPF's three chained MaxPool2d(k,1,k//2) (nn/modules/block.py:185-191).
 * x: (n,h,w,c) pitch ld.  y1,y2,y3: pooled once/twice/thrice, same pitch ld (the
 * channel slices of the SPPF concat buffer).  k odd, (k/2)*3 halo; h*w*16B*3 must
 * fit LDS (h*w <= 4608; a workgroup takes up to eight 16-byte channel chunks of one image: whole lines). */
int32_t dy_sppf_maxpool3(const void* x, void* y1, void* y2, void* y3, int32_t n, int32_t h, int32_t w,
                         int32_t c, int32_t ld, int32_t k, int32_t dtype, dy_stream_t stream);

/* ---- Detect decode ------------------------------------------------------------
 * Replaces: Detect._inference (nn/modules/head.py:100-131) = DFL (block.py:58-76) +
 * make_anchors (utils/tal.py:333-345) + dist2bbox xywh (tal.py:348-357) * stride +
 * sigmoid(cls), concatenated to (batch, 4+nc, A).
 * Per level l the raw head output is an fp32 NHWC view (n, h_l, w_l, 4*reg_max + nc)
 * with pitch ld_l: channels [0,4*reg_max) are the box bins, the next nc the class logits.
 * out: fp32 (batch, 4+nc, A) contiguous, A = sum_l h_l*w_l, levels in order. */
#define DY_MAX_LEVELS 8
typedef struct dy_decode_desc {
  const float* level[DY_MAX_LEVELS];
  int32_t h[DY_MAX_LEVELS], w[DY_MAX_LEVELS], ld[DY_MAX_LEVELS];
  float stride[DY_MAX_LEVELS];
  int32_t n_levels, batch, nc, reg_max;
  float* out;
  /* Optional fusion of the NMS candidate filter (ops.py:250,290-295) into the decode pass: when
   * nms_workspace != NULL (a dy_nms workspace for the same batch/anchors) every anchor whose best class
   * score > conf_thres (and passes classes_mask) is appended to it, and dy_nms can then be called with
   * prefiltered = 1 on `out`, skipping its own pass over the predictions. */
  void* nms_workspace;
  int64_t nms_workspace_bytes;
  float conf_thres;
  const uint8_t* classes_mask;
} dy_decode_desc;
int32_t dy_detect_decode(const dy_decode_desc* d, dy_stream_t stream);

/* ---- fused Detect tail: last 1x1 convs of both branches + decode (+ NMS filter) ---------------
 * Replaces in ONE pass, per level l: cv2[l][2] = nn.Conv2d(c_box, 4*reg_max, 1) and cv3[l][2] =
 * nn.Conv2d(c_cls, nc, 1) (nn/modules/head.py:43-57), their concat (head.py:69-70) and Detect._inference
 * (head.py:100-131), i.e. dy_conv2d_nhwc x2 + dy_detect_decode without the (batch, 4*reg_max+nc, A) fp32
 * logits ever reaching HBM.  Inference only: the raw per-level maps the reference also returns (head.py:74)
 * are not produced (use the unfused calls when they are wanted, e.g. for the training loss).
 * x_box[l] / x_cls[l]: NHWC views (batch, h_l, w_l, c_box / c_cls) of `dtype`, pitches ld_box / ld_cls —
 * the outputs of cv2[l][1] / cv3[l][1].  w_box[l] / w_cls[l]: the 1x1 weights packed as DY_WLAYOUT_FRAG1X1
 * for (cout = 4*reg_max, cin = c_box) / (cout = nc, cin = c_cls); b_box / b_cls: fp32 bias padded like there.
 * out, nms_workspace, conf_thres, classes_mask: exactly as in dy_decode_desc.
 * Built for reg_max 16, nc <= 128, c_box = 64, c_cls in {64, 96, 128, 160} (16-bit) or {64, 80} (fp32);
 * dy_detect_head_decode_supported() tells (1/0); unsupported shapes return DY_ERR_UNSUPPORTED.
 * DY_F16X2 (round 5; c_box = 64, c_cls in {64, 96, 128}): x_box / x_cls are split-float16 views; w_box[l] / w_cls[l] hold the float16
 * DY_WLAYOUT_FRAG1X1 image (k-groups of 32 channels) of the hi halves followed by the image of the lo halves (rows scaled by a power of
 * two into [2^13, 2^14) first, as for dy_conv2d_nhwc); b_box[l] / b_cls[l] hold the padded bias followed by as many inverse row scales. */
typedef struct dy_head_decode_desc {
  const void* x_box[DY_MAX_LEVELS];
  const void* x_cls[DY_MAX_LEVELS];
  const void* w_box[DY_MAX_LEVELS];
  const void* w_cls[DY_MAX_LEVELS];
  const float* b_box[DY_MAX_LEVELS];
  const float* b_cls[DY_MAX_LEVELS];
  int32_t ld_box[DY_MAX_LEVELS], ld_cls[DY_MAX_LEVELS], h[DY_MAX_LEVELS], w[DY_MAX_LEVELS];
  float stride[DY_MAX_LEVELS];
  int32_t n_levels, batch, nc, reg_max, c_box, c_cls, dtype;
  float* out;
  void* nms_workspace;
  int64_t nms_workspace_bytes;
  float conf_thres;
  const uint8_t* classes_mask;
} dy_head_decode_desc;
int32_t dy_detect_head_decode_supported(int32_t c_box, int32_t c_cls, int32_t nc, int32_t reg_max, int32_t dtype);
int32_t dy_detect_head_decode(const dy_head_decode_desc* d, dy_stream_t stream);

/* ---- NMS --------------------------------------------------------------------
 * Replaces: ops.non_max_suppression (utils/ops.py:181-332; the predictor's single-label path and the validator's multi_label path)
 * including torchvision.ops.nms (called at ops.py:312): candidates = anchors
 * whose best class score > conf_thres; xywh -> xyxy; if more than max_nms keep the
 * max_nms best; boxes offset by cls*max_wh unless agnostic; greedy NMS in
 * descending score order (ties: lower anchor index first), suppress when
 * IoU > iou_thres; first max_det survivors are emitted.
 * pred: fp32 (batch, 4+nc(+nm), A) contiguous (the decode output).
 * classes_mask: optional host-independent DEVICE array of nc bytes (1 = keep class) or NULL.
 * out: fp32 (batch, max_det, 6): x1,y1,x2,y2,conf,cls; rows >= count are zero.
 * out_count: int32 (batch).  out_index: optional int32 (batch, max_det) anchor index of
 * every kept row (for parity tests / mask gathers) or NULL.
 * workspace: dy_nms_workspace_bytes(batch, A) bytes, 16-byte aligned.
 */
typedef struct dy_nms_desc {
  const float* pred;
  int32_t batch, nc, n_extra, anchors;
  float conf_thres, iou_thres;
  int32_t max_det, max_nms;
  float max_wh;
  int32_t agnostic;
  const uint8_t* classes_mask;
  float* out;
  int32_t* out_count;
  int32_t* out_index;
  void* workspace;
  int64_t workspace_bytes;
  int32_t prefiltered; /* 1: the workspace already holds the candidates (dy_detect_decode fused filter,
                          same conf_thres / classes_mask); only sort + suppress run */
  int32_t multi_label; /* 1 (and nc > 1): the validator's NMS (utils/ops.py:255, 286-288; call site models/yolo/detect/val.py:93-106):
                          every (anchor, class) pair scored above conf_thres is a candidate of its own (ties: ascending
                          anchor * nc + class, the reference's row-major torch.where order); out_index still names the ANCHOR.
                          The workspace is then dy_nms_workspace_bytes(batch, anchors * nc) bytes; not with prefiltered */
} dy_nms_desc;
int64_t dy_nms_workspace_bytes(int32_t batch, int32_t anchors);
int32_t dy_nms(const dy_nms_desc* d, dy_stream_t stream);

/* Replaces: ops.scale_boxes + ops.clip_boxes (utils/ops.py:92-127, 335-354) as applied by
 * DetectionPredictor.construct_result (models/yolo/detect/predict.py:59-73) to the kept rows.
 * boxes: fp32 (batch, max_det, 6), updated in place for rows < counts[b]:
 *   x = clamp((x - pad_x) / gain, 0, clip_w),  y = clamp((y - pad_y) / gain, 0, clip_h).
 * params: DEVICE fp32 (batch, 5) = gain, pad_x, pad_y, clip_w, clip_h per image. */
int32_t dy_scale_boxes(float* boxes, const int32_t* counts, const float* params, int32_t batch,
                       int32_t max_det, dy_stream_t stream);

/* ---- training loss (forward + gradient w.r.t. the head outputs) -----------------------------------------------------
 * Replaces: v8DetectionLoss.__call__ (utils/loss.py:206-260) = TaskAlignedAssigner (utils/tal.py:14-295, topk /
 * alpha / beta as given) + BCE class loss + CIoU box loss (utils/metrics.py:74-134) + DFL (loss.py:65-113), on the
 * raw head outputs Detect returns in training mode (head.py:71-72).
 * level[l]: fp32 NHWC view (batch, h_l, w_l, 4*reg_max + nc), pitch ld_l (the Detect head buffers).
 * gt: DEVICE fp32 (batch, gmax, 5) = class, x1, y1, x2, y2 in input pixels, zero rows = padding (the output of
 * v8DetectionLoss.preprocess, loss.py:180-195).  out: fp32[4] = box, cls, dfl (after gains), total = sum * batch.
 * out_owner: optional int32 (batch, A): index of the ground-truth box each anchor is assigned to, or -1.
 * Assignment is sparse (no (batch, gmax, A) tensors).
 * grad_level[l] (optional, all or none): fp32 NHWC (batch, h_l, w_l, 4*reg_max + nc), pitch ld_grad[l]; receives
 * d total / d level[l] — what loss.backward() hands to the Detect head in the reference trainer (engine/trainer.py:381-389)
 * (alignment padding of a row — ld_grad[l] − (4*reg_max + nc) < 8 floats — may be overwritten with zeros: whole-line stores);
 * the assignment, soft targets, target_scores_sum and CIoU's alpha are constants exactly as under autograd
 * (tal.py:60 no_grad, metrics.py:127-128). */
typedef struct dy_loss_desc {
  const float* level[DY_MAX_LEVELS];
  int32_t h[DY_MAX_LEVELS], w[DY_MAX_LEVELS], ld[DY_MAX_LEVELS];
  float stride[DY_MAX_LEVELS];
  int32_t n_levels, batch, nc, reg_max;
  const float* gt;
  int32_t gmax, topk;
  float alpha, beta, box_gain, cls_gain, dfl_gain;
  float* out;
  int32_t* out_owner;
  void* workspace;
  int64_t workspace_bytes;
  float* grad_level[DY_MAX_LEVELS];
  int32_t ld_grad[DY_MAX_LEVELS];
} dy_loss_desc;
int64_t dy_detection_loss_workspace_bytes(int32_t batch, int32_t anchors, int32_t gmax, int32_t topk);
int32_t dy_detection_loss(const dy_loss_desc* d, dy_stream_t stream);

/* ---- convolution gradients ------------------------------------------------------------------
 * Replaces: autograd of F.conv2d inside Conv / RepVGGBlock / Detect (nn/modules/conv.py:37-55) during
 * loss.backward() (engine/trainer.py:381-389).
 * Weight gradient: dw[co][(r*ksize + q)][ci] += sum over output pixels m of dz[m][co] * x[pixel(m) + tap (r,q)][ci].
 *   d describes the FORWARD convolution (x, batch, h, w_in, cin, ld_x, ho, wo, cout, ksize, stride, pad, dtype; w / bias /
 *   y are ignored); dz: NHWC view (batch, ho, wo, cout) of `dtype`, pitch ld_dz; dw: fp32, cout*ksize*ksize*cin, ZEROED by
 *   the caller (partial sums of pixel slabs are added with fp32 atomics).  ld_x / ld_dz must cover cin / cout rounded up to one
 *   16-byte chunk (the padding is read but only reaches entries that are never written).
 * Input gradient: dx = conv_transpose(dz, w) is run through dy_conv2d_nhwc itself on re-packed weights
 *   (w'[ci][2-r][2-q][co] = w[co][r][q][ci], stride 1; stride-2 layers read dz through `dil2`, a zero-dilated gather).
 * dy_colsum: out[c] += sum over rows of z[row][c] (bias gradient of the plain Detect convolutions); out zeroed by the caller;
 *   z 16-byte aligned, pitch a multiple of 16 bytes covering c rounded up to one chunk. */
int32_t dy_conv2d_wgrad_nhwc(const dy_conv_desc* d, const void* dz, int32_t ld_dz, float* dw, dy_stream_t stream);
/* The same with a caller-owned workspace (device memory, 16-byte aligned, at least dy_conv2d_wgrad_workspace_bytes(d, ld_dz) bytes;
 * NULL / too small: the call behaves as dy_conv2d_wgrad_nhwc).  The 3x3 kernel (16-bit storage) then STORES each pixel slab's partial
 * sums in the workspace and a second launch adds them to dw (1/16 of the atomics: 30-50 us less per layer at batch 64).  Calls that
 * share a workspace must be ordered on one stream.  dy_conv2d_wgrad_workspace_bytes: 0 when the call would not use one; < 0 = dy_status. */
int32_t dy_conv2d_wgrad_nhwc_ws(const dy_conv_desc* d, const void* dz, int32_t ld_dz, float* dw, void* workspace, int64_t workspace_bytes,
                                dy_stream_t stream);
int64_t dy_conv2d_wgrad_workspace_bytes(const dy_conv_desc* d, int32_t ld_dz);
int32_t dy_colsum(const void* z, float* out, int64_t rows, int32_t c, int32_t ld, int32_t dtype, dy_stream_t stream);
/* Grouped convolution (DWConv of yolov8-p2-repvgg-sf.yaml:32,38,44; nn/modules/conv.py:102-107, g = gcd(c1, c2)): both gradients
 * of z = conv2d(x, w, stride, pad, groups) given dz — replaces autograd of F.conv2d for that layer.  d describes the FORWARD
 * convolution (x, batch, h, w_in, cin, ld_x, ho, wo, cout, ksize, stride, pad, groups, dtype).  w_oihw / dw_oihw: the fp32 master
 * weight and its gradient in torch's own (cout, cin/groups, k, k) layout; dw_oihw is ADDED to (zeroed or carrying an earlier
 * micro-batch; fp32 atomics); built for 3x3 kernels with 1, 2 or 4 input channels per group.  dx (optional, with w_oihw): NHWC view
 * (batch, h, w_in, cin) of `dtype`, pitch ld_dx, = conv_transpose(dz, w) (+ dx_accumulate when given).  Either output may be NULL. */
int32_t dy_conv2d_grouped_bwd_nhwc(const dy_conv_desc* d, const void* dz, int32_t ld_dz, const float* w_oihw, float* dw_oihw, void* dx, int32_t ld_dx,
                                   const void* dx_accumulate, int32_t ld_acc, dy_stream_t stream);

/* ---- train-mode BatchNorm2d (+ SiLU) -----------------------------------------------------
 * Replaces: the BatchNorm2d + SiLU half of Conv.forward in training mode (nn/modules/conv.py:37-55; eps 1e-3 and
 * momentum 0.03 are set by initialize_weights, utils/torch_utils.py:423-433) and their autograd backward; the two-branch
 * sum of RepVGGBlock.forward (nn/modules/block.py:1480-1490) through `addend` + dy_silu_fwd / dy_silu_bwd.
 * z: the convolution output, (rows, c) of `dtype`, pitch ld_z (rows = batch*h*w of the NHWC view), c a multiple of one
 * 16-byte chunk and <= 2048 (16-bit) / 1024 (fp32).
 * Forward:  mean/rstd (fp32[c], outputs, saved for backward) = batch statistics of z (biased variance);
 *   y = act(gamma * (z - mean) * rstd + beta (+ addend));  running_mean / running_var (optional) are updated as torch
 *   does: r = (1 - momentum) * r + momentum * stat, the variance unbiased.
 * Backward: dy = gradient w.r.t. y; dz = gradient w.r.t. z; dgamma / dbeta fp32[c] (optional).  With act = SILU the
 *   pre-activation is recomputed from z.  `addend` is not used (its gradient is dy * act'(u): dy_silu_bwd).
 * workspace: dy_bn_workspace_bytes(c) bytes (the 2c sums + one partial per reduction slab), 8-byte aligned.  Sums are
 *   accumulated in double. */
typedef struct dy_bn_desc {
  const void* z;
  void* y;
  const void* addend;
  const void* dy;
  void* dz;
  int64_t rows;
  int32_t c, ld_z, ld_y, ld_add, ld_dy, ld_dz, dtype, act;
  const float* gamma;
  const float* beta;
  float* mean;
  float* rstd;
  float* running_mean;
  float* running_var;
  float eps, momentum;
  float* dgamma;
  float* dbeta;
  void* workspace;
  int64_t workspace_bytes;
  /* > 0 = the workspace ALREADY holds that many partial-sum slots (a convolution epilogue wrote them: dy_conv_desc.bn_stats /
   * dy_conv_stats_written) and the reduction pass is skipped — forward: sums of z and z^2 from the convolution in front;
   * backward: sums of du and du * xhat from the input-gradient convolution behind (dy_conv_desc.bnb_z).  0: the pass runs. */
  int32_t partial_slabs;
} dy_bn_desc;
int64_t dy_bn_workspace_bytes(int32_t c);
int32_t dy_bn_train_fwd(const dy_bn_desc* d, dy_stream_t stream);
int32_t dy_bn_train_bwd(const dy_bn_desc* d, dy_stream_t stream);
/* y = u * sigmoid(u);  du = dy * d/du(u * sigmoid(u)).  (rows, c) views of `dtype`. */
int32_t dy_silu_fwd(const void* u, void* y, int64_t rows, int32_t c, int32_t ld_u, int32_t ld_y, int32_t dtype, dy_stream_t stream);
int32_t dy_silu_bwd(const void* u, const void* dy, void* du, int64_t rows, int32_t c, int32_t ld_u, int32_t ld_dy, int32_t ld_du,
                    int32_t dtype, dy_stream_t stream);

/* ---- small training-path ops -----------------------------------------------------------------
 * dy_upsample2x_bwd_nhwc: gradient of nn.Upsample(None, 2, 'nearest') (yolov8-p2-repvgg.yaml:30,34,38):
 *   dx (n,h,w,c) = sum of the 2x2 blocks of g (n,2h,2w,c).
 * dy_maxpool_bwd_nhwc: gradient of MaxPool2d(k, 1, k//2) inside SPPF (nn/modules/block.py:185-191): every output's
 *   gradient goes to the FIRST maximum of its window in (row, column) order (torch semantics); g_in = (accumulate ?
 *   g_in : 0) + that.  h*w*(52 + 64/elem_size) bytes of LDS per workgroup (16-bit: h*w <= ~1900), k <= 15.
 * dy_add_nhwc: out = a + b, (rows, c) views (Bottleneck's shortcut, block.py:348-350).  c % one 16-byte chunk == 0.
 * dy_add_dilated2_nhwc: dx (n, H2, W2, c)[:, 2y, 2x] += t (n, h, w, c)[:, y, x] -- the input gradient of a 1x1 STRIDE-2 convolution
 *   (RepVGGBlock's 1x1 branch, block.py:1480-1490) from t = the 1x1 stride-1 convolution of dz with the transposed weights.
 * dy_head_grad_split: the gradient of Detect's fp32 training map cat(box, cls) (head.py:69-72; g: (rows, nb + nc) fp32, pitch ld_g) as the
 *   `dtype` operands of the two 1x1 convolutions' gradient kernels in one pass: dzb = s * g[:, :nb]; dzc = s * g[:, nb:nb+nc], zero-padded
 *   to ncp channels; s = *scale (optional DEVICE scalar: the seed loss.backward() starts from, the loss scale under fp16) or 1. */
int32_t dy_upsample2x_bwd_nhwc(const void* g, void* dx, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ld_g, int32_t ld_dx,
                               int32_t dtype, dy_stream_t stream);
int32_t dy_maxpool_bwd_nhwc(const void* x, const void* g_out, void* g_in, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ld_x,
                            int32_t ld_go, int32_t ld_gi, int32_t k, int32_t accumulate, int32_t dtype, dy_stream_t stream);
int32_t dy_add_dilated2_nhwc(const void* t, void* dx, int32_t n, int32_t h, int32_t w, int32_t H2, int32_t W2, int32_t c, int32_t ld_t, int32_t ld_dx,
                             int32_t dtype, dy_stream_t stream);
int32_t dy_head_grad_split(const float* g, int32_t ld_g, int64_t rows, int32_t nb, int32_t nc, int32_t ncp, const float* scale, void* dzb, int32_t ld_b,
                           void* dzc, int32_t ld_c, int32_t dtype, dy_stream_t stream);
int32_t dy_add_nhwc(const void* a, const void* b, void* out, int64_t rows, int32_t c, int32_t ld_a, int32_t ld_b, int32_t ld_o,
                    int32_t dtype, dy_stream_t stream);

/* ---- optimizer step, gradient clipping, EMA (flat fp32 tensors) -------------------------------------
 * Replaces: torch.nn.utils.clip_grad_norm_(max_norm 10) + optimizer.step() (engine/trainer.py:591-599) for the SGD
 * (nesterov) / AdamW optimizers build_optimizer creates (trainer.py:764-825: weight decay on the weight group only,
 * so call once per group with its decay), and ModelEMA.update (utils/torch_utils.py:515-545).
 * dy_sumsq_f32: *out += sum g^2 (double; zero it first; call per tensor / bucket, then pass `out` as grad_sumsq).
 * grad_sumsq (optional DEVICE double): total sum of squared gradients; the gradient is scaled by
 *   min(1, max_norm / (sqrt(*grad_sumsq) + 1e-6)) inside the step — no host synchronisation.
 * dy_sgd_step:   g = clip*grad + wd*p;  buf = first_step ? g : momentum*buf + g;  p -= lr * (nesterov ? g + momentum*buf : buf).
 * dy_adamw_step: p *= 1 - lr*wd;  m,v = moments of clip*grad;  p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps); step >= 1.
 * amp_state (optional DEVICE float[4], needs grad_sumsq): the state of torch.cuda.amp.GradScaler (engine/trainer.py:271 `GradScaler`, :389
 *   `scaler.scale(loss).backward()`, :593-596 `unscale_` / `step` / `update`) for fp16 storage — {scale, growth tracker, found_inf of
 *   the last step, skipped steps}.  With it the step kernels divide the gradient by the scale (`unscale_`; the clip norm becomes
 *   sqrt(*grad_sumsq) / scale) and leave p / buf / m / v untouched when *grad_sumsq is not finite (`scaler.step` skips);
 *   AdamW's bias correction counts only the steps that ran.  The caller multiplies the loss gradient by amp_state[0] on the device.
 * dy_amp_update: `scaler.update()` — found_inf: scale *= backoff, tracker = 0, skipped += 1; else tracker += 1 and, every
 *   growth_interval clean steps, scale *= growth (torch defaults: 65536 initial, 2, 0.5, 2000).  Call after the step kernels.
 * dy_ema_update: ema = decay*ema + (1-decay)*p.
 * dy_grad_sink_flush: grad += sink (then sink = 0) for every parameter in ONE launch, where `sink` holds what the backward
 *   kernels produced this batch at the parameter's offset in the flat buffers: weight gradients of dy_conv2d_wgrad_nhwc in
 *   ITS layout (cout, k, k, cin), BatchNorm / bias gradients as they are.  Replaces what autograd's AccumulateGrad does per
 *   parameter (engine/trainer.py:381-389 `loss.backward()`: `param.grad += new`, here ~240 element-wise launches and the
 *   zero-fills of as many temporaries per step).  entries (device, n_entries x 4 int64): {offset, cout, cin, kk}; kk > 1: the
 *   block [offset, offset + cout*cin*kk) of `grad` is (cout, cin, k, k) = torch's layout, of `sink` (cout, k, k, cin);
 *   kk <= 1: plain vector of cout*cin elements. */
int32_t dy_sumsq_f32(const float* g, int64_t n, double* out, dy_stream_t stream);
int32_t dy_sgd_step(float* p, const float* grad, float* buf, int64_t n, float lr, float momentum, float weight_decay, int32_t nesterov,
                    int32_t first_step, const double* grad_sumsq, float max_norm, const float* amp_state, dy_stream_t stream);
int32_t dy_adamw_step(float* p, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                      float weight_decay, int32_t step, const double* grad_sumsq, float max_norm, const float* amp_state, dy_stream_t stream);
int32_t dy_amp_update(float* amp_state, const double* grad_sumsq, float growth_factor, float backoff_factor, int32_t growth_interval, dy_stream_t stream);
int32_t dy_ema_update(float* ema, const float* p, int64_t n, float decay, dy_stream_t stream);
int32_t dy_grad_sink_flush(const int64_t* entries, int32_t n_entries, float* grad, float* sink, dy_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DYOLO_H_ */
