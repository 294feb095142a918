// Kernel-side argument block shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_gemm_glds.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "common_hip.h"

namespace DY_NS {

struct ConvArgs {
  const void* x;
  const void* x2;
  const void* w;
  const float* bias;
  const void* res;
  void* y;
  int H, W, Cin, ldx, ldx2, split;  // split: channels [0,split) come from x, the rest from x2
  int HB, WB;                        // buffer dims of x (H/2, W/2 when up2x)
  int Ho, Wo, Cout, ldy, ldres;
  int ks, stride, pad;
  int Kpad, M, HoWo;
  int act, up2x;
  int tilesN, nblk;
  int vec_store;
  // conv_gemm_glds.hip, r03: buffer-addressed staging (lane-constant offsets + scalar step offsets) when every view is below 4 GiB and
  // the gather is linear in the tap (fast_addr); xb / x2b / wb = bytes addressable from x / x2 / w
  unsigned xb, x2b, wb;
  int fast_addr;
  // conv_gemm_glds.hip, r03: up2x == 2 (input gradient of a stride-2 3x3: a zero-dilated source) tiles the OUTPUT by parity class
  // (ho & 1, wo & 1): M' = 4 classes x Mq = N HB WB pixels each, a tile holds ONE class, so only the taps that meet non-zero source
  // pixels are walked (1, 2, 2 or 4 of the 9) instead of multiplying zeros three times out of four.  dil_cls: 1 when on;
  // dWB / dHW: exact divisions by WB and HB * WB for the row decode
  int dil_cls, Mq, tilesPerClass;
  FastDiv dWB, dHW;
  const float* wscale;  // DY_FP8: per-output-channel dequantisation multiplier (act_scale * weight scale), else nullptr
  float act_scale;      // DY_FP8: real value of one activation quantum (x, residual, y); 1 otherwise
  // optional (training forward in front of a train-mode BatchNorm, dy_conv_desc.bn_stats): a dy_bn_train_fwd workspace; a kernel with
  // a statistics epilogue stores per-channel sum / sum of squares of its STORED outputs in slot 1 + i of its i-th row block, zeroes
  // the totals (slot 0) and reports the slot count through note_stats(); the others ignore it
  double* stats;
  int stats_atomic;  // conv_gemm_glds.hip: more row blocks than slots -- block i ADDS into slot 1 + i % kStatSlots (the host zeroed them)
};


// conv_gemm_glds.hip: LDS-DMA staged 128 x {64,128} tile.  Returns 1 when the shape is not one it is built for
// (the caller then runs the generic kernel), else the launch status.
int conv_gemm_glds_try(const ConvArgs& a, int dtype, bool out_f32, hipStream_t st);

// conv3x3_vgemm.hip: 3x3 stride-1, Cin >= 128, Cout % 128 == 0 over the virtual flat pixel index.  Same return convention.
int conv3x3_vgemm_try(const ConvArgs& a, int dtype, bool out_f32, hipStream_t st);

}  // namespace DY_NS
