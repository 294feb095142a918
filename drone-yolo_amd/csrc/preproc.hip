// Input side of the predictor for image sources, one kernel:
//   LetterBox (ultralytics/data/augment.py:1545-1608: aspect-preserving bilinear resize + constant 114 border)
//   -> BGR->RGB, HWC->CHW, uint8 -> float / 255 (engine/predictor.py:125-135).
// src: uint8 (n, h0, w0, 3) frames of ONE shape (BGR, as cv2 delivers them); dst: fp32 (n, 3, hn, wn), the layout the
// fused stem kernel consumes.  The resize follows OpenCV's 8-bit INTER_LINEAR arithmetic exactly as restated in
// oracle/letterbox_oracle.py (11-bit fixed-point coefficients, source coordinate (d + 0.5) * scale - 0.5 clamped at the
// borders, vertical pass (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2), so the uint8 image equals the oracle's bit
// for bit.  One thread per output pixel: three planes written with lane-contiguous stores.
#include "common_hip.h"

namespace dy {

struct LbArgs {
  const uint8_t* src;
  float* dst;
  int n, h0, w0, new_w, new_h, top, left, hn, wn, swap_rb;
  float scale, pad;
  double sx, sy;  // h0/new_h, w0/new_w
};

__device__ __forceinline__ void axis_coeff(int d, double scale, int src, int* s0, int* s1, int* a0, int* a1) {
  // unfused double arithmetic so that the result equals numpy's (oracle): (d + 0.5) * scale - 0.5
  double f = __dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);
  int s = (int)floor(f);
  f = f - (double)s;
  if (s < 0) f = 0.0, s = 0;
  if (s >= src - 1) f = 0.0, s = src - 1;
  const float ff = (float)f;
  *a0 = (int)rintf(__fmul_rn(__fsub_rn(1.0f, ff), 2048.0f));
  *a1 = (int)rintf(__fmul_rn(ff, 2048.0f));
  *s0 = s;
  *s1 = s + 1 < src ? s + 1 : src - 1;
}

__global__ __launch_bounds__(256) void letterbox_kernel(const LbArgs p) {
  const long long total = (long long)p.n * p.hn * p.wn;
  const size_t plane = (size_t)p.hn * p.wn;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % p.wn);
    long long t = i / p.wn;
    const int y = (int)(t % p.hn);
    const int img = (int)(t / p.hn);
    float v[3] = {p.pad, p.pad, p.pad};
    const int ry = y - p.top, rx = x - p.left;
    if ((unsigned)ry < (unsigned)p.new_h && (unsigned)rx < (unsigned)p.new_w) {
      const uint8_t* s = p.src + (size_t)img * p.h0 * p.w0 * 3;
      if (p.new_h == p.h0 && p.new_w == p.w0) {
        const uint8_t* q = s + ((size_t)ry * p.w0 + rx) * 3;
        v[0] = (float)q[0], v[1] = (float)q[1], v[2] = (float)q[2];
      } else {
        int x0, x1, ax0, ax1, y0, y1, by0, by1;
        axis_coeff(rx, p.sx, p.w0, &x0, &x1, &ax0, &ax1);
        axis_coeff(ry, p.sy, p.h0, &y0, &y1, &by0, &by1);
        const uint8_t* r0 = s + (size_t)y0 * p.w0 * 3;
        const uint8_t* r1 = s + (size_t)y1 * p.w0 * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int h0v = (int)r0[x0 * 3 + c] * ax0 + (int)r0[x1 * 3 + c] * ax1;
          const int h1v = (int)r1[x0 * 3 + c] * ax0 + (int)r1[x1 * 3 + c] * ax1;
          int o = (((by0 * (h0v >> 4)) >> 16) + ((by1 * (h1v >> 4)) >> 16) + 2) >> 2;
          o = o < 0 ? 0 : (o > 255 ? 255 : o);
          v[c] = (float)o;
        }
      }
    }
    float* d = p.dst + (size_t)img * 3 * plane + (size_t)y * p.wn + x;
    // dst channel c takes source channel (swap_rb ? 2 - c : c); x / 255 as the reference's `im /= 255`
    d[0] = (p.swap_rb ? v[2] : v[0]) / 255.0f;
    d[plane] = v[1] / 255.0f;
    d[2 * plane] = (p.swap_rb ? v[0] : v[2]) / 255.0f;
  }
}

}  // namespace dy

using namespace dy;

extern "C" int32_t dy_letterbox_u8_to_nchw_f32(const uint8_t* src, float* dst, int32_t n, int32_t h0, int32_t w0, int32_t new_w, int32_t new_h,
                                               int32_t top, int32_t left, int32_t hn, int32_t wn, int32_t swap_rb, float pad_value,
                                               dy_stream_t stream) {
  DY_REQUIRE(src && dst && n > 0 && h0 > 0 && w0 > 0 && new_w > 0 && new_h > 0 && hn > 0 && wn > 0, DY_ERR_INVALID_ARG,
             "dy_letterbox_u8_to_nchw_f32: bad arguments");
  DY_REQUIRE(top >= 0 && left >= 0 && top + new_h <= hn && left + new_w <= wn, DY_ERR_INVALID_ARG,
             "dy_letterbox_u8_to_nchw_f32: the resized image (%dx%d at %d,%d) does not fit the %dx%d output", new_w, new_h, left, top, wn, hn);
  LbArgs a{};
  a.src = src, a.dst = dst, a.n = n, a.h0 = h0, a.w0 = w0, a.new_w = new_w, a.new_h = new_h, a.top = top, a.left = left, a.hn = hn, a.wn = wn;
  a.swap_rb = swap_rb, a.pad = pad_value;
  a.sx = (double)w0 / (double)new_w;
  a.sy = (double)h0 / (double)new_h;
  const long long total = (long long)n * hn * wn;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(letterbox_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  return check_launch("dy_letterbox_u8_to_nchw_f32");
}

// ---- tiled inference on large frames ---------------------------------------------
// The reference delegates slicing to third-party packages that are not vendored (mix6.py:84-89 `sv.InferenceSlicer`,
// examples/YOLOv8-SAHI-Inference-Video/yolov8_sahi.py:50-55): tiles of a fixed size with a fractional overlap, one
// inference per tile, detections shifted back and merged by class-aware NMS.  This build defines it:
//   dy_tiles_u8_to_nchw_f32: K crops (th x tw at offsets[k] = (y, x)) of ONE uint8 HWC frame -> fp32 (K, 3, th, tw) / 255,
//                            channel order swapped when asked; pixels beyond the frame edge take pad_value;
//   dy_rows_to_pred:         per-tile NMS rows (K, max_det, 6) + counts -> one (1, 4+nc, K*max_det) prediction tensor in
//                            frame coordinates (xywh + the row's score in its class channel) for a final dy_nms.
namespace dy {

__global__ __launch_bounds__(256) void tiles_kernel(const uint8_t* __restrict__ src, const int* __restrict__ offs, float* __restrict__ dst, int k, int hf, int wf,
                                                    int th, int tw, int swap_rb, float pad) {
  const long long total = (long long)k * th * tw;
  const size_t plane = (size_t)th * tw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % tw);
    long long t = i / tw;
    const int y = (int)(t % th);
    const int tile = (int)(t / th);
    const int sy = offs[2 * tile] + y, sx = offs[2 * tile + 1] + x;
    float v[3] = {pad, pad, pad};
    if ((unsigned)sy < (unsigned)hf && (unsigned)sx < (unsigned)wf) {
      const uint8_t* q = src + ((size_t)sy * wf + sx) * 3;
      v[0] = (float)q[0], v[1] = (float)q[1], v[2] = (float)q[2];
    }
    float* d = dst + (size_t)tile * 3 * plane + (size_t)y * tw + x;
    d[0] = (swap_rb ? v[2] : v[0]) / 255.0f;
    d[plane] = v[1] / 255.0f;
    d[2 * plane] = (swap_rb ? v[0] : v[2]) / 255.0f;
  }
}

__global__ __launch_bounds__(256) void rows_to_pred_kernel(const float* __restrict__ rows, const int* __restrict__ counts, const int* __restrict__ offs,
                                                           float* __restrict__ pred, int k, int max_det, int nc) {
  const int A = k * max_det;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < A; j += gridDim.x * 256) {
    const int tile = j / max_det, r = j - tile * max_det;
    const bool valid = r < counts[tile];
    const float* q = rows + (size_t)j * 6;
    float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, sc = 0.f;
    int c = 0;
    if (valid) {
      const float oy = (float)offs[2 * tile], ox = (float)offs[2 * tile + 1];
      x1 = q[0] + ox, y1 = q[1] + oy, x2 = q[2] + ox, y2 = q[3] + oy, sc = q[4], c = (int)q[5];
    }
    pred[j] = (x1 + x2) * 0.5f;
    pred[(size_t)A + j] = (y1 + y2) * 0.5f;
    pred[(size_t)2 * A + j] = x2 - x1;
    pred[(size_t)3 * A + j] = y2 - y1;
    for (int cc = 0; cc < nc; ++cc) pred[(size_t)(4 + cc) * A + j] = (valid && cc == c) ? sc : 0.f;
  }
}

}  // namespace dy

extern "C" int32_t dy_tiles_u8_to_nchw_f32(const uint8_t* frame, const int32_t* offsets_yx, float* dst, int32_t k, int32_t hf, int32_t wf, int32_t th, int32_t tw,
                                           int32_t swap_rb, float pad_value, dy_stream_t stream) {
  DY_REQUIRE(frame && offsets_yx && dst && k > 0 && hf > 0 && wf > 0 && th > 0 && tw > 0, DY_ERR_INVALID_ARG, "dy_tiles_u8_to_nchw_f32: bad arguments");
  const long long total = (long long)k * th * tw;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dy::tiles_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), frame, offsets_yx, dst, k, hf, wf, th, tw, swap_rb,
                     pad_value);
  return dy::check_launch("dy_tiles_u8_to_nchw_f32");
}

extern "C" int32_t dy_rows_to_pred(const float* rows, const int32_t* counts, const int32_t* offsets_yx, float* pred, int32_t k, int32_t max_det, int32_t nc,
                                   dy_stream_t stream) {
  DY_REQUIRE(rows && counts && offsets_yx && pred && k > 0 && max_det > 0 && nc > 0, DY_ERR_INVALID_ARG, "dy_rows_to_pred: bad arguments");
  const int A = k * max_det;
  hipLaunchKernelGGL(dy::rows_to_pred_kernel, dim3((unsigned)((A + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), rows, counts, offsets_yx, pred, k,
                     max_det, nc);
  return dy::check_launch("dy_rows_to_pred");
}

// ---- multi_scale of DetectionTrainer.preprocess_batch (models/yolo/detect/train.py:60-73) -------------------------------------------
// batch["img"].float() / 255 followed by nn.functional.interpolate(imgs, size=ns, mode="bilinear", align_corners=False), one pass over a
// uint8 NCHW batch: source coordinate (d + 0.5) * (in / out) - 0.5 clamped at 0, neighbours i and min(i + 1, in - 1), the four taps
// divided by 255 first (the reference scales before it resizes) and blended horizontally then vertically in fp32 as torch's CPU kernel does.
namespace dy {
__global__ __launch_bounds__(256) void resize_bilinear_u8_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int planes, int h, int w, int ho, int wo,
                                                                 float sy, float sx) {
  const long long total = (long long)planes * ho * wo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % wo);
    long long t = i / wo;
    const int y = (int)(t % ho);
    const int pl = (int)(t / ho);
    // (one rounding, as torch's own kernels compute scale * (dst + 0.5) - 0.5: the host build contracts it into an fma, and a coordinate
    // that differs in its last bit moves the blend weight by 6e-5 at 640 pixels)
    float fy = __fmaf_rn((float)y + 0.5f, sy, -0.5f), fx = __fmaf_rn((float)x + 0.5f, sx, -0.5f);
    fy = fy < 0.f ? 0.f : fy, fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const uint8_t* s = src + (size_t)pl * h * w;
    const float v00 = (float)s[(size_t)y0 * w + x0] / 255.0f, v01 = (float)s[(size_t)y0 * w + x1] / 255.0f;
    const float v10 = (float)s[(size_t)y1 * w + x0] / 255.0f, v11 = (float)s[(size_t)y1 * w + x1] / 255.0f;
    dst[i] = __fadd_rn(__fmul_rn(hy, __fadd_rn(__fmul_rn(hx, v00), __fmul_rn(lx, v01))), __fmul_rn(ly, __fadd_rn(__fmul_rn(hx, v10), __fmul_rn(lx, v11))));
  }
}
}  // namespace dy

// scale_img (utils/torch_utils.py:436-445) for test-time augmentation (DetectionModel._predict_augment, nn/tasks.py:347-383): the fp32 NCHW batch —
// read mirrored left-right when asked — resized bilinearly to (hs, ws) with the rule above and padded right / bottom with `pad` up to (ho, wo).
namespace dy {
__global__ __launch_bounds__(256) void scale_img_kernel(const float* __restrict__ src, float* __restrict__ dst, int planes, int h, int w, int hs, int ws, int ho, int wo,
                                                        float sy, float sx, int flip_lr, float pad) {
  const long long total = (long long)planes * ho * wo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % wo);
    long long t = i / wo;
    const int y = (int)(t % ho);
    const int pl = (int)(t / ho);
    if (y >= hs || x >= ws) {
      dst[i] = pad;
      continue;
    }
    float fy = __fmaf_rn((float)y + 0.5f, sy, -0.5f), fx = __fmaf_rn((float)x + 0.5f, sx, -0.5f);
    fy = fy < 0.f ? 0.f : fy, fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* s = src + (size_t)pl * h * w;
    const int c0 = flip_lr ? w - 1 - x0 : x0, c1 = flip_lr ? w - 1 - x1 : x1;  // column k of the mirrored image is column w - 1 - k of the source
    const float v00 = s[(size_t)y0 * w + c0], v01 = s[(size_t)y0 * w + c1], v10 = s[(size_t)y1 * w + c0], v11 = s[(size_t)y1 * w + c1];
    dst[i] = __fadd_rn(__fmul_rn(hy, __fadd_rn(__fmul_rn(hx, v00), __fmul_rn(lx, v01))), __fmul_rn(ly, __fadd_rn(__fmul_rn(hx, v10), __fmul_rn(lx, v11))));
  }
}
}  // namespace dy

extern "C" int32_t dy_scale_img_nchw_f32(const float* src, float* dst, int32_t n, int32_t c, int32_t h, int32_t w, int32_t hs, int32_t ws, int32_t ho, int32_t wo,
                                         int32_t flip_lr, float pad, dy_stream_t stream) {
  DY_REQUIRE(src && dst && n > 0 && c > 0 && h > 0 && w > 0 && hs > 0 && ws > 0 && ho >= hs && wo >= ws, DY_ERR_INVALID_ARG, "dy_scale_img_nchw_f32: bad arguments");
  const long long total = (long long)n * c * ho * wo;
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(dy::scale_img_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, n * c, h, w, hs, ws, ho, wo,
                     (float)h / (float)hs, (float)w / (float)ws, flip_lr ? 1 : 0, pad);
  return dy::check_launch("dy_scale_img_nchw_f32");
}

extern "C" int32_t dy_resize_bilinear_u8_nchw_f32(const uint8_t* src, float* dst, int32_t n, int32_t c, int32_t h, int32_t w, int32_t ho, int32_t wo, dy_stream_t stream) {
  DY_REQUIRE(src && dst && n > 0 && c > 0 && h > 0 && w > 0 && ho > 0 && wo > 0, DY_ERR_INVALID_ARG, "dy_resize_bilinear_u8_nchw_f32: bad arguments");
  const long long total = (long long)n * c * ho * wo;
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(dy::resize_bilinear_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, n * c, h, w, ho, wo,
                     (float)h / (float)ho, (float)w / (float)wo);
  return dy::check_launch("dy_resize_bilinear_u8_nchw_f32");
}
