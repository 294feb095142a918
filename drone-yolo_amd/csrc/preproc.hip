// Input side of the predictor for image sources, one kernel:
//   LetterBox (ultralytics/data/augment.py:1545-1608: aspect-preserving bilinear resize + constant 114 border)
//   -> BGR->RGB, HWC->CHW, uint8 -> float / 255 (engine/predictor.py:125-135).
// src: uint8 (n, h0, w0, 3) frames of ONE shape (BGR, as cv2 delivers them); dst: fp32 (n, 3, hn, wn), the layout the
// fused stem kernel consumes.  The resize follows OpenCV's 8-bit INTER_LINEAR arithmetic exactly as restated in
// oracle/letterbox_oracle.py (11-bit fixed-point coefficients, source coordinate (d + 0.5) * scale - 0.5 clamped at the
// borders, vertical pass (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2), so the uint8 image equals the oracle's bit
// for bit.  One thread per output pixel: three planes written with lane-contiguous stores.
#include "common.cuh"

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
