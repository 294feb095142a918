// Detect decode: DFL softmax-expectation + dist2bbox (xywh) * stride + sigmoid(cls),
// all levels in one launch, fp32 throughout.
// Reference: nn/modules/head.py:100-131 (_inference), block.py:58-76 (DFL),
// utils/tal.py:333-357 (make_anchors grid_cell_offset 0.5, dist2bbox).
//
// One lane per (image, anchor).  Reads: the anchor's 4*reg_max + nc logits, contiguous in the
// NHWC head buffer (float4 loads).  Writes: out[b][ch][a] — consecutive lanes are
// consecutive anchors, so every channel row is a coalesced store.
#include "common.cuh"

namespace dy {

struct DecodeArgs {
  const float* level[DY_MAX_LEVELS];
  int h[DY_MAX_LEVELS], w[DY_MAX_LEVELS], ld[DY_MAX_LEVELS], a0[DY_MAX_LEVELS + 1];
  float stride[DY_MAX_LEVELS];
  int n_levels, batch, nc, A;
  float* out;
};

template <int REG_MAX>
__global__ __launch_bounds__(256) void detect_decode_kernel(const DecodeArgs p) {
  const long long total = (long long)p.batch * p.A;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int b = (int)(idx / p.A);
    const int a = (int)(idx - (long long)b * p.A);
    int l = 0;
#pragma unroll
    for (int i = 1; i < DY_MAX_LEVELS; ++i)
      if (i < p.n_levels && a >= p.a0[i]) l = i;
    const int al = a - p.a0[l];
    const int hw = p.h[l] * p.w[l];
    const int gy = al / p.w[l], gx = al - gy * p.w[l];
    const float* src = p.level[l] + ((size_t)b * hw + al) * (size_t)p.ld[l];
    float dist[4];
#pragma unroll
    for (int side = 0; side < 4; ++side) {
      float v[REG_MAX];
#pragma unroll
      for (int i = 0; i < REG_MAX; i += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + side * REG_MAX + i);
        v[i] = t[0];
        v[i + 1] = t[1];
        v[i + 2] = t[2];
        v[i + 3] = t[3];
      }
      float mx = v[0];
#pragma unroll
      for (int i = 1; i < REG_MAX; ++i) mx = fmaxf(mx, v[i]);
      float den = 0.f, num = 0.f;
#pragma unroll
      for (int i = 0; i < REG_MAX; ++i) {
        const float e = expf(v[i] - mx);
        den += e;
        num += e * (float)i;
      }
      dist[side] = num / den;
    }
    const float ax = (float)gx + 0.5f, ay = (float)gy + 0.5f;
    const float x1 = ax - dist[0], y1 = ay - dist[1], x2 = ax + dist[2], y2 = ay + dist[3];
    const float s = p.stride[l];
    float* o = p.out + (size_t)b * (size_t)(4 + p.nc) * p.A + a;
    o[0] = (x1 + x2) * 0.5f * s;
    o[(size_t)p.A] = (y1 + y2) * 0.5f * s;
    o[(size_t)2 * p.A] = (x2 - x1) * s;
    o[(size_t)3 * p.A] = (y2 - y1) * s;
    const float* cls = src + 4 * REG_MAX;
    for (int c = 0; c < p.nc; ++c) o[(size_t)(4 + c) * p.A] = 1.0f / (1.0f + expf(-cls[c]));
  }
}

}  // namespace dy

using namespace dy;

extern "C" int32_t dy_detect_decode(const dy_decode_desc* d, dy_stream_t stream) {
  DY_REQUIRE(d && d->out, DY_ERR_INVALID_ARG, "dy_detect_decode: null descriptor/out");
  DY_REQUIRE(d->n_levels >= 1 && d->n_levels <= DY_MAX_LEVELS && d->batch > 0 && d->nc > 0, DY_ERR_INVALID_ARG,
             "dy_detect_decode: bad n_levels/batch/nc");
  DY_REQUIRE(d->reg_max == 16, DY_ERR_UNSUPPORTED, "dy_detect_decode: reg_max %d not built (only 16)", d->reg_max);
  DecodeArgs a{};
  int A = 0;
  for (int i = 0; i < d->n_levels; ++i) {
    DY_REQUIRE(d->level[i] && d->h[i] > 0 && d->w[i] > 0, DY_ERR_INVALID_ARG, "dy_detect_decode: level %d null/empty", i);
    DY_REQUIRE(d->ld[i] >= 4 * d->reg_max + d->nc && d->ld[i] % 4 == 0 && aligned16(d->level[i]), DY_ERR_INVALID_ARG,
               "dy_detect_decode: level %d pitch %d must be >= %d, a multiple of 4 floats, base 16B aligned", i, d->ld[i],
               4 * d->reg_max + d->nc);
    a.level[i] = d->level[i];
    a.h[i] = d->h[i];
    a.w[i] = d->w[i];
    a.ld[i] = d->ld[i];
    a.stride[i] = d->stride[i];
    a.a0[i] = A;
    A += d->h[i] * d->w[i];
  }
  a.a0[d->n_levels] = A;
  a.n_levels = d->n_levels;
  a.batch = d->batch;
  a.nc = d->nc;
  a.A = A;
  a.out = d->out;
  const long long total = (long long)d->batch * A;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL((detect_decode_kernel<16>), dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  return check_launch("detect_decode_kernel");
}
