// Detect decode: DFL softmax-expectation + dist2bbox (xywh) * stride + sigmoid(cls), all levels in one
// launch, fp32 throughout, with the NMS candidate filter optionally fused in.
// Reference: nn/modules/head.py:100-131 (_inference), block.py:58-76 (DFL),
// utils/tal.py:333-357 (make_anchors grid_cell_offset 0.5, dist2bbox); filter: utils/ops.py:250,290-295.
//
// A single-wave workgroup owns 64 consecutive anchors of one (image, level).  Their head rows
// (4*reg_max + nc logits each, pitch ld) are one contiguous span of the NHWC head buffer, so it is
// copied to LDS with lane-linear 16-byte loads (every fetched byte is used once; a lane-per-row
// float4 read pattern fetched ~4x the bytes) and each lane then reads its own row from LDS.
// Writes: out[b][ch][a] — consecutive lanes are consecutive anchors, every channel row a coalesced store.
#include "common_hip.h"
#include "nms_ws.h"

namespace dy {

constexpr int kDecTile = 64;  // one wave per tile: 64 rows x pitch floats of LDS (19 KB at pitch 76), 8 tiles resident per CU

struct DecodeArgs {
  const float* level[DY_MAX_LEVELS];
  int h[DY_MAX_LEVELS], w[DY_MAX_LEVELS], ld[DY_MAX_LEVELS], a0[DY_MAX_LEVELS + 1], t0[DY_MAX_LEVELS + 1];
  float stride[DY_MAX_LEVELS];
  int n_levels, batch, nc, A, tilesPerImg;
  float* out;
  // fused NMS filter (optional)
  int* counts;
  unsigned long long* keys;
  unsigned short* cls;
  int P;
  float conf;
  const uint8_t* cmask;
};

template <int REG_MAX>
__global__ __launch_bounds__(64) void detect_decode_kernel(const DecodeArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  float* rows = reinterpret_cast<float*>(dyn_smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int b = blockIdx.x / p.tilesPerImg;
  const int tr = blockIdx.x - b * p.tilesPerImg;
  int l = 0;
#pragma unroll
  for (int i = 1; i < DY_MAX_LEVELS; ++i)
    if (i < p.n_levels && tr >= p.t0[i]) l = i;
  const int hw = p.h[l] * p.w[l];
  const int al0 = (tr - p.t0[l]) * kDecTile;           // first anchor of the tile inside the level
  const int na = (hw - al0) < kDecTile ? (hw - al0) : kDecTile;
  const int ld = p.ld[l];
  const float* src = p.level[l] + ((size_t)b * hw + al0) * (size_t)ld;
  const int n4 = na * ld / 4;  // ld % 4 == 0
  for (int i = tid; i < n4; i += kDecTile) reinterpret_cast<f32x4*>(rows)[i] = reinterpret_cast<const f32x4*>(src)[i];
  __syncthreads();  // single wave: orders the LDS writes before the row reads

  const bool valid = tid < na;
  const int al = al0 + tid;
  const int a = p.a0[l] + al;
  float best = 0.f;
  int bj = 0;
  if (valid) {
    const float* r = rows + tid * ld;
    const int gy = al / p.w[l], gx = al - gy * p.w[l];
    float dist[4];
#pragma unroll
    for (int side = 0; side < 4; ++side) {
      float v[REG_MAX];
#pragma unroll
      for (int i = 0; i < REG_MAX; i += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(r + side * REG_MAX + i);
        v[i] = t[0], v[i + 1] = t[1], v[i + 2] = t[2], v[i + 3] = t[3];
      }
      float mx = v[0];
#pragma unroll
      for (int i = 1; i < REG_MAX; ++i) mx = fmaxf(mx, v[i]);
      float den = 0.f, num = 0.f;
#pragma unroll
      for (int i = 0; i < REG_MAX; ++i) {
        const float e = __builtin_amdgcn_exp2f((v[i] - mx) * 1.4426950408889634f);
        den += e;
        num += e * (float)i;
      }
      dist[side] = num * __builtin_amdgcn_rcpf(den);
    }
    const float ax = (float)gx + 0.5f, ay = (float)gy + 0.5f;
    const float x1 = ax - dist[0], y1 = ay - dist[1], x2 = ax + dist[2], y2 = ay + dist[3];
    const float s = p.stride[l];
    float* o = p.out + (size_t)b * (size_t)(4 + p.nc) * p.A + a;
    o[0] = (x1 + x2) * 0.5f * s;
    o[(size_t)p.A] = (y1 + y2) * 0.5f * s;
    o[(size_t)2 * p.A] = (x2 - x1) * s;
    o[(size_t)3 * p.A] = (y2 - y1) * s;
    const float* cl = r + 4 * REG_MAX;
    for (int c = 0; c < p.nc; ++c) {
      const float pr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cl[c] * -1.4426950408889634f));
      o[(size_t)(4 + c) * p.A] = pr;
      if (c == 0 || pr > best) {  // first arg-max, as cls.max(1) (ops.py:290)
        best = pr;
        bj = c;
      }
    }
  }
  if (p.keys != nullptr) {  // fused candidate filter: one image per workgroup, so one ballot round per wave
    bool pass = valid && best > p.conf;
    if (pass && p.cmask) pass = p.cmask[bj] != 0;
    const unsigned long long m = __ballot(pass);
    if (m != 0ull) {
      const int leader = __ffsll((long long)m) - 1;
      int base = 0;
      if (lane == leader) base = atomicAdd(p.counts + b, __popcll(m));
      base = __shfl(base, leader);
      if (pass) {
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        p.keys[(size_t)b * p.P + pos] = ((unsigned long long)(~__float_as_uint(best)) << 32) | (unsigned long long)(unsigned)a;
        p.cls[(size_t)b * p.A + a] = (unsigned short)bj;
      }
    }
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
  int A = 0, T = 0, ldmax = 0;
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
    a.t0[i] = T;
    A += d->h[i] * d->w[i];
    T += (d->h[i] * d->w[i] + kDecTile - 1) / kDecTile;
    if (d->ld[i] > ldmax) ldmax = d->ld[i];
  }
  a.a0[d->n_levels] = A;
  a.t0[d->n_levels] = T;
  a.n_levels = d->n_levels;
  a.batch = d->batch;
  a.nc = d->nc;
  a.A = A;
  a.tilesPerImg = T;
  a.out = d->out;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->nms_workspace) {
    DY_REQUIRE(d->nc <= 65535, DY_ERR_UNSUPPORTED, "dy_detect_decode: nc %d > 65535", d->nc);
    DY_REQUIRE(aligned16(d->nms_workspace) && d->nms_workspace_bytes >= (int64_t)nms_ws_bytes(d->batch, A), DY_ERR_WORKSPACE,
               "dy_detect_decode: nms_workspace too small or misaligned (need %lld bytes)", (long long)nms_ws_bytes(d->batch, A));
    const NmsWs w = nms_ws_layout(d->nms_workspace, d->batch, A);
    a.counts = w.counts;
    a.keys = w.keys;
    a.cls = w.cls;
    a.P = w.P;
    a.conf = d->conf_thres;
    a.cmask = d->classes_mask;
    zero_async(w.counts, (size_t)d->batch * 4, st);
  }
  const size_t smem = (size_t)kDecTile * ldmax * 4;
  DY_REQUIRE(smem <= 160 * 1024, DY_ERR_UNSUPPORTED, "dy_detect_decode: head pitch %d too large for the LDS tile", ldmax);
  static const hipError_t once = hipFuncSetAttribute((const void*)detect_decode_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)once;
  hipLaunchKernelGGL((detect_decode_kernel<16>), dim3((unsigned)(d->batch * T)), dim3(kDecTile), smem, st, a);
  return check_launch("detect_decode_kernel");
}
