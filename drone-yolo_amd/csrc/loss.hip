// v8DetectionLoss forward on the device: TaskAlignedAssigner + BCE / CIoU / DFL losses.
// Reference: utils/loss.py:157-260 (v8DetectionLoss), :65-113 (DFLoss, BboxLoss); utils/tal.py:14-295
// (TaskAlignedAssigner, topk 10, alpha 0.5, beta 6), :360-363 (bbox2dist); utils/metrics.py:74-134 (CIoU).
//
// The reference materialises (B, n_max_boxes, A) tensors (A = 34,000 anchors: GBs at batch 64).  Here the
// assignment is sparse: a ground-truth box can only select anchors whose centre lies strictly inside it, and
// those are an axis-aligned range of grid cells per level, so
//   K1 loss_decode_kernel   per (image, anchor): DFL softmax-expectation -> predicted box in pixels (kept for the
//                           assigner), and the dense part of the BCE loss, sum_c softplus(logit_c) (target 0);
//   K2 tal_pick_kernel      one wave per ground-truth box: enumerate its in-box anchors, metric = s^alpha * CIoU^beta,
//                           keep the top-k by wave-level arg-max rounds; every pick bumps the anchor's claim count;
//   K3 tal_resolve_kernel   per pick: an anchor claimed once belongs to that box; claimed several times it goes to the
//                           box with the highest overlap among ALL boxes of the image (first index on ties);
//   K4 tal_gtmax_kernel     per foreground anchor: per-box maxima of metric and overlap (float atomicMax on bits);
//   K5 loss_fg_kernel       per foreground anchor: soft target t = metric * max_overlap / (max_metric + eps);
//                           sums of t, x_cls*t, (1-CIoU)*t, DFL*t;
//   K6 loss_final_kernel    loss_cls = (sum softplus - sum x_cls t)/tss etc., gains, total.
// Deviation (documented in DESIGN.md): picks with metric == 0 are never taken.  torch.topk may return such ties in
// unspecified order; they carry target score 0 and weight 0, so they do not change any loss term.
#include "common_hip.h"

namespace dy {

struct LossArgs {
  const float* level[DY_MAX_LEVELS];
  int h[DY_MAX_LEVELS], w[DY_MAX_LEVELS], ld[DY_MAX_LEVELS], a0[DY_MAX_LEVELS + 1];
  float stride[DY_MAX_LEVELS];
  int n_levels, batch, nc, A, gmax, topk;
  float alpha, beta, box_gain, cls_gain, dfl_gain;
  const float* gt;   // (batch, gmax, 5): cls, x1, y1, x2, y2 (pixels); padding rows have x1+y1+x2+y2 <= 0
  float* pbox;       // (batch, A, 4) predicted xyxy in pixels
  int* claims;       // (batch, A) number of boxes that picked the anchor
  int* owner;        // (batch, A) assigned gt index or -1
  int* picks;        // (batch, gmax, topk) anchor index or -1
  unsigned* gmax_al; // (batch, gmax) max metric over the box's foreground anchors (float bits)
  unsigned* gmax_ov; // (batch, gmax) max overlap
  double* acc;       // [0] sum softplus, [1] sum t, [2] sum x_cls*t, [3] sum (1-ciou)*t, [4] sum dfl*t, [5] n_fg
  float* out;        // [4] box, cls, dfl (after gains), total = sum * batch
  float* glevel[DY_MAX_LEVELS];  // optional: d total / d level[l], same NHWC geometry, pitch gld[l]
  int gld[DY_MAX_LEVELS];
};

constexpr int kRegMax = 16;

__device__ __forceinline__ const float* row_ptr(const LossArgs& p, int b, int a, int* lvl, int* gx, int* gy) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < DY_MAX_LEVELS; ++i)
    if (i < p.n_levels && a >= p.a0[i]) l = i;
  const int al = a - p.a0[l];
  *lvl = l;
  *gy = al / p.w[l];
  *gx = al - *gy * p.w[l];
  return p.level[l] + ((size_t)b * p.h[l] * p.w[l] + al) * (size_t)p.ld[l];
}

// CIoU of two xyxy boxes, the arithmetic of utils/metrics.py:100-129 with eps = 1e-7.
__device__ __forceinline__ float ciou(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2, float by2) {
  const float eps = 1e-7f;
  const float w1 = ax2 - ax1, h1 = ay2 - ay1 + eps, w2 = bx2 - bx1, h2 = by2 - by1 + eps;
  const float inter = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f) * fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
  const float uni = w1 * h1 + w2 * h2 - inter + eps;
  const float iou = inter / uni;
  const float cw = fmaxf(ax2, bx2) - fminf(ax1, bx1), ch = fmaxf(ay2, by2) - fminf(ay1, by1);
  const float c2 = cw * cw + ch * ch + eps;
  const float dx = bx1 + bx2 - ax1 - ax2, dy_ = by1 + by2 - ay1 - ay2;
  const float rho2 = (dx * dx + dy_ * dy_) / 4.f;
  const float dat = atanf(w2 / h2) - atanf(w1 / h1);
  const float v = 0.4052847345693511f * dat * dat;  // 4 / pi^2
  const float alpha = v / (v - iou + (1.f + eps));
  return iou - (rho2 / c2 + v * alpha);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;
}

// Sum of v over the 256-thread workgroup, in thread 0 (callers add it to a global accumulator: ONE atomic per workgroup and value.
// The first form ended every WAVE with its own double atomics on the same few addresses -- 16 k waves x 5 values in loss_fg_kernel --
// and those serialise: 505 us for a kernel whose work is 3 % of the anchors).
__device__ __forceinline__ double block_sum(double v, double* red /* [4] in LDS */) {
  v = wave_sum(v);
  __syncthreads();  // red may still be read from the previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---- K1 -------------------------------------------------------------------------------------------------------
// Thread = (anchor, side): the side's 16 bins are ONE 64-byte run, read as four 16-byte loads (one thread per anchor read its 296-byte
// row a float at a time, twice: 374 us at B = 64); lane `side` writes its own corner; side 0 also sums the class softplus terms.
__global__ __launch_bounds__(256) void loss_decode_kernel(const LossArgs p) {
  __shared__ double red[4];
  double sp = 0.0;
  for (int l = 0; l < p.n_levels; ++l) {  // level by level: per-level arguments are scalars (see loss_grad_dense_kernel)
    const unsigned w = (unsigned)p.w[l], hw = (unsigned)p.h[l] * w, total = (unsigned)p.batch * hw * 4u;
    const float* __restrict__ lv = p.level[l];
    const size_t ld = (size_t)p.ld[l];
    const unsigned a0 = (unsigned)p.a0[l];
    const float st = p.stride[l];
    for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) {
      const unsigned row = t >> 2;  // b * hw + anchor of the level
      const int sd = (int)(t & 3);
      const unsigned b = row / hw, al = row - b * hw;
      const unsigned gy = al / w, gx = al - gy * w;
      const size_t idx = (size_t)b * p.A + a0 + al;
      const float* r = lv + (size_t)row * ld;
      float q[kRegMax];
      const float* src = r + sd * kRegMax;
      if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
#pragma unroll
        for (int i = 0; i < kRegMax; i += 4) {
          const float4 v = *reinterpret_cast<const float4*>(src + i);
          q[i] = v.x, q[i + 1] = v.y, q[i + 2] = v.z, q[i + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int i = 0; i < kRegMax; ++i) q[i] = src[i];
      }
      float mx = q[0];
#pragma unroll
      for (int i = 1; i < kRegMax; ++i) mx = fmaxf(mx, q[i]);
      float den = 0.f, num = 0.f;
#pragma unroll
      for (int i = 0; i < kRegMax; ++i) {
        const float e = expf(q[i] - mx);
        den += e;
        num += e * (float)i;
      }
      const float d = num / den;
      const float ac = (float)((sd & 1) ? gy : gx) + 0.5f;
      p.pbox[idx * 4 + sd] = (sd < 2 ? ac - d : ac + d) * st;
      if (sd == 0) {
        const float* cl = r + 4 * kRegMax;
        float sm = 0.f;
        for (int c = 0; c < p.nc; ++c) {
          const float x = cl[c];
          sm += fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));  // BCEWithLogits against target 0
        }
        sp += (double)sm;
        p.claims[idx] = 0;
        p.owner[idx] = -1;
      }
    }
  }
  sp = block_sum(sp, red);
  if (threadIdx.x == 0 && sp != 0.0) atomicAdd(p.acc + 0, sp);
}

// metric and overlap of (gt box g of image b, anchor a); 0 when the anchor centre is not strictly inside the box
__device__ __forceinline__ void pair_metric(const LossArgs& p, int b, const float* g5, int a, float* metric, float* overlap) {
  int l, gx, gy;
  const float* r = row_ptr(p, b, a, &l, &gx, &gy);
  const float cx = ((float)gx + 0.5f) * p.stride[l], cy = ((float)gy + 0.5f) * p.stride[l];
  const float dmin = fminf(fminf(cx - g5[1], cy - g5[2]), fminf(g5[3] - cx, g5[4] - cy));
  *metric = 0.f;
  *overlap = 0.f;
  if (!(dmin > 1e-9f)) return;  // tal.py:262
  const float* pb = p.pbox + ((size_t)b * p.A + a) * 4;
  const float ov = fmaxf(ciou(g5[1], g5[2], g5[3], g5[4], pb[0], pb[1], pb[2], pb[3]), 0.f);  // tal.py:155
  const float x = r[4 * kRegMax + (int)g5[0]];
  const float sc = 1.0f / (1.0f + expf(-x));
  *overlap = ov;
  *metric = powf(sc, p.alpha) * powf(ov, p.beta);
}

// ---- K2: one wave per ground-truth box ----------------------------------------------------------------------
__global__ __launch_bounds__(64) void tal_pick_kernel(const LossArgs p) {
  const int bg = blockIdx.x, b = bg / p.gmax, lane = threadIdx.x;
  const float* g5 = p.gt + (size_t)bg * 5;
  int* picks = p.picks + (size_t)bg * p.topk;
  const bool valid = (g5[1] + g5[2] + g5[3] + g5[4]) > 0.f;  // mask_gt, loss.py:229
  for (int k = lane; k < p.topk; k += 64) picks[k] = -1;
  if (!valid) return;
  float last_m = 3.0e38f;
  int last_a = -1;  // picks proceed in (metric desc, anchor asc) order; ties resolved by anchor index
  for (int k = 0; k < p.topk; ++k) {
    float bm = 0.f;
    int ba = 0x7fffffff;
    for (int l = 0; l < p.n_levels; ++l) {
      const float st = p.stride[l];
      // cells whose centre (g+0.5)*st lies strictly inside (x1, x2): g > x1/st - 0.5 and g < x2/st - 0.5
      int gx0 = (int)floorf(g5[1] / st - 0.5f) + 1, gx1 = (int)ceilf(g5[3] / st - 0.5f) - 1;
      int gy0 = (int)floorf(g5[2] / st - 0.5f) + 1, gy1 = (int)ceilf(g5[4] / st - 0.5f) - 1;
      gx0 = gx0 < 0 ? 0 : gx0;
      gy0 = gy0 < 0 ? 0 : gy0;
      gx1 = gx1 >= p.w[l] ? p.w[l] - 1 : gx1;
      gy1 = gy1 >= p.h[l] ? p.h[l] - 1 : gy1;
      const int nx = gx1 - gx0 + 1, ny = gy1 - gy0 + 1;
      if (nx <= 0 || ny <= 0) continue;
      for (int i = lane; i < nx * ny; i += 64) {
        const int yy = gy0 + i / nx, xx = gx0 + i % nx;
        const int a = p.a0[l] + yy * p.w[l] + xx;
        float m, ov;
        pair_metric(p, b, g5, a, &m, &ov);
        // candidates strictly after the previous pick in (metric desc, anchor asc) order
        const bool after = (m < last_m) || (m == last_m && a > last_a);
        if (m > 0.f && after && (m > bm || (m == bm && a < ba))) {
          bm = m;
          ba = a;
        }
      }
    }
    // wave arg-max
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(bm, o);
      const int oa = __shfl_xor(ba, o);
      if (om > bm || (om == bm && oa < ba)) {
        bm = om;
        ba = oa;
      }
    }
    if (!(bm > 0.f)) break;  // no positive-metric candidate left
    if (lane == 0) {
      picks[k] = ba;
      atomicAdd(p.claims + (size_t)b * p.A + ba, 1);
    }
    last_m = bm;
    last_a = ba;
  }
}

// ---- K3: one thread per pick -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tal_resolve_kernel(const LossArgs p) {
  const long long total = (long long)p.batch * p.gmax * p.topk;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int a = p.picks[idx];
    if (a < 0) continue;
    const int bg = (int)(idx / p.topk), b = bg / p.gmax, g = bg - b * p.gmax;
    int own = g;
    if (p.claims[(size_t)b * p.A + a] > 1) {  // tal.py:281-288: arg-max of overlaps over all boxes, first index on ties
      float best = -1.f;
      for (int gg = 0; gg < p.gmax; ++gg) {
        const float* g5 = p.gt + ((size_t)b * p.gmax + gg) * 5;
        float m = 0.f, ov = 0.f;
        if ((g5[1] + g5[2] + g5[3] + g5[4]) > 0.f) pair_metric(p, b, g5, a, &m, &ov);
        if (ov > best) {
          best = ov;
          own = gg;
        }
      }
    }
    p.owner[(size_t)b * p.A + a] = own;  // every claimant of a shared anchor computes the same owner
  }
}

// ---- K4 / K5: one thread per (image, anchor), foreground only ----------------------------------------------
__global__ __launch_bounds__(256) void tal_gtmax_kernel(const LossArgs p) {
  const long long total = (long long)p.batch * p.A;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int g = p.owner[idx];
    if (g < 0) continue;
    const int b = (int)(idx / p.A), a = (int)(idx - (long long)b * p.A);
    float m, ov;
    pair_metric(p, b, p.gt + ((size_t)b * p.gmax + g) * 5, a, &m, &ov);
    atomicMax(p.gmax_al + (size_t)b * p.gmax + g, __float_as_uint(m));  // non-negative floats order like their bits
    atomicMax(p.gmax_ov + (size_t)b * p.gmax + g, __float_as_uint(ov));
  }
}

__global__ __launch_bounds__(256) void loss_fg_kernel(const LossArgs p) {
  __shared__ double red[4];
  const long long total = (long long)p.batch * p.A;
  double s_t = 0.0, s_xt = 0.0, s_box = 0.0, s_dfl = 0.0, s_n = 0.0;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int g = p.owner[idx];
    if (g < 0) continue;
    const int b = (int)(idx / p.A), a = (int)(idx - (long long)b * p.A);
    const float* g5 = p.gt + ((size_t)b * p.gmax + g) * 5;
    float m, ov;
    pair_metric(p, b, g5, a, &m, &ov);
    const float pal = __uint_as_float(p.gmax_al[(size_t)b * p.gmax + g]), pov = __uint_as_float(p.gmax_ov[(size_t)b * p.gmax + g]);
    const float t = m * pov / (pal + 1e-9f);  // tal.py:111-116
    int l, gx, gy;
    const float* r = row_ptr(p, b, a, &l, &gx, &gy);
    const float st = p.stride[l];
    const float* pb = p.pbox + (size_t)idx * 4;
    // box loss in grid units (loss.py:252-255: target_bboxes /= stride; pred boxes are in grid units there)
    const float c = ciou(pb[0] / st, pb[1] / st, pb[2] / st, pb[3] / st, g5[1] / st, g5[2] / st, g5[3] / st, g5[4] / st);
    // DFL (loss.py:73-88): targets = ltrb distances clamped to [0, reg_max-1-0.01]
    const float ax = (float)gx + 0.5f, ay = (float)gy + 0.5f;
    const float tgt[4] = {ax - g5[1] / st, ay - g5[2] / st, g5[3] / st - ax, g5[4] / st - ay};
    float dfl = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float tv = fminf(fmaxf(tgt[s], 0.f), (float)(kRegMax - 1) - 0.01f);
      const int tl = (int)tv;
      const float wl = (float)(tl + 1) - tv, wr = 1.f - wl;
      const float* q = r + s * kRegMax;
      float mx = q[0];
      for (int i = 1; i < kRegMax; ++i) mx = fmaxf(mx, q[i]);
      float den = 0.f;
      for (int i = 0; i < kRegMax; ++i) den += expf(q[i] - mx);
      const float lse = mx + logf(den);
      dfl += (lse - q[tl]) * wl + (lse - q[tl + 1]) * wr;
    }
    dfl *= 0.25f;  // mean over the four sides
    const float x = r[4 * kRegMax + (int)g5[0]];
    s_t += (double)t;
    s_xt += (double)(x * t);
    s_box += (double)((1.f - c) * t);
    s_dfl += (double)(dfl * t);
    s_n += 1.0;
  }
  s_t = block_sum(s_t, red), s_xt = block_sum(s_xt, red), s_box = block_sum(s_box, red), s_dfl = block_sum(s_dfl, red), s_n = block_sum(s_n, red);
  if (threadIdx.x == 0 && s_n != 0.0) {
    atomicAdd(p.acc + 1, s_t);
    atomicAdd(p.acc + 2, s_xt);
    atomicAdd(p.acc + 3, s_box);
    atomicAdd(p.acc + 4, s_dfl);
    atomicAdd(p.acc + 5, s_n);
  }
}

// ---- K7: gradient of total = (box + cls + dfl) * batch w.r.t. the raw head outputs -----------------------------------
// What autograd gives the reference for loss.sum() * batch_size (loss.py:260, trainer.py:381-389): the assignment, the
// soft targets and target_scores_sum are constants (assigner under no_grad, tal.py:60); CIoU's alpha is a constant
// (metrics.py:127-128).  Forward-mode differentiation of CIoU w.r.t. the four predicted corners.
struct D4 {
  float v, g[4];
};
__device__ __forceinline__ D4 dconst(float c) { return D4{c, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 dvar(float c, int i) {
  D4 r = dconst(c);
  r.g[i] = 1.f;
  return r;
}
__device__ __forceinline__ D4 operator+(D4 a, D4 b) { return D4{a.v + b.v, {a.g[0] + b.g[0], a.g[1] + b.g[1], a.g[2] + b.g[2], a.g[3] + b.g[3]}}; }
__device__ __forceinline__ D4 operator-(D4 a, D4 b) { return D4{a.v - b.v, {a.g[0] - b.g[0], a.g[1] - b.g[1], a.g[2] - b.g[2], a.g[3] - b.g[3]}}; }
__device__ __forceinline__ D4 operator*(D4 a, D4 b) {
  return D4{a.v * b.v, {a.g[0] * b.v + a.v * b.g[0], a.g[1] * b.v + a.v * b.g[1], a.g[2] * b.v + a.v * b.g[2], a.g[3] * b.v + a.v * b.g[3]}};
}
__device__ __forceinline__ D4 operator/(D4 a, D4 b) {
  const float q = a.v / b.v, ib = 1.f / b.v;
  return D4{q, {(a.g[0] - q * b.g[0]) * ib, (a.g[1] - q * b.g[1]) * ib, (a.g[2] - q * b.g[2]) * ib, (a.g[3] - q * b.g[3]) * ib}};
}
__device__ __forceinline__ D4 dscale(D4 a, float c) { return D4{a.v * c, {a.g[0] * c, a.g[1] * c, a.g[2] * c, a.g[3] * c}}; }
__device__ __forceinline__ D4 dmaxc(D4 a, float c) { return a.v >= c ? a : dconst(c); }  // torch.maximum(a, const) / clamp(min=c)
__device__ __forceinline__ D4 dminc(D4 a, float c) { return a.v <= c ? a : dconst(c); }
__device__ __forceinline__ D4 datan(D4 a) {
  const float d = 1.f / (1.f + a.v * a.v);
  return D4{atanf(a.v), {a.g[0] * d, a.g[1] * d, a.g[2] * d, a.g[3] * d}};
}

// 1 - CIoU(pred, target) differentiated w.r.t. pred = (x1, y1, x2, y2); same arithmetic as ciou() above with box1 = pred.
__device__ __forceinline__ D4 ciou_dual(const float* pb, const float* tb) {
  const float eps = 1e-7f;
  const D4 x1 = dvar(pb[0], 0), y1 = dvar(pb[1], 1), x2 = dvar(pb[2], 2), y2 = dvar(pb[3], 3);
  const D4 w1 = x2 - x1, h1 = (y2 - y1) + dconst(eps);
  const float w2 = tb[2] - tb[0], h2 = tb[3] - tb[1] + eps;
  const D4 iw = dmaxc(dminc(x2, tb[2]) - dmaxc(x1, tb[0]), 0.f), ih = dmaxc(dminc(y2, tb[3]) - dmaxc(y1, tb[1]), 0.f);
  const D4 inter = iw * ih;
  const D4 uni = w1 * h1 + dconst(w2 * h2) - inter + dconst(eps);
  const D4 iou = inter / uni;
  const D4 cw = dmaxc(x2, tb[2]) - dminc(x1, tb[0]), ch = dmaxc(y2, tb[3]) - dminc(y1, tb[1]);
  const D4 c2 = cw * cw + ch * ch + dconst(eps);
  const D4 dx = dconst(tb[0] + tb[2]) - x1 - x2, dy_ = dconst(tb[1] + tb[3]) - y1 - y2;
  const D4 rho2 = dscale(dx * dx + dy_ * dy_, 0.25f);
  const D4 dat = dconst(atanf(w2 / h2)) - datan(w1 / h1);
  const D4 v = dscale(dat * dat, 0.4052847345693511f);
  const float alpha = v.v / (v.v - iou.v + (1.f + eps));  // constant under autograd (metrics.py:127)
  return iou - (rho2 / c2 + dscale(v, alpha));
}

// Two launches.  DENSE: thread = one 16-byte chunk of a BACKGROUND anchor's row of the gradient map (97 % of the rows): zeros in the
// 64 box bins, sigmoid(x) * gain in the class slots -- consecutive lanes write consecutive 16 bytes.  SPARSE: thread = anchor, foreground
// anchors only: box / DFL bins and the class slots with the soft target.  (One thread per anchor for everything wrote its 74 floats one at a
// time, each store instruction touching 64 different lines: 591 us at B = 64.)
// Rows whose pitch leaves fewer than 8 floats of alignment padding behind the 4 * reg_max + nc values are written WHOLE (zeros in the
// padding): a row with holes makes the L2 fetch every partly written line from HBM before it can write it back (389 -> 2xx us).
__global__ __launch_bounds__(256) void loss_grad_dense_kernel(const LossArgs p, unsigned chunks_values) {
  const double tss_d = p.acc[1] > 1.0 ? p.acc[1] : 1.0;
  const float inv = (float)((double)p.batch / tss_d);  // d total / d (sum of a loss term's numerator)
  // level by level: everything that depends on the level is then a SCALAR (indexing the per-level arrays of the kernel arguments with a
  // per-lane level makes every access a vector load from the argument segment, one dependent round trip each: 485 us for this kernel)
  for (int l = 0; l < p.n_levels; ++l) {
    const unsigned hw = (unsigned)(p.h[l] * p.w[l]);
    const float* __restrict__ lv = p.level[l];
    float* __restrict__ gl = p.glevel[l];
    const size_t ld = (size_t)p.ld[l], gld = (size_t)p.gld[l];
    const unsigned a0 = (unsigned)p.a0[l];
    const int nval = 4 * kRegMax + p.nc;
    const bool whole = gld % 4 == 0 && (int)gld - nval < 8;  // scalar
    const unsigned chunks = whole ? (unsigned)(gld / 4) : chunks_values;
    const int lim = whole ? (int)gld : nval;  // floats of a row this kernel writes
    const unsigned total = (unsigned)p.batch * hw * chunks;
    for (unsigned tt = blockIdx.x * 256u + threadIdx.x; tt < total; tt += gridDim.x * 256u) {
      const unsigned row = tt / chunks;  // b * hw + anchor of the level
      const int ch = (int)(tt - row * chunks);
      const unsigned b = row / hw, al = row - b * hw;
      if (p.owner[(size_t)b * p.A + a0 + al] >= 0) continue;  // the sparse launch writes that row
      float* go = gl + (size_t)row * gld;
      if (ch < kRegMax) {
        float* gs = go + ch * 4;
        if ((reinterpret_cast<uintptr_t>(gs) & 15) == 0) *reinterpret_cast<float4*>(gs) = float4{0.f, 0.f, 0.f, 0.f};
        else gs[0] = 0.f, gs[1] = 0.f, gs[2] = 0.f, gs[3] = 0.f;
      } else {
        const int c0 = (ch - kRegMax) * 4;
        const float* cl = lv + (size_t)row * ld + 4 * kRegMax;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = c0 + j < p.nc ? (1.0f / (1.0f + expf(-cl[c0 + j])) - 0.f) * p.cls_gain * inv : 0.f;
        float* gs = go + 4 * kRegMax + c0;
        if (4 * kRegMax + c0 + 4 <= lim && (reinterpret_cast<uintptr_t>(gs) & 15) == 0) {
          *reinterpret_cast<float4*>(gs) = float4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (4 * kRegMax + c0 + j < lim) gs[j] = v[j];
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void loss_grad_kernel(const LossArgs p) {
  const long long total = (long long)p.batch * p.A;
  const double tss_d = p.acc[1] > 1.0 ? p.acc[1] : 1.0;
  const float inv = (float)((double)p.batch / tss_d);  // d total / d (sum of a loss term's numerator)
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int g = p.owner[idx];
    if (g < 0) continue;  // background rows: loss_grad_dense_kernel
    const int b = (int)(idx / p.A), a = (int)(idx - (long long)b * p.A);
    int l, gx, gy;
    const float* r = row_ptr(p, b, a, &l, &gx, &gy);
    const int al = a - p.a0[l];
    float* go = p.glevel[l] + ((size_t)b * p.h[l] * p.w[l] + al) * (size_t)p.gld[l];
    const float* g5 = p.gt + ((size_t)b * p.gmax + g) * 5;
    float m, ov;
    pair_metric(p, b, g5, a, &m, &ov);
    const float pal = __uint_as_float(p.gmax_al[(size_t)b * p.gmax + g]), pov = __uint_as_float(p.gmax_ov[(size_t)b * p.gmax + g]);
    const float t = m * pov / (pal + 1e-9f);
    const int tc = (int)g5[0];
    // class logits: d BCE / dx = sigmoid(x) - target
    const float* cl = r + 4 * kRegMax;
    for (int c = 0; c < p.nc; ++c) {
      const float sg = 1.0f / (1.0f + expf(-cl[c]));
      go[4 * kRegMax + c] = (sg - (c == tc ? t : 0.f)) * p.cls_gain * inv;
    }
    // box bins of a foreground anchor
    const float st = p.stride[l];
    const float ax = (float)gx + 0.5f, ay = (float)gy + 0.5f;
    float pr[4][kRegMax], d[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float* q = r + s * kRegMax;
      float mx = q[0];
      for (int i = 1; i < kRegMax; ++i) mx = fmaxf(mx, q[i]);
      float den = 0.f;
      for (int i = 0; i < kRegMax; ++i) {
        pr[s][i] = expf(q[i] - mx);
        den += pr[s][i];
      }
      float num = 0.f;
      for (int i = 0; i < kRegMax; ++i) {
        pr[s][i] /= den;
        num += pr[s][i] * (float)i;
      }
      d[s] = num;
    }
    const float pb[4] = {ax - d[0], ay - d[1], ax + d[2], ay + d[3]};
    const float tb[4] = {g5[1] / st, g5[2] / st, g5[3] / st, g5[4] / st};
    const D4 c = ciou_dual(pb, tb);
    // d/d dist: x1 = ax - d0, y1 = ay - d1, x2 = ax + d2, y2 = ay + d3; loss term (1 - ciou) * t
    const float kb = -t * p.box_gain * inv;
    const float gd[4] = {-c.g[0] * kb, -c.g[1] * kb, c.g[2] * kb, c.g[3] * kb};
    const float tgt[4] = {ax - tb[0], ay - tb[1], tb[2] - ax, tb[3] - ay};
    const float kd = t * p.dfl_gain * inv * 0.25f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float tv = fminf(fmaxf(tgt[s], 0.f), (float)(kRegMax - 1) - 0.01f);
      const int tl = (int)tv;
      const float wl = (float)(tl + 1) - tv, wr = 1.f - wl;
      for (int i = 0; i < kRegMax; ++i) {
        const float onehot = (i == tl ? wl : 0.f) + (i == tl + 1 ? wr : 0.f);
        go[s * kRegMax + i] = gd[s] * pr[s][i] * ((float)i - d[s]) + kd * (pr[s][i] - onehot);
      }
    }
  }
}

__global__ void loss_final_kernel(const LossArgs p) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double tss = p.acc[1] > 1.0 ? p.acc[1] : 1.0;  // loss.py:247
  const double cls = (p.acc[0] - p.acc[2]) / tss;
  const double box = p.acc[5] > 0.0 ? p.acc[3] / tss : 0.0;
  const double dfl = p.acc[5] > 0.0 ? p.acc[4] / tss : 0.0;
  p.out[0] = (float)(box * p.box_gain);
  p.out[1] = (float)(cls * p.cls_gain);
  p.out[2] = (float)(dfl * p.dfl_gain);
  p.out[3] = (float)((box * p.box_gain + cls * p.cls_gain + dfl * p.dfl_gain) * p.batch);
}

static inline size_t lalign(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace dy

using namespace dy;

extern "C" int64_t dy_detection_loss_workspace_bytes(int32_t batch, int32_t anchors, int32_t gmax, int32_t topk) {
  if (batch <= 0 || anchors <= 0 || gmax < 0 || topk <= 0) return -1;
  const size_t ba = (size_t)batch * anchors, bg = (size_t)batch * (gmax > 0 ? gmax : 1);
  return (int64_t)(lalign(ba * 16) + 2 * lalign(ba * 4) + lalign(bg * topk * 4) + 2 * lalign(bg * 4) + lalign(6 * 8));
}

extern "C" int32_t dy_detection_loss(const dy_loss_desc* d, dy_stream_t stream) {
  DY_REQUIRE(d && d->out && d->workspace, DY_ERR_INVALID_ARG, "dy_detection_loss: null pointer");
  DY_REQUIRE(d->n_levels >= 1 && d->n_levels <= DY_MAX_LEVELS && d->batch > 0 && d->nc > 0 && d->gmax >= 0 && d->topk > 0, DY_ERR_INVALID_ARG,
             "dy_detection_loss: bad dims");
  DY_REQUIRE(d->reg_max == kRegMax, DY_ERR_UNSUPPORTED, "dy_detection_loss: reg_max %d not built (only 16)", d->reg_max);
  DY_REQUIRE(d->gmax == 0 || d->gt, DY_ERR_INVALID_ARG, "dy_detection_loss: gt is null");
  LossArgs a{};
  int A = 0;
  for (int i = 0; i < d->n_levels; ++i) {
    DY_REQUIRE(d->level[i] && d->h[i] > 0 && d->w[i] > 0 && d->ld[i] >= 4 * d->reg_max + d->nc, DY_ERR_INVALID_ARG, "dy_detection_loss: level %d invalid", i);
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
  a.gmax = d->gmax;
  a.topk = d->topk;
  a.alpha = d->alpha;
  a.beta = d->beta;
  a.box_gain = d->box_gain;
  a.cls_gain = d->cls_gain;
  a.dfl_gain = d->dfl_gain;
  a.gt = d->gt;
  a.out = d->out;
  const int64_t need = dy_detection_loss_workspace_bytes(d->batch, A, d->gmax, d->topk);
  DY_REQUIRE(d->workspace_bytes >= need && aligned16(d->workspace), DY_ERR_WORKSPACE, "dy_detection_loss: workspace %lld < %lld bytes",
             (long long)d->workspace_bytes, (long long)need);
  unsigned char* ws = reinterpret_cast<unsigned char*>(d->workspace);
  const size_t ba = (size_t)d->batch * A, bg = (size_t)d->batch * (d->gmax > 0 ? d->gmax : 1);
  a.pbox = reinterpret_cast<float*>(ws);
  ws += lalign(ba * 16);
  a.claims = reinterpret_cast<int*>(ws);
  ws += lalign(ba * 4);
  a.owner = reinterpret_cast<int*>(ws);
  ws += lalign(ba * 4);
  a.picks = reinterpret_cast<int*>(ws);
  ws += lalign(bg * d->topk * 4);
  a.gmax_al = reinterpret_cast<unsigned*>(ws);
  ws += lalign(bg * 4);
  a.gmax_ov = reinterpret_cast<unsigned*>(ws);
  ws += lalign(bg * 4);
  a.acc = reinterpret_cast<double*>(ws);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // zero the per-box maxima and the accumulators (contiguous tail of the workspace)
  zero_async(a.gmax_al, 2 * lalign(bg * 4) + lalign(6 * 8), st);  // a kernel, never a memset node (common_hip.h)
  const long long tot = (long long)d->batch * A;
  const unsigned blocks = (unsigned)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
  const unsigned blocks4 = (unsigned)((tot * 4 + 255) / 256 < 8192 ? (tot * 4 + 255) / 256 : 8192);
  hipLaunchKernelGGL(loss_decode_kernel, dim3(blocks4), dim3(256), 0, st, a);
  if (d->gmax > 0) {
    hipLaunchKernelGGL(tal_pick_kernel, dim3((unsigned)(d->batch * d->gmax)), dim3(64), 0, st, a);
    const long long np = (long long)d->batch * d->gmax * d->topk;
    hipLaunchKernelGGL(tal_resolve_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(tal_gtmax_kernel, dim3(blocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(loss_fg_kernel, dim3(blocks), dim3(256), 0, st, a);
  }
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, a);
  if (d->grad_level[0]) {
    for (int i = 0; i < d->n_levels; ++i) {
      DY_REQUIRE(d->grad_level[i] && d->ld_grad[i] >= 4 * d->reg_max + d->nc, DY_ERR_INVALID_ARG, "dy_detection_loss: grad level %d invalid", i);
      a.glevel[i] = d->grad_level[i];
      a.gld[i] = d->ld_grad[i];
    }
    const unsigned chunks = (unsigned)(kRegMax + (d->nc + 3) / 4);  // 16-byte chunks of a row that carry values
    DY_REQUIRE(tot * chunks < (1ll << 32), DY_ERR_UNSUPPORTED, "dy_detection_loss: gradient map too large for 32-bit chunk indices");
    const long long nb = (tot * (chunks + 2) + 255) / 256;
    hipLaunchKernelGGL(loss_grad_dense_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, st, a, chunks);
    hipLaunchKernelGGL(loss_grad_kernel, dim3(blocks), dim3(256), 0, st, a);
  }
  if (d->out_owner) {
    if (hipMemcpyAsync(d->out_owner, a.owner, ba * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return check_launch("dy_detection_loss copy");
  }
  return check_launch("dy_detection_loss");
}
