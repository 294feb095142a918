// Batched NMS for the single-label path of ops.non_max_suppression (utils/ops.py:181-332)
// including the greedy NMS of torchvision.ops.nms (call site ops.py:312).
//
//   K1 nms_filter_kernel   all (image, anchor) pairs in parallel: best class score and FIRST
//                          arg-max class (ops.py:290), keep score > conf (ops.py:250,291) and
//                          the optional class filter (ops.py:294-295); survivors are appended
//                          (wave-aggregated atomic) as 64-bit keys
//                              key = (~float_bits(score) << 32) | anchor
//                          Ascending key order == descending score, ties by ascending anchor
//                          index == the stable descending sort torchvision's CPU kernel uses on
//                          candidates listed in anchor order (ops.py:269 keeps anchor order).
//   K2 nms_suppress_kernel one workgroup per image: bitonic sort of the keys (LDS when they
//                          fit, the global workspace otherwise), truncate to max_nms
//                          (ops.py:301-302), then wave 0 runs the exact greedy scan in chunks of
//                          64 candidates: every lane tests its candidate against the kept
//                          list (LDS), then the chunk is resolved in score order with
//                          ballot/shuffle.  The scan stops at max_det kept (ops.py:313), which is
//                          exact because greedy NMS never revisits a kept box.
//
// IoU and box arithmetic follow the reference operation by operation in fp32 with
// contraction off, so that kept indices are bit-identical to the CPU path for identical
// inputs:  xyxy = xy -/+ wh/2 (ops.py:432-449); boxes + cls*max_wh (ops.py:305,311);
// area = (x2-x1)*(y2-y1); inter = max(0,xx2-xx1)*max(0,yy2-yy1);
// suppress iff inter / (area_i + area_j - inter) > iou_thres.
#include "common_hip.h"
#include "nms_ws.h"

#pragma clang fp contract(off)

namespace dy {

typedef unsigned long long u64;

struct NmsArgs {
  const float* pred;
  int batch, nc, nch, A;  // nch = 4 + nc + n_extra rows per image
  float conf, iou;
  int max_det, max_nms;
  float max_wh;
  int agnostic;
  const uint8_t* cmask;
  float* out;
  int* out_count;
  int* out_index;
  int* counts;          // [batch]
  u64* keys;            // [batch][P]
  unsigned short* cls;  // [batch][A] best class of every candidate anchor
  int P;                // per-image key capacity (power of two >= A)
  int SL;               // LDS sort capacity in keys (power of two)
  int multi;            // multi_label (ops.py:286-288): a candidate is an (anchor, class) pair, key low word = anchor * nc + class
  int cap;              // candidates an image can have: A, or A * nc with multi
};

__global__ __launch_bounds__(256) void nms_filter_kernel(const NmsArgs p) {
  const int lane = threadIdx.x & 63;
  const long long total = (long long)p.batch * p.A;
  const long long nthreads = (long long)gridDim.x * 256;
  // every wave iterates the same number of times so the ballot below is well defined
  const long long iters = (total + nthreads - 1) / nthreads;
  for (long long it = 0; it < iters; ++it) {
    const long long idx = it * nthreads + (long long)blockIdx.x * 256 + threadIdx.x;
    const bool in = idx < total;
    const int b = in ? (int)(idx / p.A) : 0;
    const int a = in ? (int)(idx - (long long)b * p.A) : 0;
    bool pass = false;
    float best = 0.f;
    int bj = 0;
    if (in) {
      const float* s = p.pred + ((size_t)b * p.nch + 4) * (size_t)p.A + a;
      best = s[0];
      for (int c = 1; c < p.nc; ++c) {
        const float v = s[(size_t)c * p.A];
        if (v > best) {
          best = v;
          bj = c;
        }
      }
      pass = best > p.conf;
      if (pass && p.cmask) pass = p.cmask[bj] != 0;
    }
    // A wave's 64 consecutive (image, anchor) pairs may straddle images: peel one image per
    // round, led by the lowest lane that still has a survivor to append.
    u64 todo = __ballot(pass);
    while (todo != 0ull) {
      const int leader = __ffsll((long long)todo) - 1;
      const int bb = __shfl(b, leader);
      const bool mine = pass && b == bb;
      const u64 m = __ballot(mine);
      int base = 0;
      if (lane == leader) base = atomicAdd(p.counts + bb, __popcll(m));
      base = __shfl(base, leader);
      if (mine) {
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        const unsigned bits = __float_as_uint(best);
        p.keys[(size_t)bb * p.P + pos] = ((u64)(~bits) << 32) | (u64)(unsigned)a;
        p.cls[(size_t)bb * p.A + a] = (unsigned short)bj;
      }
      todo &= ~m;
    }
  }
}

// multi_label (the validator's NMS, ops.py:286-288: `i, j = torch.where(cls > conf_thres)`): every (anchor, class) pair scored above conf
// is a candidate of its own.  Candidate order in the reference is row-major over (anchor, class), so the tie order of the stable
// descending sort is ascending anchor * nc + class: that number is the key's low word.
__global__ __launch_bounds__(256) void nms_filter_ml_kernel(const NmsArgs p) {
  const int lane = threadIdx.x & 63;
  const long long total = (long long)p.batch * p.A;
  const long long nthreads = (long long)gridDim.x * 256;
  const long long iters = (total + nthreads - 1) / nthreads;
  for (long long it = 0; it < iters; ++it) {
    const long long idx = it * nthreads + (long long)blockIdx.x * 256 + threadIdx.x;
    const bool in = idx < total;
    const int b = in ? (int)(idx / p.A) : 0;
    const int a = in ? (int)(idx - (long long)b * p.A) : 0;
    const float* s = p.pred + ((size_t)b * p.nch + 4) * (size_t)p.A + a;
    for (int c = 0; c < p.nc; ++c) {  // (wave-uniform trip count: the ballots below are well defined)
      const float v = in ? s[(size_t)c * p.A] : 0.f;
      bool pass = in && v > p.conf;
      if (pass && p.cmask) pass = p.cmask[c] != 0;
      u64 todo = __ballot(pass);
      while (todo != 0ull) {
        const int leader = __ffsll((long long)todo) - 1;
        const int bb = __shfl(b, leader);
        const bool mine = pass && b == bb;
        const u64 m = __ballot(mine);
        int base = 0;
        if (lane == leader) base = atomicAdd(p.counts + bb, __popcll(m));
        base = __shfl(base, leader);
        if (mine) {
          const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
          p.keys[(size_t)bb * p.P + pos] = ((u64)(~__float_as_uint(v)) << 32) | (u64)((unsigned)a * (unsigned)p.nc + (unsigned)c);
        }
        todo &= ~m;
      }
    }
  }
}

__device__ __forceinline__ bool iou_gt(float ix1, float iy1, float ix2, float iy2, float iarea, float jx1, float jy1,
                                       float jx2, float jy2, float jarea, float thr) {
  const float xx1 = fmaxf(ix1, jx1), yy1 = fmaxf(iy1, jy1);
  const float xx2 = fminf(ix2, jx2), yy2 = fminf(iy2, jy2);
  const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
  const float inter = w * h;
  const float ovr = inter / (iarea + jarea - inter);
  return ovr > thr;
}

__global__ __launch_bounds__(1024) void nms_suppress_kernel(const NmsArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  u64* skeys = reinterpret_cast<u64*>(dyn_smem);
  float* kept = reinterpret_cast<float*>(dyn_smem + (size_t)p.SL * 8);  // 5 arrays of max_det
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  int n = p.counts[b];
  if (n > p.cap) n = p.cap;
  u64* gkeys = p.keys + (size_t)b * p.P;

  int P2 = 1;
  while (P2 < n) P2 <<= 1;
  const bool in_lds = P2 <= p.SL;
  // Bitonic sort, one compare-exchange pair per thread per pass.  The LDS and the global variant are
  // separate loops on purpose: a pointer select between the two address spaces makes the compiler
  // fall back to generic (flat) accesses, which were ~10x slower per pass.
  auto bitonic = [&](auto* keys) {
    for (int k = 2; k <= P2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (P2 >> 1); t += 1024) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // element with bit j clear
          const int ixj = i | j;
          const u64 x = keys[i], y = keys[ixj];
          const bool up = (i & k) == 0;
          if ((x > y) == up) {
            keys[i] = y;
            keys[ixj] = x;
          }
        }
        __syncthreads();
      }
    }
  };
  if (in_lds) {
    for (int i = tid; i < P2; i += 1024) skeys[i] = i < n ? gkeys[i] : ~0ull;
    __syncthreads();
    if (n > 1) bitonic(skeys);
  } else {
    for (int i = n + tid; i < P2; i += 1024) gkeys[i] = ~0ull;
    __syncthreads();
    bitonic(gkeys);
  }
  if (tid >= 64) return;  // wave 0 runs the greedy scan

  const int lane = tid;
  const int nn = n < p.max_nms ? n : p.max_nms;
  float* kx1 = kept;
  float* ky1 = kept + p.max_det;
  float* kx2 = kept + 2 * p.max_det;
  float* ky2 = kept + 3 * p.max_det;
  float* kar = kept + 4 * p.max_det;
  const float* pr = p.pred + (size_t)b * p.nch * (size_t)p.A;
  float* outb = p.out + (size_t)b * p.max_det * 6;
  int nk = 0;
  for (int c0 = 0; c0 < nn && nk < p.max_det; c0 += 64) {
    const int i = c0 + lane;
    bool alive = i < nn;
    float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, ox1 = 0.f, oy1 = 0.f, ox2 = 0.f, oy2 = 0.f, area = 0.f, score = 0.f;
    int anchor = 0, cls = 0;
    if (alive) {
      const u64 key = in_lds ? skeys[i] : gkeys[i];
      anchor = (int)(unsigned)(key & 0xffffffffull);
      score = __uint_as_float(~(unsigned)(key >> 32));
      if (p.multi) {
        cls = anchor % p.nc;
        anchor = anchor / p.nc;
      } else {
        cls = (int)p.cls[(size_t)b * p.A + anchor];
      }
      const float cx = pr[anchor], cy = pr[(size_t)p.A + anchor];
      const float hw = pr[(size_t)2 * p.A + anchor] / 2.f, hh = pr[(size_t)3 * p.A + anchor] / 2.f;
      x1 = cx - hw;
      y1 = cy - hh;
      x2 = cx + hw;
      y2 = cy + hh;
      const float off = p.agnostic ? 0.f : (float)cls * p.max_wh;
      ox1 = x1 + off;
      oy1 = y1 + off;
      ox2 = x2 + off;
      oy2 = y2 + off;
      area = (ox2 - ox1) * (oy2 - oy1);
      for (int k = 0; k < nk; ++k)
        if (iou_gt(kx1[k], ky1[k], kx2[k], ky2[k], kar[k], ox1, oy1, ox2, oy2, area, p.iou)) {
          alive = false;
          break;
        }
    }
    u64 am = __ballot(alive);
    while (am != 0ull) {
      const int t = __ffsll((long long)am) - 1;  // best surviving candidate of the chunk (wave-uniform)
      // t is uniform, so broadcast its box with v_readlane (SGPR result) rather than LDS-routed shuffles
      auto bcast = [&](float v) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), t)); };
      const float tx1 = bcast(ox1), ty1 = bcast(oy1), tx2 = bcast(ox2), ty2 = bcast(oy2), tar = bcast(area);
      if (lane == t) {
        kx1[nk] = ox1;
        ky1[nk] = oy1;
        kx2[nk] = ox2;
        ky2[nk] = oy2;
        kar[nk] = area;
        float* o = outb + (size_t)nk * 6;
        o[0] = x1;
        o[1] = y1;
        o[2] = x2;
        o[3] = y2;
        o[4] = score;
        o[5] = (float)cls;
        if (p.out_index) p.out_index[(size_t)b * p.max_det + nk] = anchor;
        alive = false;
      }
      ++nk;
      if (nk >= p.max_det) break;
      if (alive && lane > t && iou_gt(tx1, ty1, tx2, ty2, tar, ox1, oy1, ox2, oy2, area, p.iou)) alive = false;
      am = __ballot(alive);
    }
  }
  for (int r = nk * 6 + lane; r < p.max_det * 6; r += 64) outb[r] = 0.f;
  if (p.out_index)
    for (int r = nk + lane; r < p.max_det; r += 64) p.out_index[(size_t)b * p.max_det + r] = -1;
  if (lane == 0) p.out_count[b] = nk;
}

// scale_boxes + clip_boxes on the kept rows (utils/ops.py:92-127, 335-354).
__global__ __launch_bounds__(256) void scale_boxes_kernel(float* __restrict__ boxes, const int* __restrict__ counts,
                                                          const float* __restrict__ params, int batch, int max_det) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= batch * max_det) return;
  const int b = idx / max_det, r = idx - b * max_det;
  if (r >= counts[b]) return;
  const float gain = params[b * 5], px = params[b * 5 + 1], py = params[b * 5 + 2];
  const float cw = params[b * 5 + 3], ch = params[b * 5 + 4];
  float* o = boxes + (size_t)idx * 6;
  o[0] = fminf(fmaxf((o[0] - px) / gain, 0.f), cw);
  o[1] = fminf(fmaxf((o[1] - py) / gain, 0.f), ch);
  o[2] = fminf(fmaxf((o[2] - px) / gain, 0.f), cw);
  o[3] = fminf(fmaxf((o[3] - py) / gain, 0.f), ch);
}

static inline int next_pow2(int v) { return nms_next_pow2(v); }
static inline size_t align_up(size_t v, size_t a) { return nms_align_up(v, a); }

}  // namespace dy

using namespace dy;

extern "C" int64_t dy_nms_workspace_bytes(int32_t batch, int32_t anchors) {
  if (batch <= 0 || anchors <= 0) return -1;
  return (int64_t)nms_ws_bytes(batch, anchors);
}

extern "C" int32_t dy_nms(const dy_nms_desc* d, dy_stream_t stream) {
  DY_REQUIRE(d && d->pred && d->out && d->out_count && d->workspace, DY_ERR_INVALID_ARG, "dy_nms: null pointer");
  DY_REQUIRE(d->batch > 0 && d->nc > 0 && d->anchors > 0 && d->n_extra >= 0, DY_ERR_INVALID_ARG, "dy_nms: bad dims");
  DY_REQUIRE(d->nc <= 65535, DY_ERR_UNSUPPORTED, "dy_nms: nc %d > 65535", d->nc);
  DY_REQUIRE(d->conf_thres >= 0.f && d->conf_thres <= 1.f && d->iou_thres >= 0.f && d->iou_thres <= 1.f,
             DY_ERR_INVALID_ARG, "dy_nms: thresholds must be in [0,1] (utils/ops.py:233-234)");
  DY_REQUIRE(d->max_det >= 1 && d->max_det <= 4096 && d->max_nms >= 1, DY_ERR_INVALID_ARG, "dy_nms: max_det must be in [1,4096]");
  const bool multi = d->multi_label != 0 && d->nc > 1;  // (ops.py:255: multi_label &= nc > 1)
  DY_REQUIRE(!multi || (long long)d->anchors * d->nc < (1ll << 30), DY_ERR_UNSUPPORTED, "dy_nms: multi_label with anchors * nc >= 2^30");
  DY_REQUIRE(!(multi && d->prefiltered), DY_ERR_UNSUPPORTED, "dy_nms: the fused candidate filter is single-label; multi_label filters here");
  const int wcap = multi ? d->anchors * d->nc : d->anchors;
  const int64_t need = dy_nms_workspace_bytes(d->batch, wcap);
  DY_REQUIRE(d->workspace_bytes >= need, DY_ERR_WORKSPACE, "dy_nms: workspace %lld < %lld bytes", (long long)d->workspace_bytes,
             (long long)need);
  DY_REQUIRE(aligned16(d->workspace), DY_ERR_INVALID_ARG, "dy_nms: workspace not 16-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);

  NmsArgs a{};
  a.pred = d->pred;
  a.batch = d->batch;
  a.nc = d->nc;
  a.nch = 4 + d->nc + d->n_extra;
  a.A = d->anchors;
  a.conf = d->conf_thres;
  a.iou = d->iou_thres;
  a.max_det = d->max_det;
  a.max_nms = d->max_nms;
  a.max_wh = d->max_wh;
  a.agnostic = d->agnostic;
  a.cmask = d->classes_mask;
  a.out = d->out;
  a.out_count = d->out_count;
  a.out_index = d->out_index;
  const NmsWs w = nms_ws_layout(d->workspace, d->batch, wcap);
  a.multi = multi ? 1 : 0;
  a.cap = wcap;
  a.P = w.P;
  a.counts = w.counts;
  a.keys = reinterpret_cast<u64*>(w.keys);
  a.cls = w.cls;
  a.SL = a.P < 16384 ? a.P : 16384;

  if (!d->prefiltered) {
    zero_async(a.counts, (size_t)d->batch * 4, st);
    const long long total = (long long)d->batch * d->anchors;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (multi) hipLaunchKernelGGL(nms_filter_ml_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(nms_filter_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    const int rc = check_launch(multi ? "nms_filter_ml_kernel" : "nms_filter_kernel");
    if (rc != DY_OK) return rc;
  }
  const size_t smem = (size_t)a.SL * 8 + align_up((size_t)d->max_det * 5 * 4, 16);
  static const hipError_t attr_once = hipFuncSetAttribute((const void*)nms_suppress_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)attr_once;
  hipLaunchKernelGGL(nms_suppress_kernel, dim3((unsigned)d->batch), dim3(1024), smem, st, a);
  return check_launch("nms_suppress_kernel");
}

extern "C" int32_t dy_scale_boxes(float* boxes, const int32_t* counts, const float* params, int32_t batch,
                                  int32_t max_det, dy_stream_t stream) {
  DY_REQUIRE(boxes && counts && params, DY_ERR_INVALID_ARG, "dy_scale_boxes: null pointer");
  DY_REQUIRE(batch > 0 && max_det > 0, DY_ERR_INVALID_ARG, "dy_scale_boxes: bad dims");
  const int total = batch * max_det;
  hipLaunchKernelGGL(scale_boxes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), boxes, counts, params, batch, max_det);
  return check_launch("scale_boxes_kernel");
}
