// One Detect branch from its second 3x3 convolution to the decoded output, in one kernel (16-bit storage):
//
//   box branch (KIND 1):  Conv 3x3 64->64 + SiLU  ->  Conv2d 1x1 64->64 (+bias)  ->  DFL softmax expectation per side
//                         ->  dist2bbox with the anchor grid, x stride  ->  rows 0..3 of pred (N, 4 + nc, A)
//   class branch (KIND 2): Conv 3x3 64->64 + SiLU  ->  Conv2d 1x1 64->nc (+bias)  ->  sigmoid  ->  rows 4.. of pred,
//                         best class / score per anchor  ->  NMS candidate list (conf filter, optional class mask)
//
// Reference: Detect.forward head.py:64-70 (cv2[i][1], cv2[i][2], cv3[i][1], cv3[i][2] of the legacy v8 head, head.py:43-57),
// Detect._inference head.py:100-131, DFL block.py:58-76, make_anchors / dist2bbox tal.py:333-357, the candidate filter of
// non_max_suppression ops.py:250,290-295.
//
// Why: layer by layer the trunk outputs (34,000 anchors x 2 branches x 64 channels) are written by the 3x3 kernels and read
// back by the fused tail (detect_head.hip): 256 B per anchor each way, 4.5 GB per pass at B = 256 — the tail runs at the HBM
// rate and cannot get faster on its own.  Here the 3x3 output tile never leaves the CU: it is rounded to the storage type
// exactly where the layer-by-layer path rounds it, laid down in LDS in the halo image's format, and consumed by the 1x1 MFMAs.
//
// The 3x3 part is conv3x3_hreg.hip's (weights of the wave's 16-cout fragment in 72 registers, swizzled 64-byte-pitch halo
// image filled by LDS-DMA, 8 x 16 pixel tiles, three 256-thread workgroups per CU).  After the tile's two chunks:
//   1. every wave SiLUs its 16 couts x 128 pixels and writes them (8 bytes per lane) into the `mid` image
//      [chunk 2][row 8][pixel 16] x 64 B with the halo's part swizzle; barrier;
//   2. 1x1: the wave's A fragment(s) sit in 8 registers; B fragments are conflict-free ds_read_b128 of `mid`.
//      Box: wave w computes side w (16 bins) for all 8 rows: 16 MFMAs.  Class: the waves split the rows (2 each): 4 MFMAs;
//   3. box: a lane holds 4 of a side's 16 bins of one pixel; the softmax expectation is reduced over the four lane quarters with
//      two xor-shuffles; the four sides meet through 2 KB of LDS and 256 threads write (cx, w) / (cy, h) of the 128 pixels;
//      class: sigmoid per class, first arg-max over the quarters by shuffles (lowest class wins ties, as cls.max(1)), ballot +
//      one atomicAdd per wave to append the candidates (key = ~score bits << 32 | anchor, as detect_head.hip / nms.hip).
#include "common_hip.h"
#include "nms_ws.h"

namespace DY_NS {


struct HheadArgs {
  const void* x;       // trunk input, NHWC (N, H, W, 64), pitch ldx
  const void* w3;      // 3x3 64->64, DY_WLAYOUT_HALO3X3 (NF = 4)
  const float* b3;     // 64
  const void* w1;      // 1x1, DY_WLAYOUT_FRAG1X1: box cout 64 (4 fragments per k-group), class cout nc <= 16 (1 fragment)
  const float* b1;     // 64 / 16
  float* out;          // pred (N, 4 + nc, A) fp32
  int N, H, W, ldx, A, a0, nc;
  unsigned x_bytes;
  float stride;
  int tilesX, tilesY, nSpatial;
  int* counts;
  unsigned long long* keys;
  unsigned short* cls;
  int P;
  float conf;
  const uint8_t* cmask;
};

constexpr int kHhTH = 8, kHhTW = 16, kHhHH = 10, kHhHW = 24;
constexpr int kHhStage = 16 * 1024, kHhStages = 2;
constexpr int kHhMid = 2 * kHhTH * kHhTW * 64;  // 16 KB: the tile's 3x3 output, [chunk][row][pixel] x 64 B

// Exchange between the four 16-lane quarters of a wave without touching LDS (v_permlane16_swap / v_permlane32_swap, gfx950):
// quarter_pair(v): every lane gets (its own quarter pair's two values) -> combine with op: rows {0,1} and {2,3}; half_pair: {0,1} with {2,3}.
__device__ __forceinline__ void quarter_views(float v, float& a, float& b) {  // a = rows (0,0,2,2), b = rows (1,1,3,3) of v
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void half_views(float v, float& a, float& b) {  // a = rows (0,1,0,1), b = rows (2,3,2,3) of v
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_quarters_max(float v) {
  float a, b;
  quarter_views(v, a, b);
  v = fmaxf(a, b);
  half_views(v, a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float wave_quarters_sum(float v) {
  float a, b;
  quarter_views(v, a, b);
  v = a + b;
  half_views(v, a, b);
  return a + b;
}

template <typename T, int KIND>
__global__ __launch_bounds__(256, 3) void conv3x3_hhead_kernel(const HheadArgs p) {
  constexpr int EPC = Elem<T>::EPC;  // 8
  constexpr int NCH = 2;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kHhStages * kHhStage + kHhMid + 4 * kHhTH * kHhTW * 4];
  unsigned char* const mid = smem + kHhStages * kHhStage;
  float* const dsm = reinterpret_cast<float*>(mid + kHhMid);  // [side 4][pixel 128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lr = lane & 15;

  const int G = (int)gridDim.x;
  const int sb = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);  // XCD-contiguous tile order (guide T1)
  const int myTiles = sb < p.nSpatial ? (p.nSpatial - sb + G - 1) / G : 0;
  if (myTiles <= 0) return;
  const int nItems = myTiles * NCH;

  // ---- register-resident weights: the wave's 16-cout fragment of the 3x3, and its fragment(s) of the 1x1 ----
  u32x4 wreg[NCH][9];
  {
    const u32x4* wg = reinterpret_cast<const u32x4*>(p.w3);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t) wreg[c][t] = wg[((c * 9 + t) * 4 + wave) * 64 + lane];
  }
  const f32x4 bias3 = *reinterpret_cast<const f32x4*>(p.b3 + wave * 16 + lq * 4);
  // the 1x1 fragments (2 x 16 B per lane) and its bias are re-read per tile (L2 / L1 hits): 12 registers the 3x3 loop needs more
  const u32x4* const w1g = reinterpret_cast<const u32x4*>(p.w1) + (KIND == 1 ? wave * 64 : 0) + lane;
  const float* const b1g = p.b1 + (KIND == 1 ? wave * 16 : 0) + lq * 4;

  // ---- halo loader (conv3x3_hreg.hip): buffer-addressed LDS-DMA, lane-constant offsets + a scalar tile offset, zeros by range check ----
  constexpr int NDMA = 4;
  constexpr unsigned kOob = 0xfffffff0u;
  const unsigned pre = (unsigned)((p.W + 1) * p.ldx) * 2u;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) - pre, 0, p.x_bytes + pre, 0x00020000);
  unsigned rel[NDMA];
  int hyx[NDMA];
#pragma unroll
  for (int k = 0; k < NDMA; ++k) {
    const int s = (k * 4 + wave) * 64 + lane;
    const int pix = s >> 2, part = s & 3;
    const int hy = pix / kHhHW, hx = pix - hy * kHhHW;
    const bool dead = hx >= kHhTW + 2 || hy >= kHhHH;
    rel[k] = dead ? kOob : (unsigned)((hy * p.W + hx) * p.ldx + (part ^ ((hx >> 1) & 3)) * EPC) * 2u;
    hyx[k] = hy | (hx << 8);
  }
  unsigned voff[NDMA];
  unsigned l_base = 0;
  int l_tile = sb, l_chunk = 0, l_item = 0;
  auto setup_tile = [&](int tile) {
    const int tx = tile % p.tilesX;
    const int r = tile / p.tilesX;
    const int ty = r % p.tilesY, n = r / p.tilesY;
    const int y0 = ty * kHhTH, x0 = tx * kHhTW;
    l_base = (unsigned)(((n * p.H + y0) * p.W + x0) * p.ldx) * 2u;
    const bool interior = y0 > 0 && y0 + kHhTH + 1 <= p.H && x0 > 0 && x0 + kHhTW + 1 <= p.W;  // wave-uniform
    if (interior) {
#pragma unroll
      for (int k = 0; k < NDMA; ++k) voff[k] = rel[k];
    } else {
#pragma unroll
      for (int k = 0; k < NDMA; ++k) {
        const int gy = y0 - 1 + (hyx[k] & 255), gx = x0 - 1 + (hyx[k] >> 8);
        voff[k] = ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) ? rel[k] : kOob;
      }
    }
  };
  auto issue_dma = [&](int stage) {
    unsigned char* sa = smem + stage * kHhStage;
    const bool live = l_item < nItems;
    if (live) {
      const unsigned soff = l_base + (unsigned)l_chunk * (4u * EPC * (unsigned)sizeof(T));
#pragma unroll
      for (int k = 0; k < NDMA; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(sa + (k * 4 + wave) * 1024), 16, (int)voff[k], (int)soff, 0, 0);
      ++l_item;
      if (++l_chunk == NCH) {
        l_chunk = 0;
        l_tile += G;
        if (l_item < nItems) setup_tile(l_tile);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NDMA; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(sa + (k * 4 + wave) * 1024), 16, (int)kOob, 0, 0, 0);
    }
  };

  int lane_base[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) lane_base[q] = (lr + q) * 64 + ((lq ^ (((lr + q) >> 1) & 3)) * 16);

  f32x4 acc[kHhTH];

  auto compute = [&](int stg, int c) {
    const unsigned char* sa = smem + stg * kHhStage;
#pragma unroll
    for (int iy = 0; iy < kHhHH; ++iy) {
      u32x4 a[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) a[q] = *reinterpret_cast<const u32x4*>(sa + lane_base[q] + iy * (kHhHW * 64));
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int o = iy - r;
        if (o >= 0 && o < kHhTH) {
#pragma unroll
          for (int q = 0; q < 3; ++q) acc[o] = Elem<T>::mma(wreg[c][r * 3 + q], a[q], acc[o]);
        }
      }
    }
  };

  // ---- the tail of one tile ----
  typedef __attribute__((ext_vector_type(4))) T t4;
  const int mid_w = (wave >> 1) * (kHhTH * kHhTW * 64) + lr * 64 + ((((wave & 1) * 2 + (lq >> 1)) ^ ((lr >> 1) & 3)) * 16) + (lq & 1) * 8;  // + o * 1024
  const int mid_r = lane_base[0];                                                                                                          // + c2 * 8192 + o * 1024
  // tail, part 1 (end of the tile's last item): SiLU, round to the storage type (where the layer-by-layer path rounds), into `mid`.
  // The item's closing barrier publishes it.
  auto tail_mid = [&]() {
#pragma unroll
    for (int o = 0; o < kHhTH; ++o) {
      t4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = Elem<T>::from_f32(silu_f32(acc[o][e]));
      *reinterpret_cast<u32x2*>(mid + mid_w + o * (kHhTW * 64)) = __builtin_bit_cast(u32x2, ov);
    }
  };
  // tail, part 2, DEFERRED to the start of the next item (after that item's DMA has been issued, before its MFMAs): the global
  // stores then have a whole item of matrix work behind them before the item's closing s_waitcnt vmcnt(0), instead of being waited
  // for right after they were issued (first version: the fused kernels cost as much as conv + separate tail kernel)
  // class branch: the candidate append of a tile is FINISHED one tile later (r04).  The counter atomic returns the wave's slot range;
  // the first form used it at once — every wave stood still for the round trip (1-2 us of an 11 us tile, in front of the item's MFMAs):
  // the class branch ran 17 % slower than the box branch for 1/6 of its 1x1 work.  Now the atomic is issued at the end of tail_out and
  // its value read at the start of the next one (or after the last tile): it returns under a whole item of matrix work.
  int pv_base = 0, pv_n = 0, pv_n0 = 0, pv_total = 0;
  unsigned long long pv_mk0 = 0ull, pv_mk1 = 0ull;
  float pv_best[2] = {0.f, 0.f};
  int pv_bj[2] = {0, 0}, pv_a[2] = {0, 0};
  auto flush_keys = [&]() {
    if (pv_total != 0) {
      const int base = __shfl(pv_base, 0);
      const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned long long mk = j ? pv_mk1 : pv_mk0;
        if ((mk >> lane) & 1ull) {
          const int pos = base + (j ? pv_n0 : 0) + __popcll(mk & below);
          p.keys[(size_t)pv_n * p.P + pos] = ((unsigned long long)(~__float_as_uint(pv_best[j])) << 32) | (unsigned long long)(unsigned)pv_a[j];
          p.cls[(size_t)pv_n * p.A + pv_a[j]] = (unsigned short)pv_bj[j];
        }
      }
      pv_total = 0;
    }
  };
  auto tail_out = [&](int tile) {
    if constexpr (KIND == 2) flush_keys();
    const int tx = tile % p.tilesX;
    const int r_ = tile / p.tilesX;
    const int ty = r_ % p.tilesY, n = r_ / p.tilesY;
    u32x4 w1reg[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) w1reg[c] = w1g[c * (KIND == 1 ? 4 : 1) * 64];
    const f32x4 bias1 = *reinterpret_cast<const f32x4*>(b1g);
    if constexpr (KIND == 1) {
      // 1x1: side `wave`, bins lq*4 .. +3 of pixel (o, lr); then DFL = softmax expectation over the side's 16 bins
#pragma unroll
      for (int o = 0; o < kHhTH; ++o) {
        f32x4 lg = bias1;
#pragma unroll
        for (int c2 = 0; c2 < NCH; ++c2)
          lg = Elem<T>::mma(w1reg[c2], *reinterpret_cast<const u32x4*>(mid + mid_r + c2 * (kHhTH * kHhTW * 64) + o * (kHhTW * 64)), lg);
        const float m = wave_quarters_max(fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3])));
        float den = 0.f, num = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ex = __builtin_amdgcn_exp2f((lg[e] - m) * 1.4426950408889634f);
          den += ex;
          num += ex * (float)(lq * 4 + e);
        }
        den = wave_quarters_sum(den);
        num = wave_quarters_sum(num);
        if (lq == 0) dsm[wave * (kHhTH * kHhTW) + o * kHhTW + lr] = num * __builtin_amdgcn_rcpf(den);
      }
      __syncthreads();
      // dist2bbox (tal.py:348-357) x stride: thread -> (pixel, axis); sides: 0 left, 1 top, 2 right, 3 bottom
      const int px = tid & 127, axis = tid >> 7;
      const int o = px >> 4, xi = px & 15;
      const int yy = ty * kHhTH + o, xx = tx * kHhTW + xi;
      if (yy < p.H && xx < p.W) {
        const float d_lo = dsm[axis * (kHhTH * kHhTW) + px], d_hi = dsm[(axis + 2) * (kHhTH * kHhTW) + px];
        const float ctr = (float)(axis == 0 ? xx : yy) + 0.5f;
        const float lo = ctr - d_lo, hi = ctr + d_hi;
        float* op = p.out + (size_t)n * (size_t)(4 + p.nc) * p.A + (size_t)(p.a0 + yy * p.W + xx);
        op[(size_t)axis * p.A] = (lo + hi) * 0.5f * p.stride;
        op[(size_t)(axis + 2) * p.A] = (hi - lo) * p.stride;
      }
    } else {
      // 1x1 64 -> nc (one 16-cout fragment): the waves split the rows; sigmoid, scores out, first arg-max, candidate filter.
      // The wave's two rows are filtered together: ONE counter atomic (with return: the wave waits for it) per wave and tile
      // instead of one per row - on the P2 level every tile has candidates and the atomics of an image all hit one address.
      float bestv[2];
      int bjv[2], av[2];
      bool passv[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int o = wave * 2 + j;
        f32x4 lg = bias1;
#pragma unroll
        for (int c2 = 0; c2 < NCH; ++c2)
          lg = Elem<T>::mma(w1reg[c2], *reinterpret_cast<const u32x4*>(mid + mid_r + c2 * (kHhTH * kHhTW * 64) + o * (kHhTW * 64)), lg);
        const int yy = ty * kHhTH + o, xx = tx * kHhTW + lr;
        const bool inside = yy < p.H && xx < p.W;
        const int a = p.a0 + yy * p.W + xx;
        float* op = p.out + (size_t)n * (size_t)(4 + p.nc) * p.A + (size_t)a;
        float best = -1.f;
        int bj = 0x7fff;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = lq * 4 + e;
          const float pr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(lg[e] * -1.4426950408889634f));
          if (c < p.nc) {
            if (inside) op[(size_t)(4 + c) * p.A] = pr;
            if (pr > best) best = pr, bj = c;  // ascending c: the first maximum stays
          }
        }
#pragma unroll
        for (int sh = 16; sh <= 32; sh <<= 1) {  // the four quarters hold classes 0-3, 4-7, 8-11, 12-15 of this pixel
          const float ob = __shfl_xor(best, sh);
          const int oj = __shfl_xor(bj, sh);
          if (ob > best || (ob == best && oj < bj)) best = ob, bj = oj;
        }
        bool pass = p.keys != nullptr && inside && lq == 0 && best > p.conf;
        if (pass && p.cmask) pass = p.cmask[bj] != 0;
        bestv[j] = best, bjv[j] = bj, av[j] = a, passv[j] = pass;
      }
      if (p.keys != nullptr) {
        const unsigned long long mk0 = __ballot(passv[0]), mk1 = __ballot(passv[1]);
        const int n0 = __popcll(mk0), total = n0 + __popcll(mk1);
        if (total != 0) {
          if (lane == 0) pv_base = atomicAdd(p.counts + n, total);  // (value used by flush_keys, one tile later)
          pv_n = n, pv_n0 = n0, pv_total = total, pv_mk0 = mk0, pv_mk1 = mk1;
#pragma unroll
          for (int j = 0; j < 2; ++j) pv_best[j] = bestv[j], pv_bj[j] = bjv[j], pv_a[j] = av[j];
        }
      }
    }
  };

  // ---- item pipeline: two stages, the next item's DMA in flight during this item's MFMAs, one barrier per item ----
  setup_tile(l_tile);
  issue_dma(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int c_tile = sb, pending = -1;
  for (int it = 0; it < nItems; it += NCH) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      issue_dma((c & 1) ^ 1);  // the other stage was last read one item ago: every wave has passed that item's barrier
      if (c == 0) {
        if (pending >= 0) tail_out(pending);  // the previous tile's 1x1 + decode: `mid` was published by the barrier that ended it
#pragma unroll
        for (int o = 0; o < kHhTH; ++o) acc[o] = bias3;  // (only now: the accumulator registers are free during tail_out)
      }
      compute(c & 1, c);
      if (c == NCH - 1) {
        tail_mid();
        pending = c_tile;
        c_tile += G;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // next item's halo complete and visible; `mid` written (last chunk) / `dsm` free again
    }
  }
  if (pending >= 0) tail_out(pending);
  if constexpr (KIND == 2) flush_keys();
}

template <typename T>
static int launch_hhead(const HheadArgs& a, int kind, hipStream_t st) {
  int grid = 256 * 3;
  if (a.nSpatial < grid) grid = a.nSpatial;
  grid = (grid + 7) / 8 * 8;
  if (kind == 1)
    hipLaunchKernelGGL((conv3x3_hhead_kernel<T, 1>), dim3((unsigned)grid), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv3x3_hhead_kernel<T, 2>), dim3((unsigned)grid), dim3(256), 0, st, a);
  return check_launch("conv3x3_hhead_kernel");
}

}  // namespace DY_NS

using namespace DY_NS;

#ifndef DYOLO_L2E_BUILD
extern "C" int32_t dy_detect_branch_fused_supported(int32_t c_in, int32_t c_mid, int32_t c_out, int32_t kind, int32_t nc, int32_t reg_max, int32_t dtype) {
  if (!(dtype == DY_BF16 || dtype == DY_F16) || reg_max != 16 || c_in != 64 || c_mid != 64) return 0;
  if (kind == 1) return c_out == 64;
  if (kind == 2) return c_out == nc && nc >= 1 && nc <= 16;
  return 0;
}

namespace dy_l2e {
int32_t branch_entry(const dy_branch_desc* d, dy_stream_t stream);
}
namespace dy {
int32_t branch_entry(const dy_branch_desc* d, dy_stream_t stream);
}
extern "C" int32_t dy_detect_branch_fused(const dy_branch_desc* d, dy_stream_t stream) {
  return (d != nullptr && d->act_l2e) ? dy_l2e::branch_entry(d, stream) : dy::branch_entry(d, stream);
}

extern "C" int32_t dy_nms_reset_counts(void* nms_workspace, int32_t batch, dy_stream_t stream) {
  DY_REQUIRE(nms_workspace && batch > 0, DY_ERR_INVALID_ARG, "dy_nms_reset_counts: bad arguments");
  zero_async(nms_workspace, (size_t)batch * 4, reinterpret_cast<hipStream_t>(stream));
  return check_launch("dy_nms_reset_counts");
}
#endif

namespace DY_NS {
int32_t branch_entry(const dy_branch_desc* d, dy_stream_t stream) {
  DY_REQUIRE(d && d->x && d->w3 && d->b3 && d->w1 && d->b1 && d->out, DY_ERR_INVALID_ARG, "dy_detect_branch_fused: null pointer");
  DY_REQUIRE(dy_detect_branch_fused_supported(d->c_in, d->c_mid, d->kind == 1 ? 4 * d->reg_max : d->nc, d->kind, d->nc, d->reg_max, d->dtype), DY_ERR_UNSUPPORTED,
             "dy_detect_branch_fused: kind %d c_in %d c_mid %d nc %d reg_max %d dtype %d is not built (run dy_conv2d_nhwc + dy_detect_head_decode)", d->kind, d->c_in,
             d->c_mid, d->nc, d->reg_max, d->dtype);
  DY_REQUIRE(d->batch > 0 && d->h > 0 && d->w > 0 && d->ld_x >= d->c_in && (d->ld_x * 2) % 16 == 0 && aligned16(d->x) && aligned16(d->w3) && aligned16(d->w1) &&
                 aligned16(d->b3) && aligned16(d->b1),
             DY_ERR_INVALID_ARG, "dy_detect_branch_fused: views must be whole 16-byte chunks");
  DY_REQUIRE(d->anchors > 0 && d->anchor0 >= 0 && d->anchor0 + d->h * d->w <= d->anchors, DY_ERR_INVALID_ARG, "dy_detect_branch_fused: the level's anchors [%d, %d) exceed A = %d",
             d->anchor0, d->anchor0 + d->h * d->w, d->anchors);
  DY_REQUIRE((long long)d->batch * d->h * d->w * d->ld_x * 2 < (1ll << 31), DY_ERR_UNSUPPORTED, "dy_detect_branch_fused: input view exceeds 2 GiB (32-bit element offsets)");
  HheadArgs a{};
  a.x = d->x, a.w3 = d->w3, a.b3 = d->b3, a.w1 = d->w1, a.b1 = d->b1, a.out = d->out;
  a.N = d->batch, a.H = d->h, a.W = d->w, a.ldx = d->ld_x, a.A = d->anchors, a.a0 = d->anchor0, a.nc = d->nc, a.stride = d->stride;
  a.x_bytes = (unsigned)((long long)d->batch * d->h * d->w * d->ld_x * 2);
  a.tilesX = (d->w + kHhTW - 1) / kHhTW, a.tilesY = (d->h + kHhTH - 1) / kHhTH;
  a.nSpatial = d->batch * a.tilesX * a.tilesY;
  if (d->kind == 2 && d->nms_workspace) {
    DY_REQUIRE(aligned16(d->nms_workspace) && d->nms_workspace_bytes >= (int64_t)nms_ws_bytes(d->batch, d->anchors), DY_ERR_WORKSPACE,
               "dy_detect_branch_fused: nms_workspace too small or misaligned (need %lld bytes)", (long long)nms_ws_bytes(d->batch, d->anchors));
    const NmsWs w = nms_ws_layout(d->nms_workspace, d->batch, d->anchors);
    a.counts = w.counts, a.keys = w.keys, a.cls = w.cls, a.P = w.P;
    a.conf = d->conf_thres, a.cmask = d->classes_mask;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return d->dtype == DY_BF16 ? launch_hhead<bf16_t>(a, d->kind, st) : launch_hhead<f16_t>(a, d->kind, st);
}
}  // namespace DY_NS

