// The first TWO layers in one kernel: fp32 NCHW image -> Conv(3, 32, 3, 2)+SiLU -> RepVGGBlock(32, 64, stride 2)+SiLU
// (deploy form: one 3x3 kernel + bias) -> NHWC activations at 1/4 resolution.
//
// Replaces yolov8-p2-repvgg.yaml layers 0 and 1 (ultralytics/nn/modules/conv.py:37-55, block.py:1393-1490) plus the
// predictor's layout step.  Run separately (conv_stem.hip, then the stride-2 implicit GEMM) both layers are purely
// HBM-bound and the 320 x 320 x 32 intermediate is written and read back: per 640 x 640 image 4.9 MB in, 6.6 MB out and
// in again, 3.3 MB out.  Fused, the intermediate lives in LDS: 4.9 + 3.3 MB per image (2.6x less traffic).
//
// A 256-thread workgroup produces 8 x 16 layer-1 pixels x 64 channels per tile (persistent over tiles):
//   A. the 3 x 35 x 67 fp32 input patch -> LDS by LDS-DMA (global_load_lds, 16 bytes per lane, zero page outside the
//      image), requested one tile ahead: it lands while the previous tile's layer 1 and epilogue run;
//   B. the 17 x 33 stem pixels the tile's taps touch: per 16-pixel fragment every lane gathers its 8 taps from the patch
//      (im2col on the fly, K = 27 padded to 32), 2 MFMAs, SiLU, and writes 4 channels of its pixel into the LDS stem
//      buffer — zero where the stem pixel lies outside the image, which is layer 1's zero padding;
//   C. layer 1 as 9 taps x one 32-deep MFMA k-step from the stem buffer: every wave computes all 128 pixels for ITS 16
//      output channels, so its weights (16 x 288) stay in 36 registers as MFMA A operands for the lifetime of the
//      workgroup; SiLU, the four waves assemble whole pixel rows in LDS, 16-byte row stores.
// The kernel is bound by VALU instruction issue (im2col gather, SiLU on 1.5x the outputs of layer 1 alone), not by HBM:
// every phase is written for instruction count (branch-free gather, index arithmetic hoisted out of the tile loop).
// Stem buffer layout: [row 0..16][column parity][index 0..16][32 channels], i.e. stem column 2*index + parity, so that
// the 16 pixels of a stride-2 fragment are CONSECUTIVE entries of one plane; the four 16-byte chunks of an entry are
// XOR-swizzled by (index >> 2) & 3: conflict-free ds_read_b128 operand reads.
// LDS 64 KiB -> two workgroups per CU: one loads its patch while the other computes.
#include "common_hip.h"

namespace DY_NS {

__device__ __attribute__((aligned(16))) const unsigned int g_s2zero[4] = {0, 0, 0, 0};

struct Stem2Args {
  const float* x;
  const void* w0;
  const float* b0;
  const void* w1;
  const float* b1;
  void* y;
  int N, H, W, H0, W0, H1, W1, ldy, act0, act1;
  int tilesX, tilesY, ntiles;
};

constexpr int kS2TH = 8, kS2TW = 16;                              // layer-1 pixels per tile
constexpr int kS2SH = 2 * kS2TH + 1, kS2SW = 2 * kS2TW + 1;       // 17 x 33 stem pixels
constexpr int kS2PH = 2 * kS2SH + 1;                              // 35 input rows
constexpr int kS2Pitch = 68;                                      // floats per patch row: 17 aligned float4
constexpr int kS2Plane = kS2PH * kS2Pitch;
constexpr int kS2PatchBytes = 28 * 1024;                          // 3 * 35 * 68 floats = 28560 B, padded to 28 DMA instructions of 1 KiB
constexpr int kS2Idx = 17;                                        // entries per (row, parity) plane
constexpr int kS2StemBytes = kS2SH * 2 * kS2Idx * 64;             // 36992
constexpr int kS2EpPitch = 128 + 16;
static_assert(3 * kS2Plane * 4 <= kS2PatchBytes && 128 * kS2EpPitch <= kS2StemBytes, "epilogue scratch aliases the stem buffer");

template <typename T>
__global__ __launch_bounds__(256, 2) void stem2_fused_kernel(const Stem2Args p) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  __shared__ __attribute__((aligned(16))) unsigned char smem[kS2PatchBytes + kS2StemBytes];
  float* patch = reinterpret_cast<float*>(smem);
  unsigned char* stem = smem + kS2PatchBytes;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lr = lane & 15;
  const T* __restrict__ w0 = reinterpret_cast<const T*>(p.w0);
  const T* __restrict__ w1 = reinterpret_cast<const T*>(p.w1);
  T* __restrict__ yg = reinterpret_cast<T*>(p.y);

  // ---- weights -> registers, once per workgroup ----
  u32x4 wf0[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) wf0[j] = *reinterpret_cast<const u32x4*>(w0 + (size_t)(j * 16 + lr) * 32 + lq * 8);
  u32x4 wf1[9];  // layer 1: wave w owns couts 16w .. 16w+15 for all 128 pixels of the tile (36 registers of weights)
#pragma unroll
  for (int t = 0; t < 9; ++t) wf1[t] = *reinterpret_cast<const u32x4*>(w1 + (size_t)(wave * 16 + lr) * 288 + t * 32 + lq * 8);
  const f32x4 bias1 = *reinterpret_cast<const f32x4*>(p.b1 + wave * 16 + lq * 4);
  // patch offset of tap k = 8*lq + e relative to the pixel's window origin.  The padding taps k >= 27 read the origin:
  // their weights are zero and the patch holds image data or zeros, so the gather needs no select (branch-free).
  int koff[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = lq * 8 + e;
    const int c = k / 9, r = (k - c * 9) / 3, q = k - c * 9 - r * 3;
    koff[e] = k < 27 ? c * kS2Plane + r * kS2Pitch + q : 0;
  }
  // phase A (LDS-DMA): float4 number i = (7*wave + k)*64 + lane of the patch = row i / 17 (= c*35 + py), column group i % 17
  auto request_patch = [&](int tile) {
    int t = tile;
    const int tx = t % p.tilesX;
    t /= p.tilesX;
    const int ty = t % p.tilesY;
    const int n = t / p.tilesY;
    const int gy0 = 4 * ty * kS2TH - 3, gx0 = 4 * tx * kS2TW - 4;
    const float* img = p.x + (size_t)n * 3 * p.H * p.W;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int i = (wave * 7 + k) * 64 + lane;
      const int row = (i * 3856) >> 16;         // i / 17 for i < 1792
      const int c = (row * 1873) >> 16;         // row / 35 for row < 106
      const int gy = gy0 + row - c * kS2PH, gx = gx0 + 4 * (i - row * 17);
      const bool ok = row < 3 * kS2PH && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      const float* src = ok ? img + (size_t)(c * p.H + gy) * p.W + gx : reinterpret_cast<const float*>(g_s2zero);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + (wave * 7 + k) * 1024), 16, 0, 0);
    }
  };
  f32x4 bias0[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bias0[j] = *reinterpret_cast<const f32x4*>(p.b0 + j * 16 + lq * 4);

  // Tile order (r05).  Workgroups go to the 8 XCDs round robin, and with tile = blockIdx + k * gridDim neighbouring tiles ran behind DIFFERENT L2s: a
  // patch row is 17 float4 starting 16 bytes in front of a 256-byte tile column, i.e. three 128-byte lines for 272 bytes — the left halo's line is the
  // neighbour's data — and 35 rows for 32: the counters showed 1.99 GB fetched for the 1.26 GB image (1.58x).  Now every XCD owns a contiguous range of
  // tiles and its workgroups walk it interleaved, so the tiles in flight behind one L2 are neighbours in x and the shared lines are fetched once.
  int t_first, t_step, t_end;
  if (((int)gridDim.x & 7) == 0) {
    const int xcd = (int)blockIdx.x & 7, q = p.ntiles >> 3, r = p.ntiles & 7;
    const int start = xcd * q + (xcd < r ? xcd : r);
    t_first = start + ((int)blockIdx.x >> 3), t_step = (int)gridDim.x >> 3, t_end = start + q + (xcd < r ? 1 : 0);
  } else {
    t_first = (int)blockIdx.x, t_step = (int)gridDim.x, t_end = p.ntiles;
  }
  if (t_first < t_end) request_patch(t_first);
  for (int tile = t_first; tile < t_end; tile += t_step) {
    int t = tile;
    const int tx = t % p.tilesX;
    t /= p.tilesX;
    const int ty = t % p.tilesY;
    const int n = t / p.tilesY;
    const int oy0 = ty * kS2TH, ox0 = tx * kS2TW;

    // ---- A: the patch was requested during the previous tile (LDS-DMA) ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // patch landed in every wave; the previous tile's row stores are done reading the scratch

    // ---- B: stem pixels (sy_l, sx_l) = stem row 2*oy0 - 1 + sy_l, column 2*ox0 - 1 + sx_l ----
#pragma unroll 3
    for (int it = 0; it < ((kS2SH * kS2SW + 15) / 16 + 3) / 4; ++it) {
      const int tt = (wave + 4 * it) * 16 + lr;
      const bool valid = tt < kS2SH * kS2SW;
      const int tc = valid ? tt : kS2SH * kS2SW - 1;
      const int sy = (tc * 1986) >> 16;  // tc / 33 for tc < 561
      const int sx = tc - sy * kS2SW;
      const float* org = patch + (2 * sy) * kS2Pitch + 2 * sx + 1;
      float g[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = org[koff[e]];
      const u32x4 a = Chunk<T>::pack(g);
      const int gsy = 2 * oy0 - 1 + sy, gsx = 2 * ox0 - 1 + sx;
      const bool inside = (unsigned)gsy < (unsigned)p.H0 && (unsigned)gsx < (unsigned)p.W0;
      const int idx = sx >> 1;
      unsigned char* ent = stem + ((sy * 2 + (sx & 1)) * kS2Idx + idx) * 64 + (lq & 1) * 8;
      const int sw = (idx >> 2) & 3;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 acc = Elem<T>::mma(wf0[j], a, bias0[j]);
        typedef __attribute__((ext_vector_type(4))) T t4;
        t4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f32(silu_f32(acc[e]));
        u32x2 ov = __builtin_bit_cast(u32x2, o);
        if (!inside) ov = u32x2{0u, 0u};  // layer 1's zero padding
        if (valid) *reinterpret_cast<u32x2*>(ent + (((j * 2 + (lq >> 1)) ^ sw) * 16)) = ov;
      }
    }
    __syncthreads();
    if (tile + t_step < t_end) request_patch(tile + t_step);  // the patch is dead: fetch the next one under phase C

    // ---- C: layer 1: every wave runs all 8 rows x 16 columns for its 16 couts ----
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = bias1;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int idx = lr + (q == 2 ? 1 : 0);
        const unsigned char* base = stem + ((r * 2 + (q == 1 ? 1 : 0)) * kS2Idx + idx) * 64 + ((lq ^ ((idx >> 2) & 3)) * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i)  // stem row 2*i + r
          acc[i] = Elem<T>::mma(wf1[r * 3 + q], *reinterpret_cast<const u32x4*>(base + i * (4 * kS2Idx * 64)), acc[i]);
      }

    // ---- epilogue: SiLU; the four waves assemble whole 128-byte pixel rows in the (then dead) stem buffer ----
    __syncthreads();  // every wave is done reading the stem buffer
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      typedef __attribute__((ext_vector_type(4))) T t4;
      t4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f32(silu_f32(acc[i][e]));
      *reinterpret_cast<u32x2*>(stem + (i * 16 + lr) * kS2EpPitch + (wave * 16 + lq * 4) * 2) = __builtin_bit_cast(u32x2, o);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int id = k * 256 + tid;
      const int px = id >> 3, cc = id & 7;
      const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
      if (oy < p.H1 && ox < p.W1) {
        const u32x4 val = *reinterpret_cast<const u32x4*>(stem + px * kS2EpPitch + cc * 16);
        *reinterpret_cast<u32x4*>(yg + ((size_t)(n * p.H1 + oy) * p.W1 + ox) * (size_t)p.ldy + cc * 8) = val;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- DY_F16X2 (split float16, include/dyolo.h; r05): the same fusion for the precision YOLO.predict runs by default.  Layer by layer the
// type wrote the 32-channel half-resolution map (3.4 GB at B = 256) and read it back: 1.5 + 1.3 ms of a 40 ms pass.  The kernel above with
//   * the gathered image taps split into (hi, lo) on the fly and three MFMAs per product (conv_stem_split_kernel's arithmetic);
//   * TWO stem buffers of the 16-bit kernel's layout — the hi halves and the lo halves of the same entries — so that every index, parity plane
//     and swizzle of phases B and C is the 16-bit kernel's, read twice;
//   * a tile of 4 x 16 layer-1 pixels (9 x 33 stem pixels, 19 patch rows): 16 KB of patch + 2 x 19.1 KB of stem = 55 KB, two workgroups per CU
//     as before (the 8-row tile would need 102 KB);
//   * layer 1's x_lo w_hi products in their own accumulators, joined with 2^-11 in the epilogue (conv_gemm_fk.hip), the rows scaled by the
//     inverse powers of two of the weight pack; outputs leave as [hi x 8 | lo x 8] groups through fp32 scratch.
// w0: conv_stem's split pack ([32][32] hi | lo | fp32[32] inverse scales); w1: DY_WLAYOUT_ROWS split rows [64][(tap, 8-channel group) x [hi x 8 | lo x 8]].
#ifndef DYOLO_L2E_BUILD
#ifndef X2_UNROLL_B
#define X2_UNROLL_B 1  // (2: 23 registers spilled at the 256 cap, same speed)
#endif
constexpr int kX2TH = 4, kX2TW = 16;
constexpr int kX2SH = 2 * kX2TH + 1, kX2SW = 2 * kX2TW + 1;  // 9 x 33 stem pixels
constexpr int kX2PH = 2 * kX2SH + 1;                          // 19 input rows
constexpr int kX2Plane = kX2PH * kS2Pitch;                    // 1292 floats
constexpr int kX2PatchBytes = 16 * 1024;                      // 3 * 19 * 68 floats = 15504 B, padded to 16 DMA instructions of 1 KiB
constexpr int kX2StemBytes = kX2SH * 2 * kS2Idx * 64;         // 19584 per half
constexpr int kX2EpPitch = 64 * 4 + 16;                       // fp32 row of a pixel's 64 channels
static_assert(3 * kX2Plane * 4 <= kX2PatchBytes && kX2TH * kX2TW * kX2EpPitch <= kX2StemBytes, "epilogue scratch aliases the hi stem buffer");

__device__ __forceinline__ void split4(const float (&v)[4], u32x2& hi, u32x2& lo) {  // common_hip.h: split8, four values
  typedef __attribute__((ext_vector_type(4))) f16_t h4;
  h4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x = __builtin_fminf(__builtin_fmaxf(v[e], -65504.f), 65504.f);
    const f16_t hh = __builtin_fabsf(x) < 6.103515625e-5f ? (f16_t)0.f : (f16_t)x;
    h[e] = hh;
    l[e] = (f16_t)((x - (float)hh) * kSplitScale);
  }
  hi = __builtin_bit_cast(u32x2, h);
  lo = __builtin_bit_cast(u32x2, l);
}

__global__ __launch_bounds__(256, 2) void stem2_split_kernel(const Stem2Args p, const float* __restrict__ sc1) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[kX2PatchBytes + 2 * kX2StemBytes];
  float* patch = reinterpret_cast<float*>(smem);
  unsigned char* stem_h = smem + kX2PatchBytes;
  unsigned char* stem_l = stem_h + kX2StemBytes;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lr = lane & 15;
  unsigned char* __restrict__ yb = reinterpret_cast<unsigned char*>(p.y);

  // ---- weights -> registers, once per workgroup ----
  const f16_t* w0h = reinterpret_cast<const f16_t*>(p.w0);
  const f16_t* w0l = w0h + 32 * 32;
  const float* sc0p = reinterpret_cast<const float*>(w0l + 32 * 32);
  u32x4 f0h[2], f0l[2], f0s[2];
  f32x4 bias0[2], sc0[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    f0h[j] = *reinterpret_cast<const u32x4*>(w0h + (size_t)(j * 16 + lr) * 32 + lq * 8);
    f0l[j] = *reinterpret_cast<const u32x4*>(w0l + (size_t)(j * 16 + lr) * 32 + lq * 8);
    f0s[j] = __builtin_bit_cast(u32x4, __builtin_bit_cast(f16x8, f0h[j]) * (f16_t)kSplitInv);  // exact: rows are scaled into [2^13, 2^14)
    bias0[j] = *reinterpret_cast<const f32x4*>(p.b0 + j * 16 + lq * 4);
    sc0[j] = *reinterpret_cast<const f32x4*>(sc0p + j * 16 + lq * 4);
  }
  u32x4 w1h[9], w1l[9];  // layer 1: wave w owns couts 16w .. 16w+15; per tap this lane's 8 channels (group lq) of row 16w + lr, hi and lo
  {
    const f16_t* row = reinterpret_cast<const f16_t*>(p.w1) + (size_t)(wave * 16 + lr) * 576;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      w1h[t] = *reinterpret_cast<const u32x4*>(row + (t * 4 + lq) * 16);
      w1l[t] = *reinterpret_cast<const u32x4*>(row + (t * 4 + lq) * 16 + 8);
    }
  }
  const f32x4 bias1 = *reinterpret_cast<const f32x4*>(p.b1 + wave * 16 + lq * 4);
  const f32x4 scl1 = *reinterpret_cast<const f32x4*>(sc1 + wave * 16 + lq * 4);
  int koff[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = lq * 8 + e;
    const int c = k / 9, r = (k - c * 9) / 3, q = k - c * 9 - r * 3;
    koff[e] = k < 27 ? c * kX2Plane + r * kS2Pitch + q : 0;  // (padding taps read the origin: their weights are zero)
  }
  auto request_patch = [&](int tile) {
    int t = tile;
    const int tx = t % p.tilesX;
    t /= p.tilesX;
    const int ty = t % p.tilesY;
    const int n = t / p.tilesY;
    const int gy0 = 4 * ty * kX2TH - 3, gx0 = 4 * tx * kX2TW - 4;
    const float* img = p.x + (size_t)n * 3 * p.H * p.W;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = (wave * 4 + k) * 64 + lane;
      const int row = (i * 3856) >> 16;         // i / 17 for i < 1792
      const int c = (row * 3450) >> 16;         // row / 19 for row < 58
      const int gy = gy0 + row - c * kX2PH, gx = gx0 + 4 * (i - row * 17);
      const bool ok = row < 3 * kX2PH && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      const float* src = ok ? img + (size_t)(c * p.H + gy) * p.W + gx : reinterpret_cast<const float*>(g_s2zero);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + (wave * 4 + k) * 1024), 16, 0, 0);
    }
  };

  int t_first, t_step, t_end;  // XCD-contiguous tile order, as above
  if (((int)gridDim.x & 7) == 0) {
    const int xcd = (int)blockIdx.x & 7, q = p.ntiles >> 3, r = p.ntiles & 7;
    const int start = xcd * q + (xcd < r ? xcd : r);
    t_first = start + ((int)blockIdx.x >> 3), t_step = (int)gridDim.x >> 3, t_end = start + q + (xcd < r ? 1 : 0);
  } else {
    t_first = (int)blockIdx.x, t_step = (int)gridDim.x, t_end = p.ntiles;
  }
  if (t_first < t_end) request_patch(t_first);
  for (int tile = t_first; tile < t_end; tile += t_step) {
    int t = tile;
    const int tx = t % p.tilesX;
    t /= p.tilesX;
    const int ty = t % p.tilesY;
    const int n = t / p.tilesY;
    const int oy0 = ty * kX2TH, ox0 = tx * kX2TW;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // patch landed in every wave; the previous tile's row stores are done reading the scratch

    // ---- B: stem pixels: 297 = 19 fragments of 16, five rounds of the four waves ----
#pragma unroll X2_UNROLL_B
    for (int it = 0; it < ((kX2SH * kX2SW + 15) / 16 + 3) / 4; ++it) {
      const int tt = (wave + 4 * it) * 16 + lr;
      const bool valid = tt < kX2SH * kX2SW;
      const int tc = valid ? tt : kX2SH * kX2SW - 1;
      const int sy = (tc * 1986) >> 16;  // tc / 33 for tc < 561
      const int sx = tc - sy * kX2SW;
      const float* org = patch + (2 * sy) * kS2Pitch + 2 * sx + 1;
      float g[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = org[koff[e]];
      u32x4 ah, al;
      split8(g, ah, al);
      const int gsy = 2 * oy0 - 1 + sy, gsx = 2 * ox0 - 1 + sx;
      const bool inside = (unsigned)gsy < (unsigned)p.H0 && (unsigned)gsx < (unsigned)p.W0;
      const int idx = sx >> 1;
      const int ent = ((sy * 2 + (sx & 1)) * kS2Idx + idx) * 64 + (lq & 1) * 8;
      const int sw = (idx >> 2) & 3;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 acc = Elem<f16_t>::mma(f0h[j], ah, f32x4{0.f, 0.f, 0.f, 0.f});
        acc = Elem<f16_t>::mma(f0l[j], ah, acc);
        acc = Elem<f16_t>::mma(f0s[j], al, acc);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f32(acc[e] * sc0[j][e] + bias0[j][e]);
        u32x2 oh, ol;
        split4(v, oh, ol);
        if (!inside) oh = u32x2{0u, 0u}, ol = u32x2{0u, 0u};  // layer 1's zero padding
        if (valid) {
          const int off = ent + (((j * 2 + (lq >> 1)) ^ sw) * 16);
          *reinterpret_cast<u32x2*>(stem_h + off) = oh;
          *reinterpret_cast<u32x2*>(stem_l + off) = ol;
        }
      }
    }
    __syncthreads();
    if (tile + t_step < t_end) request_patch(tile + t_step);  // the patch is dead: fetch the next one under phase C

    // ---- C: layer 1: every wave runs all 4 rows x 16 columns for its 16 couts ----
    f32x4 acc[kX2TH], accl[kX2TH];
#pragma unroll
    for (int i = 0; i < kX2TH; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}, accl[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int idx = lr + (q == 2 ? 1 : 0);
        const int base = ((r * 2 + (q == 1 ? 1 : 0)) * kS2Idx + idx) * 64 + ((lq ^ ((idx >> 2) & 3)) * 16);
#pragma unroll
        for (int i = 0; i < kX2TH; ++i) {  // stem row 2*i + r
          const u32x4 xh = *reinterpret_cast<const u32x4*>(stem_h + base + i * (4 * kS2Idx * 64));
          const u32x4 xl = *reinterpret_cast<const u32x4*>(stem_l + base + i * (4 * kS2Idx * 64));
          acc[i] = Elem<f16_t>::mma(w1h[r * 3 + q], xh, acc[i]);
          acc[i] = Elem<f16_t>::mma(w1l[r * 3 + q], xh, acc[i]);
          accl[i] = Elem<f16_t>::mma(w1h[r * 3 + q], xl, accl[i]);
        }
      }

    // ---- epilogue: scale, bias, SiLU in fp32; the four waves assemble whole pixel rows (fp32) in the then dead hi buffer ----
    __syncthreads();  // every wave is done reading the stem buffers
#pragma unroll
    for (int i = 0; i < kX2TH; ++i) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = silu_f32((acc[i][e] + accl[i][e] * kSplitInv) * scl1[e] + bias1[e]);
      *reinterpret_cast<f32x4*>(stem_h + (i * 16 + lr) * kX2EpPitch + (wave * 16 + lq * 4) * 4) = o;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int id = k * 256 + tid;
      const int px = id >> 3, cg = id & 7;
      const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
      if (oy < p.H1 && ox < p.W1) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(stem_h + px * kX2EpPitch + cg * 32);
        const f32x4 b = *reinterpret_cast<const f32x4*>(stem_h + px * kX2EpPitch + cg * 32 + 16);
        const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        u32x4 oh, ol;
        split8(f, oh, ol);
        unsigned char* dst = yb + (((size_t)(n * p.H1 + oy) * p.W1 + ox) * (size_t)p.ldy + (size_t)cg * 8) * 4;
        *reinterpret_cast<u32x4*>(dst) = oh;
        *reinterpret_cast<u32x4*>(dst + 16) = ol;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#endif

}  // namespace DY_NS

using namespace DY_NS;

#ifndef DYOLO_L2E_BUILD
extern "C" int32_t dy_stem2_fused_supported(int32_t cin, int32_t c0, int32_t c1, int32_t h, int32_t w, int32_t dtype) {
  return (cin == 3 && c0 == 32 && c1 == 64 && h > 0 && w > 0 && h % 4 == 0 && w % 4 == 0 && (dtype == DY_BF16 || dtype == DY_F16 || dtype == DY_F16X2)) ? 1 : 0;
}

namespace dy_l2e {
int32_t stem2_entry(const dy_stem2_desc* d, dy_stream_t stream);
}
namespace dy {
int32_t stem2_entry(const dy_stem2_desc* d, dy_stream_t stream);
}
extern "C" int32_t dy_stem2_fused(const dy_stem2_desc* d, dy_stream_t stream) {
  if (d != nullptr && d->act0 == DY_ACT_SILU_L2E && d->act1 == DY_ACT_SILU_L2E) {
    dy_stem2_desc c = *d;
    c.act0 = c.act1 = DY_ACT_SILU;
    return dy_l2e::stem2_entry(&c, stream);
  }
  return dy::stem2_entry(d, stream);
}
#endif

namespace DY_NS {
int32_t stem2_entry(const dy_stem2_desc* d, dy_stream_t stream) {
  DY_REQUIRE(d && d->x && d->w0 && d->b0 && d->w1 && d->b1 && d->y, DY_ERR_INVALID_ARG, "dy_stem2_fused: null descriptor or pointer");
  DY_REQUIRE(dy_stem2_fused_supported(3, 32, 64, d->h, d->w, d->dtype), DY_ERR_UNSUPPORTED,
             "dy_stem2_fused: built for 3 -> 32 -> 64 channels, 16-bit or split-float16 storage, image sides that are multiples of 4 (got %d x %d, dtype %d)",
             d->h, d->w, d->dtype);
  DY_REQUIRE(d->act0 == DY_ACT_SILU && d->act1 == DY_ACT_SILU, DY_ERR_UNSUPPORTED, "dy_stem2_fused: both layers must end in SiLU");
  DY_REQUIRE(d->n > 0 && d->ld_y >= 64 && d->ld_y % 8 == 0 && aligned16(d->y) && aligned16(d->x) && aligned16(d->w0) && aligned16(d->w1) &&
                 aligned16(d->b0) && aligned16(d->b1),
             DY_ERR_INVALID_ARG, "dy_stem2_fused: pointers must be 16-byte aligned, output pitch a multiple of 8 elements >= 64");
  Stem2Args a{};
  a.x = d->x;
  a.w0 = d->w0;
  a.b0 = d->b0;
  a.w1 = d->w1;
  a.b1 = d->b1;
  a.y = d->y;
  a.N = d->n;
  a.H = d->h;
  a.W = d->w;
  a.H0 = d->h / 2;
  a.W0 = d->w / 2;
  a.H1 = d->h / 4;
  a.W1 = d->w / 4;
  a.ldy = d->ld_y;
  a.act0 = d->act0;
  a.act1 = d->act1;
  const bool x2 = d->dtype == DY_F16X2;
#ifdef DYOLO_L2E_BUILD
  DY_REQUIRE(!x2, DY_ERR_UNSUPPORTED, "dy_stem2_fused: DY_F16X2 runs in the reference's activation units (DY_ACT_SILU)");
#endif
  DY_REQUIRE(!x2 || (d->w1_scale && aligned16(d->w1_scale) && d->ld_y % 8 == 0), DY_ERR_INVALID_ARG, "dy_stem2_fused: DY_F16X2 needs w1_scale (fp32[64], 16-byte aligned) and an output pitch in whole groups of 8 channels");
  a.tilesX = (a.W1 + kS2TW - 1) / kS2TW;
  a.tilesY = (a.H1 + (x2 ? 4 : kS2TH) - 1) / (x2 ? 4 : kS2TH);
  const long long nt = (long long)a.N * a.tilesY * a.tilesX;
  DY_REQUIRE(nt < (1ll << 31) && (long long)a.N * 3 * a.H * a.W < (1ll << 40), DY_ERR_INVALID_ARG, "dy_stem2_fused: batch too large");
  a.ntiles = (int)nt;
  int grid = 512;  // two workgroups per CU
  if (grid > a.ntiles) grid = a.ntiles >= 8 ? (a.ntiles & ~7) : a.ntiles;  // (a multiple of 8 keeps the XCD-contiguous tile order)
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#ifndef DYOLO_L2E_BUILD
  if (x2) {
    hipLaunchKernelGGL(stem2_split_kernel, dim3((unsigned)grid), dim3(256), 0, st, a, d->w1_scale);
    return check_launch("stem2_split_kernel");
  }
#endif
  if (d->dtype == DY_BF16)
    hipLaunchKernelGGL((stem2_fused_kernel<bf16_t>), dim3((unsigned)grid), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((stem2_fused_kernel<f16_t>), dim3((unsigned)grid), dim3(256), 0, st, a);
  return check_launch("stem2_fused_kernel");
}
}  // namespace DY_NS
