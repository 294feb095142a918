// Fused tail of the Detect head: the two plain 1x1 convolutions (box bins, class logits) + DFL decode + sigmoid
// + the NMS candidate filter, in one pass per 64-anchor tile, so the (batch, 4*reg_max + nc, A) fp32 logits never
// reach HBM (unfused they are written once by two conv launches and read once by dy_detect_decode: 608 B per
// anchor of traffic against 256 B of bf16 input + 56 B of output here).
// Reference: nn/modules/head.py:43-57 (cv2[i][2], cv3[i][2]), :69-70 (cat), :100-131 (_inference),
// block.py:58-76 (DFL), utils/tal.py:333-357 (make_anchors, dist2bbox); filter: utils/ops.py:250,290-295.
//
// One wave per workgroup; a workgroup walks kGroup consecutive 64-anchor tiles of one (image, level).
//  * weights of both convs (DY_WLAYOUT_FRAG1X1 fragments) + biases are copied to LDS once per workgroup;
//  * per tile the wave loads the two branch inputs straight from global memory in MFMA operand order
//    (lane (lr, lq) = pixel lr of a 16-pixel group, 16-byte k-chunk lq) and keeps them in registers;
//  * MFMA with the WEIGHTS as the A operand: D[cout][pixel], so a lane ends up with 4 consecutive output channels
//    of one pixel -> one 16-byte LDS write into that pixel's logit row (pitch chosen bank-conflict free);
//  * the next tile's inputs are requested right after the last MFMA, then the decode phase (one lane per anchor,
//    identical arithmetic to detect_decode.hip) runs from LDS while those loads are in flight.
#include "common_hip.h"
#include "nms_ws.h"

namespace dy {

constexpr int kHeadTile = 64;  // anchors per tile = one wave, lane per anchor in the decode phase
constexpr int kGroup = 8;      // tiles per workgroup (amortises the weight copy: 10 KB per 128 KB of input)

struct HeadArgs {
  const void* xb[DY_MAX_LEVELS];
  const void* xc[DY_MAX_LEVELS];
  const void* wb[DY_MAX_LEVELS];
  const void* wc[DY_MAX_LEVELS];
  const float* bb[DY_MAX_LEVELS];
  const float* bc[DY_MAX_LEVELS];
  int ldb[DY_MAX_LEVELS], ldc[DY_MAX_LEVELS], h[DY_MAX_LEVELS], w[DY_MAX_LEVELS], a0[DY_MAX_LEVELS + 1], g0[DY_MAX_LEVELS + 1];
  float stride[DY_MAX_LEVELS];
  int n_levels, batch, nc, A, groupsPerImg, nfc, mtc, pitch;  // nfc: fragments per k-group in the packed cls weights
  float* out;
  int* counts;
  unsigned long long* keys;
  unsigned short* cls;
  int P;
  float conf;
  const uint8_t* cmask;
};

// NKB / NKC: k-groups (4 chunks of 16 B) of the box / class branch input channels.
template <typename T, int NKB, int NKC>
__global__ __launch_bounds__(64) void detect_head_kernel(const HeadArgs p) {
  constexpr int E = Elem<T>::EPC, KC = 4 * E, REG_MAX = 16, NB = 4 * REG_MAX;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  const int lane = threadIdx.x, lr = lane & 15, lq = lane >> 4;
  const int b = blockIdx.x / p.groupsPerImg;
  const int gr = blockIdx.x - b * p.groupsPerImg;
  int l = 0;
#pragma unroll
  for (int i = 1; i < DY_MAX_LEVELS; ++i)
    if (i < p.n_levels && gr >= p.g0[i]) l = i;
  const int hw = p.h[l] * p.w[l], wl = p.w[l];
  const int tiles = (hw + kHeadTile - 1) / kHeadTile;
  const int t_begin = (gr - p.g0[l]) * kGroup;
  const int t_end = (t_begin + kGroup < tiles) ? t_begin + kGroup : tiles;
  const int pitch = p.pitch;

  // ---- LDS: [box weight fragments][cls weight fragments][bias box 64 | bias cls mtc*16][rows 64 x pitch] ----
  u32x4* wB = reinterpret_cast<u32x4*>(dyn_smem);       // NKB*4 fragments of 64 chunks
  u32x4* wC = wB + NKB * 4 * 64;                        // NKC*nfc fragments
  float* bias = reinterpret_cast<float*>(wC + NKC * p.nfc * 64);
  float* rows = bias + NB + p.mtc * 16;
  {
    const u32x4* gb = reinterpret_cast<const u32x4*>(p.wb[l]);
    const u32x4* gc = reinterpret_cast<const u32x4*>(p.wc[l]);
#pragma unroll
    for (int i = 0; i < NKB * 4; ++i) wB[i * 64 + lane] = gb[i * 64 + lane];
    for (int i = 0; i < NKC * p.nfc; ++i) wC[i * 64 + lane] = gc[i * 64 + lane];
    bias[lane] = p.bb[l][lane];
    if (lane < p.mtc * 16) bias[NB + lane] = p.bc[l][lane];
    if (lane + 64 < p.mtc * 16) bias[NB + lane + 64] = p.bc[l][lane + 64];
  }

  const T* xb = reinterpret_cast<const T*>(p.xb[l]) + (size_t)b * hw * p.ldb[l] + lq * E;
  const T* xc = reinterpret_cast<const T*>(p.xc[l]) + (size_t)b * hw * p.ldc[l] + lq * E;
  const int ldb = p.ldb[l], ldc = p.ldc[l];
  u32x4 fb[NKB][4], fc[NKC][4];  // [k-group][16-pixel group]: this lane's chunk of pixel (pt*16 + lr)

  auto load_tile = [&](int t) {
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      int px = t * kHeadTile + pt * 16 + lr;
      px = px < hw ? px : hw - 1;  // ragged last tile: duplicate the last pixel, its lanes are masked later
      const T* rb = xb + (size_t)px * ldb;
      const T* rc = xc + (size_t)px * ldc;
#pragma unroll
      for (int k = 0; k < NKB; ++k) fb[k][pt] = *reinterpret_cast<const u32x4*>(rb + k * KC);
#pragma unroll
      for (int k = 0; k < NKC; ++k) fc[k][pt] = *reinterpret_cast<const u32x4*>(rc + k * KC);
    }
  };

  load_tile(t_begin);
  __syncthreads();  // weights/bias visible (single wave: orders LDS writes before reads)

  for (int t = t_begin; t < t_end; ++t) {
    // ---- MFMA phase: logits of 64 anchors -> LDS rows ----
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {  // box: 4 cout tiles = the 4 sides
      const f32x4 bi = *reinterpret_cast<const f32x4*>(bias + mt * 16 + lq * 4);
      f32x4 acc[4] = {bi, bi, bi, bi};
#pragma unroll
      for (int k = 0; k < NKB; ++k) {
        const u32x4 a = wB[(k * 4 + mt) * 64 + lane];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) acc[pt] = Elem<T>::mma(a, fb[k][pt], acc[pt]);
      }
      mfma_epilogue_fence<T>();
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) *reinterpret_cast<f32x4*>(rows + (pt * 16 + lr) * pitch + mt * 16 + lq * 4) = acc[pt];
    }
    for (int mt = 0; mt < p.mtc; ++mt) {  // classes
      const f32x4 bi = *reinterpret_cast<const f32x4*>(bias + NB + mt * 16 + lq * 4);
      f32x4 acc[4] = {bi, bi, bi, bi};
#pragma unroll
      for (int k = 0; k < NKC; ++k) {
        const u32x4 a = wC[(k * p.nfc + mt) * 64 + lane];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) acc[pt] = Elem<T>::mma(a, fc[k][pt], acc[pt]);
      }
      mfma_epilogue_fence<T>();
      if (mt * 16 + lq * 4 < p.nc) {
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
          *reinterpret_cast<f32x4*>(rows + (pt * 16 + lr) * pitch + NB + mt * 16 + lq * 4) = acc[pt];
      }
    }
    if (t + 1 < t_end) load_tile(t + 1);  // in flight during the decode phase
    __syncthreads();

    // ---- decode phase: lane = anchor (same arithmetic as detect_decode_kernel) ----
    const int al = t * kHeadTile + lane;
    const bool valid = al < hw;
    const int a = p.a0[l] + al;
    float best = 0.f;
    int bj = 0;
    if (valid) {
      const float* r = rows + lane * pitch;
      const int gy = al / wl, gx = al - gy * wl;
      float dist[4];
#pragma unroll
      for (int side = 0; side < 4; ++side) {
        float v[REG_MAX];
#pragma unroll
        for (int i = 0; i < REG_MAX; i += 4) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(r + side * REG_MAX + i);
          v[i] = q[0], v[i + 1] = q[1], v[i + 2] = q[2], v[i + 3] = q[3];
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
      const float* cl = r + NB;
      for (int c = 0; c < p.nc; ++c) {
        const float pr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cl[c] * -1.4426950408889634f));
        o[(size_t)(4 + c) * p.A] = pr;
        if (c == 0 || pr > best) {  // first arg-max, as cls.max(1) (ops.py:290)
          best = pr;
          bj = c;
        }
      }
    }
    if (p.keys != nullptr) {
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
    __syncthreads();  // rows are rewritten by the next tile
  }
}

// DY_F16X2 (split float16, include/dyolo.h) form of the same tail: a k-group is 32 channels = the 128 bytes of a pixel row that hold four
// (hi, lo) chunk pairs; lane (lr, lq) loads pair lq of its pixel.  Weights: the float16 FRAG1X1 image of the hi halves followed by the image
// of the lo halves (rows scaled into [2^13, 2^14) first); the bias array is followed by the inverse row scales.  Three MFMAs per fragment
// pair — w_hi x_hi + w_lo x_hi + (w_hi 2^-11) x_lo —, then acc * scale + bias into the logit rows; the decode phase is the one above.
// Replaces, for the type, two flat-K launches per level with fp32 outputs + dy_detect_decode: 7.6 GB of HBM traffic at the P2 level -> 3.7 GB.
template <int NKB, int NKC>
__global__ __launch_bounds__(64) void detect_head_split_kernel(const HeadArgs p) {
  constexpr int REG_MAX = 16, NB = 4 * REG_MAX;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  const int lane = threadIdx.x, lr = lane & 15, lq = lane >> 4;
  const int b = blockIdx.x / p.groupsPerImg;
  const int gr = blockIdx.x - b * p.groupsPerImg;
  int l = 0;
#pragma unroll
  for (int i = 1; i < DY_MAX_LEVELS; ++i)
    if (i < p.n_levels && gr >= p.g0[i]) l = i;
  const int hw = p.h[l] * p.w[l], wl = p.w[l];
  const int tiles = (hw + kHeadTile - 1) / kHeadTile;
  const int t_begin = (gr - p.g0[l]) * kGroup;
  const int t_end = (t_begin + kGroup < tiles) ? t_begin + kGroup : tiles;
  const int pitch = p.pitch;
  const int nbc = p.mtc * 16;  // padded class rows

  // ---- LDS: [box hi][box lo][cls hi][cls lo][bias box 64 | scale box 64 | bias cls | scale cls][rows 64 x pitch] ----
  u32x4* wBh = reinterpret_cast<u32x4*>(dyn_smem);
  u32x4* wBl = wBh + NKB * 4 * 64;
  u32x4* wCh = wBl + NKB * 4 * 64;
  u32x4* wCl = wCh + NKC * p.nfc * 64;
  float* bias = reinterpret_cast<float*>(wCl + NKC * p.nfc * 64);
  float* rows = bias + 2 * NB + 2 * nbc;
  {
    const u32x4* gb = reinterpret_cast<const u32x4*>(p.wb[l]);
    const u32x4* gc = reinterpret_cast<const u32x4*>(p.wc[l]);
#pragma unroll
    for (int i = 0; i < 2 * NKB * 4; ++i) wBh[i * 64 + lane] = gb[i * 64 + lane];
    for (int i = 0; i < 2 * NKC * p.nfc; ++i) wCh[i * 64 + lane] = gc[i * 64 + lane];
    bias[lane] = p.bb[l][lane], bias[NB + lane] = p.bb[l][NB + lane];
    for (int i = lane; i < 2 * nbc; i += 64) bias[2 * NB + i] = p.bc[l][i];
  }
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.xb[l]) + ((size_t)b * hw * p.ldb[l]) * 4 + lq * 32;
  const unsigned char* xc = reinterpret_cast<const unsigned char*>(p.xc[l]) + ((size_t)b * hw * p.ldc[l]) * 4 + lq * 32;
  const size_t ldb = (size_t)p.ldb[l] * 4, ldc = (size_t)p.ldc[l] * 4;
  u32x4 fbh[NKB][4], fbl[NKB][4], fch[NKC][4], fcl[NKC][4];

  auto load_tile = [&](int t) {
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      int px = t * kHeadTile + pt * 16 + lr;
      px = px < hw ? px : hw - 1;
      const unsigned char* rb = xb + (size_t)px * ldb;
      const unsigned char* rc = xc + (size_t)px * ldc;
#pragma unroll
      for (int k = 0; k < NKB; ++k) fbh[k][pt] = *reinterpret_cast<const u32x4*>(rb + k * 128), fbl[k][pt] = *reinterpret_cast<const u32x4*>(rb + k * 128 + 16);
#pragma unroll
      for (int k = 0; k < NKC; ++k) fch[k][pt] = *reinterpret_cast<const u32x4*>(rc + k * 128), fcl[k][pt] = *reinterpret_cast<const u32x4*>(rc + k * 128 + 16);
    }
  };

  load_tile(t_begin);
  __syncthreads();

  for (int t = t_begin; t < t_end; ++t) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int k = 0; k < NKB; ++k) {
        const u32x4 ah = wBh[(k * 4 + mt) * 64 + lane], al = wBl[(k * 4 + mt) * 64 + lane];
        const f16x8 sv = __builtin_bit_cast(f16x8, ah) * (f16_t)kSplitInv;
        const u32x4 as = __builtin_bit_cast(u32x4, sv);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
          acc[pt] = Elem<f16_t>::mma(ah, fbh[k][pt], acc[pt]);
          acc[pt] = Elem<f16_t>::mma(al, fbh[k][pt], acc[pt]);
          acc[pt] = Elem<f16_t>::mma(as, fbl[k][pt], acc[pt]);
        }
      }
      const f32x4 bi = *reinterpret_cast<const f32x4*>(bias + mt * 16 + lq * 4), sc = *reinterpret_cast<const f32x4*>(bias + NB + mt * 16 + lq * 4);
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
        *reinterpret_cast<f32x4*>(rows + (pt * 16 + lr) * pitch + mt * 16 + lq * 4) =
            f32x4{acc[pt][0] * sc[0] + bi[0], acc[pt][1] * sc[1] + bi[1], acc[pt][2] * sc[2] + bi[2], acc[pt][3] * sc[3] + bi[3]};
    }
    for (int mt = 0; mt < p.mtc; ++mt) {
      f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int k = 0; k < NKC; ++k) {
        const u32x4 ah = wCh[(k * p.nfc + mt) * 64 + lane], al = wCl[(k * p.nfc + mt) * 64 + lane];
        const f16x8 sv = __builtin_bit_cast(f16x8, ah) * (f16_t)kSplitInv;
        const u32x4 as = __builtin_bit_cast(u32x4, sv);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
          acc[pt] = Elem<f16_t>::mma(ah, fch[k][pt], acc[pt]);
          acc[pt] = Elem<f16_t>::mma(al, fch[k][pt], acc[pt]);
          acc[pt] = Elem<f16_t>::mma(as, fcl[k][pt], acc[pt]);
        }
      }
      if (mt * 16 + lq * 4 < p.nc) {
        const f32x4 bi = *reinterpret_cast<const f32x4*>(bias + 2 * NB + mt * 16 + lq * 4), sc = *reinterpret_cast<const f32x4*>(bias + 2 * NB + nbc + mt * 16 + lq * 4);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
          *reinterpret_cast<f32x4*>(rows + (pt * 16 + lr) * pitch + NB + mt * 16 + lq * 4) =
              f32x4{acc[pt][0] * sc[0] + bi[0], acc[pt][1] * sc[1] + bi[1], acc[pt][2] * sc[2] + bi[2], acc[pt][3] * sc[3] + bi[3]};
      }
    }
    if (t + 1 < t_end) load_tile(t + 1);
    __syncthreads();

    // ---- decode phase: lane = anchor (the arithmetic of detect_head_kernel / detect_decode_kernel) ----
    const int al = t * kHeadTile + lane;
    const bool valid = al < hw;
    const int a = p.a0[l] + al;
    float best = 0.f;
    int bj = 0;
    if (valid) {
      const float* r = rows + lane * pitch;
      const int gy = al / wl, gx = al - gy * wl;
      float dist[4];
#pragma unroll
      for (int side = 0; side < 4; ++side) {
        float v[REG_MAX];
#pragma unroll
        for (int i = 0; i < REG_MAX; i += 4) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(r + side * REG_MAX + i);
          v[i] = q[0], v[i + 1] = q[1], v[i + 2] = q[2], v[i + 3] = q[3];
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
      const float* cl = r + NB;
      for (int c = 0; c < p.nc; ++c) {
        const float pr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cl[c] * -1.4426950408889634f));
        o[(size_t)(4 + c) * p.A] = pr;
        if (c == 0 || pr > best) {
          best = pr;
          bj = c;
        }
      }
    }
    if (p.keys != nullptr) {
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
    __syncthreads();
  }
}

template <int NKB, int NKC>
static int launch_head_split(const HeadArgs& a, size_t smem, hipStream_t st) {
  static const hipError_t once = hipFuncSetAttribute((const void*)detect_head_split_kernel<NKB, NKC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)once;
  hipLaunchKernelGGL((detect_head_split_kernel<NKB, NKC>), dim3((unsigned)(a.batch * a.groupsPerImg)), dim3(64), smem, st, a);
  return check_launch("detect_head_split_kernel");
}

template <typename T, int NKB, int NKC>
static int launch_head(const HeadArgs& a, size_t smem, hipStream_t st) {
  static const hipError_t once =
      hipFuncSetAttribute((const void*)detect_head_kernel<T, NKB, NKC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)once;
  hipLaunchKernelGGL((detect_head_kernel<T, NKB, NKC>), dim3((unsigned)(a.batch * a.groupsPerImg)), dim3(64), smem, st, a);
  return check_launch("detect_head_kernel");
}

template <typename T>
static int dispatch_head(const HeadArgs& a, int nkb, int nkc, size_t smem, hipStream_t st) {
  constexpr int NKB = sizeof(T) == 4 ? 4 : 2;  // 4*reg_max = 64 input channels of the box branch
  if (nkb != NKB) return 1;
  if constexpr (sizeof(T) == 4) {
    switch (nkc) {
      case 4: return launch_head<T, NKB, 4>(a, smem, st);
      case 5: return launch_head<T, NKB, 5>(a, smem, st);
      default: return 1;
    }
  } else {
    switch (nkc) {
      case 2: return launch_head<T, NKB, 2>(a, smem, st);
      case 3: return launch_head<T, NKB, 3>(a, smem, st);
      case 4: return launch_head<T, NKB, 4>(a, smem, st);
      case 5: return launch_head<T, NKB, 5>(a, smem, st);
      default: return 1;
    }
  }
}

}  // namespace dy

using namespace dy;

extern "C" int32_t dy_detect_head_decode_supported(int32_t c_box, int32_t c_cls, int32_t nc, int32_t reg_max, int32_t dtype) {
  if (reg_max != 16 || nc < 1 || nc > 128) return 0;
  if (dtype == DY_F16X2) return (c_box == 64 && c_cls % 32 == 0 && c_cls >= 64 && c_cls <= 128) ? 1 : 0;  // k-groups of 32 channels: 2 for the box branch, 2-4 for the class branch
  const int esz = dtype_size_no_fp8(dtype);
  if (esz == 0) return 0;
  const int kc = 4 * (16 / esz);
  if (c_box % kc || c_cls % kc) return 0;  // whole k-groups only: no reads past the channels (garbage * 0 could be NaN)
  const int nkb = c_box / kc, nkc = c_cls / kc;
  if (nkb != (esz == 4 ? 4 : 2)) return 0;
  if (esz == 4) return nkc == 4 || nkc == 5;
  return nkc >= 2 && nkc <= 5;
}

extern "C" int32_t dy_detect_head_decode(const dy_head_decode_desc* d, dy_stream_t stream) {
  DY_REQUIRE(d && d->out, DY_ERR_INVALID_ARG, "dy_detect_head_decode: null descriptor/out");
  DY_REQUIRE(d->n_levels >= 1 && d->n_levels <= DY_MAX_LEVELS && d->batch > 0, DY_ERR_INVALID_ARG,
             "dy_detect_head_decode: bad n_levels/batch");
  DY_REQUIRE(dy_detect_head_decode_supported(d->c_box, d->c_cls, d->nc, d->reg_max, d->dtype), DY_ERR_UNSUPPORTED,
             "dy_detect_head_decode: shape c_box %d c_cls %d nc %d reg_max %d dtype %d not built (use dy_conv2d_nhwc + "
             "dy_detect_decode)", d->c_box, d->c_cls, d->nc, d->reg_max, d->dtype);
  const bool split = d->dtype == DY_F16X2;
  const int esz = split ? 4 : dtype_size_no_fp8(d->dtype), epc = 16 / esz, kc = split ? 32 : 4 * epc;
  HeadArgs a{};
  int A = 0, G = 0;
  for (int i = 0; i < d->n_levels; ++i) {
    DY_REQUIRE(d->x_box[i] && d->x_cls[i] && d->w_box[i] && d->w_cls[i] && d->b_box[i] && d->b_cls[i] && d->h[i] > 0 && d->w[i] > 0,
               DY_ERR_INVALID_ARG, "dy_detect_head_decode: level %d has a null pointer or empty map", i);
    DY_REQUIRE(d->ld_box[i] >= d->c_box && d->ld_cls[i] >= d->c_cls && d->ld_box[i] % epc == 0 && d->ld_cls[i] % epc == 0 &&
                   aligned16(d->x_box[i]) && aligned16(d->x_cls[i]) && aligned16(d->w_box[i]) && aligned16(d->w_cls[i]),
               DY_ERR_INVALID_ARG, "dy_detect_head_decode: level %d pitches must cover the channels in 16-byte chunks, bases 16B aligned", i);
    a.xb[i] = d->x_box[i], a.xc[i] = d->x_cls[i], a.wb[i] = d->w_box[i], a.wc[i] = d->w_cls[i];
    a.bb[i] = d->b_box[i], a.bc[i] = d->b_cls[i];
    a.ldb[i] = d->ld_box[i], a.ldc[i] = d->ld_cls[i], a.h[i] = d->h[i], a.w[i] = d->w[i], a.stride[i] = d->stride[i];
    a.a0[i] = A, a.g0[i] = G;
    const int hw = d->h[i] * d->w[i], tiles = (hw + kHeadTile - 1) / kHeadTile;
    A += hw;
    G += (tiles + kGroup - 1) / kGroup;
  }
  a.a0[d->n_levels] = A, a.g0[d->n_levels] = G;
  a.n_levels = d->n_levels, a.batch = d->batch, a.nc = d->nc, a.A = A, a.groupsPerImg = G;
  a.mtc = (d->nc + 15) / 16;
  a.nfc = d->nc > 64 ? 8 : (d->nc > 16 ? 4 : 1);  // DY_WLAYOUT_FRAG1X1: BN/16 fragments per k-group
  int pitch = 4 * d->reg_max + (d->nc + 3) / 4 * 4;
  if (((pitch / 4) & 1) == 0) pitch += 4;  // odd count of 16-byte units per row: conflict-free row writes and reads
  a.pitch = pitch;
  a.out = d->out;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->nms_workspace) {
    DY_REQUIRE(d->nc <= 65535, DY_ERR_UNSUPPORTED, "dy_detect_head_decode: nc %d > 65535", d->nc);
    DY_REQUIRE(aligned16(d->nms_workspace) && d->nms_workspace_bytes >= (int64_t)nms_ws_bytes(d->batch, A), DY_ERR_WORKSPACE,
               "dy_detect_head_decode: nms_workspace too small or misaligned (need %lld bytes)", (long long)nms_ws_bytes(d->batch, A));
    const NmsWs w = nms_ws_layout(d->nms_workspace, d->batch, A);
    a.counts = w.counts, a.keys = w.keys, a.cls = w.cls, a.P = w.P;
    a.conf = d->conf_thres;
    a.cmask = d->classes_mask;
    zero_async(w.counts, (size_t)d->batch * 4, st);
  }
  const int nkb = (d->c_box + kc - 1) / kc, nkc = (d->c_cls + kc - 1) / kc;
  const size_t smem = (size_t)(nkb * 4 + nkc * a.nfc) * 1024 * (split ? 2 : 1) + (size_t)(4 * d->reg_max + a.mtc * 16) * 4 * (split ? 2 : 1) + (size_t)kHeadTile * pitch * 4;
  DY_REQUIRE(smem <= 160 * 1024, DY_ERR_UNSUPPORTED, "dy_detect_head_decode: %zu bytes of LDS needed", smem);
  int rc = 1;
  if (split) {
    switch (nkc) {
      case 2: rc = launch_head_split<2, 2>(a, smem, st); break;
      case 3: rc = launch_head_split<2, 3>(a, smem, st); break;
      case 4: rc = launch_head_split<2, 4>(a, smem, st); break;
      default: break;
    }
    DY_REQUIRE(rc <= 0, DY_ERR_UNSUPPORTED, "dy_detect_head_decode: no split-float16 kernel for this shape");
    return rc;
  }
  switch (d->dtype) {
    case DY_BF16: rc = dispatch_head<bf16_t>(a, nkb, nkc, smem, st); break;
    case DY_F16: rc = dispatch_head<f16_t>(a, nkb, nkc, smem, st); break;
    case DY_F32: rc = dispatch_head<float>(a, nkb, nkc, smem, st); break;
    default: break;
  }
  DY_REQUIRE(rc <= 0, DY_ERR_UNSUPPORTED, "dy_detect_head_decode: no kernel for this shape");
  return rc;
}
