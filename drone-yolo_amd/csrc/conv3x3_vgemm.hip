// 3x3 stride-1 convolution of the deep layers (Cin >= 128) as a GEMM over a VIRTUAL flat pixel index, so that a
// workgroup stages each input pixel ONCE per channel chunk instead of once per tap.
//
// conv_gemm_glds.hip gathers a fresh 128-pixel x 64-channel A tile for every (tap, chunk) K-step: 9x the input bytes
// through the L2 -> LDS DMA, which is what bounds it (~40 GB/s per CU measured, 30 % MFMA utilisation).  Here the batch
// is viewed as ONE tall image: every image gets a zero row below it and every row a zero column to its right,
//     v = (n * (H + 1) + y) * (W + 1) + x,      Wv = W + 1, Hv = H + 1,
// so the left / right / top / bottom zero padding of every pixel IS one of those shared zero slots, and tap (r, q) of
// output pixel v reads input slot v + (r - 1) * Wv + (q - 1): a constant offset.  A tile = 256 consecutive virtual
// pixels; its input halo = 256 + 2 * Wv + 2 consecutive slots, staged per 64-channel chunk (128 B per slot) and read by
// all nine taps at shifted LDS addresses.  Virtual pad pixels are computed and dropped (1 / Wv + 1 / Hv: 5 % at 40 x 40).
// The input tensor itself stays dense NHWC: pad slots are DMA'd from a zero page.
//
// Per K-step (one tap of one chunk) a workgroup moves 16 KiB of weights + 1/9 of the halo (~5 KiB) for 256 x 128 x 64
// MACs: 2.6x fewer DMA bytes per flop than the per-tap gather.
//
// Workgroup = 512 threads = 8 waves as 4 (M) x 2 (N); wave tile 64 pixels x 64 couts, weights as the MFMA A operand.
// LDS: 2 halo stages x 48 KiB (384 slots: maps up to 62 wide; 56 KiB / 448 slots up to 94 wide) + 3 weight stages x 16 KiB = 144 /
// 160 KiB.  Weight rows are swizzled as in conv_gemm_glds.hip (chunk ^ ((row >> 1) & 7): fragment bases are multiples of 16).  The
// HALO is read at nine different shifts, so its swizzle has to be conflict-free at EVERY fragment base: chunk ^ (slot & 6) (r04).
// A ds_read_b128 is served in groups of 16 lanes — rows lr in {0..3, 12..15} of lane quarter q and lr in {4..11} of quarter q ^ 1 —
// and a row's bank half is slot & 1, so the 8 same-parity rows of a group must land on 8 different chunks: slot & 6 gives the 4
// rows two apart 4 different values, and the rows EIGHT apart, which share it, always sit in different quarters (bit 0 of the chunk
// differs).  With (slot >> 1) & 7, right for aligned bases only, every shifted read took 8 LDS cycles instead of 4
// (SQ_LDS_BANK_CONFLICT = 23 % of the LDS cycles, profiles/r03_pmc_vgemm16_128x128_3x3_40.txt).
// One continuous software pipeline across K-steps AND tiles (persistent grid): weights run two steps ahead, the halo one
// chunk ahead, behind counted s_waitcnt vmcnt(N) and raw s_barriers; the only bubble is the epilogue.
#include "common_hip.h"
#include "conv_args.h"

namespace DY_NS {

__device__ __attribute__((aligned(256))) const unsigned int g_vzero_page[64] = {0};

struct VGeom {
  int Wv, Hv, V;      // virtual row length, rows per image, total virtual pixels
  int S;              // halo slots per tile: 256 + 2 * Wv + 2
  int tilesM, nchunk; // 256-pixel tiles, 128-byte channel chunks
  FastDiv dWv, dHv;
};

// PA: halo pieces per wave and chunk (8 * PA * 8 slots per stage), BST: weight stages (weights run BST - 1 steps ahead),
// PREF: fetch the next step's pixel fragments into registers during this step's MFMAs.
template <typename T, int PA, int BST, bool PREF, int DBG = 0>
__global__ __launch_bounds__(512) void conv3x3_vgemm_kernel(const ConvArgs p, const VGeom g) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BKE = 8 * EPC;
  constexpr int BM = 256, BN = 128, NW = 8;
  constexpr int A_BYTES = PA * NW * 1024;
  constexpr int B_BYTES = BN * 128;           // 16384
  constexpr int PB = BN / 8 / NW;             // 2
  constexpr int D = BST - 1;                  // weight prefetch distance in steps
  constexpr int NFR = 4;                      // cout fragments per wave
  constexpr int EG0 = 128 / (16 * (int)sizeof(T));
  constexpr int EG = NFR < EG0 ? NFR : EG0;
  constexpr int CPP = EG * (int)sizeof(T);
  constexpr int EP_PITCH = 128 + 16;
  static_assert(32 * EP_PITCH * NW <= A_BYTES, "epilogue scratch must fit one halo stage");
  static_assert(2 * A_BYTES + BST * B_BYTES <= 160 * 1024, "LDS budget");

  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * A_BYTES + BST * B_BYTES];
  unsigned char* const smA = smem;
  unsigned char* const smB = smem + 2 * A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lq = lane >> 4, lr = lane & 15;
  const int prow = lane >> 3;
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
  const T* zp = reinterpret_cast<const T*>(g_vzero_page) + (lane & 7) * EPC;
  T* __restrict__ yg = reinterpret_cast<T*>(p.y);
  const T* __restrict__ rg = reinterpret_cast<const T*>(p.res);
  const int G = (int)gridDim.x;
  const int NC = g.nchunk;

  // real pixel index of virtual pixel v, or -1 for a pad slot / out of range
  auto real_px = [&](int v) -> int {
    if (v < 0 || v >= g.V) return -1;
    const unsigned t = fastdiv((unsigned)v, g.dWv);
    const int x = v - (int)t * g.Wv;
    const unsigned n = fastdiv(t, g.dHv);
    const int y = (int)t - (int)n * g.Hv;
    if (x >= p.W || y >= p.H) return -1;
    return ((int)n * p.H + y) * p.W + x;
  };

  // ---- issue side: halo (one chunk ahead) and weights (D steps ahead), each with its own tile cursor.  Every issue
  // point ALWAYS emits its full number of DMA instructions (from the zero page once the tiles are exhausted), so that
  // the counted s_waitcnt vmcnt(N) below stay exact ----
  int a_off[PA];          // element offset of this lane's source chunk for each of its pieces, -1 = zero page
  int ia_tile, ia_c;      // next halo chunk to issue
  auto setup_a = [&](int tile) {
    const int tM = tile / p.tilesN;
    const int vstart = tM * BM - g.Wv - 1;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int s = (i * NW + wave) * 8 + prow;
      const int px = s < g.S ? real_px(vstart + s) : -1;
      a_off[i] = px < 0 ? -1 : px * p.ldx + ((lane & 7) ^ (s & 6)) * EPC;  // halo swizzle: see the header (r04)
    }
  };
  auto issue_a = [&](int stage) {
    unsigned char* sa = smA + stage * A_BYTES;
    const bool live = ia_tile < p.nblk;
    const int cofs = ia_c * BKE;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const T* src = (!live || a_off[i] < 0) ? zp : xg + (size_t)(unsigned)(a_off[i] + cofs);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sa + (i * NW + wave) * 1024), 16, 0, 0);
    }
    if (live && ++ia_c == NC) {
      ia_c = 0;
      ia_tile += G;
      if (ia_tile < p.nblk) setup_a(ia_tile);
    }
  };
  const T* b_ptr[PB];
  int ib_tile, ib_c, ib_tap;  // next weight step to issue
  auto setup_b = [&](int tile) {
    const int tN = tile % p.tilesN;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int row = (j * NW + wave) * 8 + prow;
      b_ptr[j] = wg + (size_t)(tN * BN + row) * (size_t)p.Kpad + (size_t)(((lane & 7) ^ ((row >> 1) & 7)) * EPC);
    }
  };
  auto issue_b = [&](int stage) {
    unsigned char* sb = smB + stage * B_BYTES;
    const bool live = ib_tile < p.nblk;
    const int kofs = ib_tap * p.Cin + ib_c * BKE;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const T* src = live ? b_ptr[j] + kofs : zp;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sb + (j * NW + wave) * 1024), 16, 0, 0);
    }
    if (live && ++ib_tap == 9) {
      ib_tap = 0;
      if (++ib_c == NC) {
        ib_c = 0;
        ib_tile += G;
        if (ib_tile < p.nblk) setup_b(ib_tile);
      }
    }
  };

  int ga = 0, gb = 0;  // stages of the chunk / step being COMPUTED
  f32x4 acc[NFR][4];
  const int sl0 = wm * 64 + lr;
  // The halo is resident for all nine taps, so with PREF the pixel fragments of the NEXT step are fetched while the MFMAs
  // of this one run (the LDS read time of a step equals its MFMA time for 64 x 64 wave tiles).
  u32x4 an[2][4];
  auto load_a = [&](int astage, int tapoff, u32x4 (&dst)[2][4]) {
    const int s = sl0 + tapoff;
    const int swzA = s & 6;
    const unsigned char* sa = smA + astage * A_BYTES + s * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ca = ((kk * 4 + lq) ^ swzA) * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[kk][i] = *reinterpret_cast<const u32x4*>(sa + i * 16 * 128 + ca);
    }
  };
  auto compute = [&](int astage, int tapoff, int bstage, int next_astage, int next_tapoff) {
    const unsigned char* sb = smB + bstage * B_BYTES + (wn * 64 + lr) * 128;
    const int swzB = lr >> 1;
    u32x4 a[2][4], b[2][NFR];
    if constexpr (DBG == 2) {  // timing experiment: no LDS reads
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[kk][i] = an[kk][i];
          b[kk][i] = an[kk][i];
        }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < NFR; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[j][i] = Elem<T>::mma(b[kk][j], a[kk][i], acc[j][i]);
      return;
    }
    if constexpr (PREF) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) a[kk][i] = an[kk][i];
    } else {
      load_a(astage, tapoff, a);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int cb = ((kk * 4 + lq) ^ swzB) * 16;
#pragma unroll
      for (int j = 0; j < NFR; ++j) b[kk][j] = *reinterpret_cast<const u32x4*>(sb + j * 16 * 128 + cb);
    }
    if constexpr (PREF) load_a(next_astage, next_tapoff, an);
    if constexpr (DBG == 1) {  // timing experiment: no MFMAs
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(a[kk][i]), "v"(b[kk][i]));
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < NFR; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = Elem<T>::mma(b[kk][j], a[kk][i], acc[j][i]);
  };

  int tile = (int)blockIdx.x;
  if (tile >= p.nblk) return;
  ia_tile = tile, ia_c = 0;
  ib_tile = tile, ib_c = 0, ib_tap = 0;
  setup_a(tile);
  setup_b(tile);
  issue_a(0);
#pragma unroll
  for (int d = 0; d < D; ++d) issue_b(d);
  if constexpr (PREF) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_a(0, 0, an);
  }

  while (true) {
    const int tM = tile / p.tilesN, tN = tile % p.tilesN;
    const int n0 = tN * BN + wn * 64;
#pragma unroll
    for (int j = 0; j < NFR; ++j) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + n0 + j * 16 + lq * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = bb;
    }
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        // Outstanding DMA at the top of step s, oldest first: W(s) [halo if step s-D was a tap 0] W(s+1) ... W(s+D-1)
        // [halo if step s-1 was a tap 0].  W(s) must have landed: everything younger may stay in flight, i.e.
        // 2 (D - 1) weight pieces plus the 7 halo pieces when one of the last D steps was a tap 0 (tap in 1 .. D).
        // (raw s_barrier: a __syncthreads() carries its own vmcnt(0).)
        constexpr int young = PB * (D - 1);
        if (tap >= 1 && tap <= D)
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(young + PA) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(young) : "memory");
        __builtin_amdgcn_s_barrier();
        issue_b(gb + D >= BST ? gb + D - BST : gb + D);
        if (tap == 0) issue_a(ga ^ 1);
        if (tap < 8)
          compute(ga, (tap / 3) * g.Wv + (tap % 3), gb, ga, ((tap + 1) / 3) * g.Wv + ((tap + 1) % 3));
        else
          compute(ga, 2 * g.Wv + 2, gb, ga ^ 1, 0);  // next: first tap of the next chunk (requested at tap 0, published since)
        gb = gb + 1 == BST ? 0 : gb + 1;
      }
      ga ^= 1;
    }
    mfma_epilogue_fence<T>();
    __syncthreads();  // every wave is done reading the last chunk's halo stage (ga ^ 1 now): it becomes the scratch

    // ---- epilogue: 32 pixels per pass through the per-wave scratch ----
    unsigned char* escr = smA + (ga ^ 1) * A_BYTES + wave * (32 * EP_PITCH);
    const int vbase = tM * BM + wm * 64;
    int rpx[4];
    if (rg != nullptr) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rpx[i] = real_px(vbase + i * 16 + lr);
    }
#pragma unroll
    for (int gq = 0; gq < NFR / EG; ++gq) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int jj = 0; jj < EG; ++jj) {
          const int j = gq * EG + jj;
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const int i = half * 2 + ii;
            float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
            if (p.act == DY_ACT_SILU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = silu_f32(v[e]);
            }
            if (rg != nullptr) {
              if (rpx[i] >= 0) {
                typedef __attribute__((ext_vector_type(4))) T t4;
                const t4 rv = *reinterpret_cast<const t4*>(rg + (size_t)rpx[i] * (size_t)p.ldres + (size_t)(n0 + j * 16 + lq * 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f32(rv[e]);
              }
            }
            unsigned char* sp = escr + (ii * 16 + lr) * EP_PITCH + (jj * 16 + lq * 4) * (int)sizeof(T);
            if constexpr (sizeof(T) == 4) {
              *reinterpret_cast<f32x4*>(sp) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
              typedef __attribute__((ext_vector_type(4))) T t4;
              t4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f32(v[e]);
              *reinterpret_cast<u32x2*>(sp) = __builtin_bit_cast(u32x2, o);
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < (32 * CPP + 63) / 64; ++k) {
          const int idx = k * 64 + lane;
          const int px = idx / CPP, cc = idx % CPP;
          const int m = px < 32 ? real_px(vbase + half * 32 + px) : -1;
          if (m >= 0) {
            const u32x4 val = *reinterpret_cast<const u32x4*>(escr + px * EP_PITCH + cc * 16);
            *reinterpret_cast<u32x4*>(yg + (size_t)m * (size_t)p.ldy + (size_t)(n0 + gq * EG * 16 + cc * EPC)) = val;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    tile += G;
    if (tile >= p.nblk) break;
    // the epilogue's global stores (and residual loads) share vmcnt with the DMA: settle them so the counts above hold
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PB * D + PA) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
}

// ---- 16-wave variant: the same tile and staging with wave tiles of 32 pixels x 64 couts, four waves per SIMD (<= 128
// registers each).  The counters of the 8-wave kernel show no saturated unit and 39 % of the wave cycles waiting; more
// waves hide more of that at the price of 0.75 instead of 0.5 LDS reads per MFMA.
// (r04: a BatchNorm-statistics epilogue as in conv_gemm_glds.hip was measured here and dropped: its 32 accumulators do not fit beside
// the tile at the 128-register cap of a 1024-thread workgroup -- 34 spills, 40 -> 64 us on 128 -> 128 @40x40 at B = 64, more than the
// reduction pass over those 26 MB outputs costs.)
template <typename T, int PA, int BST>
__global__ __launch_bounds__(1024) void conv3x3_vgemm16_kernel(const ConvArgs p, const VGeom g) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BKE = 8 * EPC;
  constexpr int BM = 256, BN = 128, NW = 16, MI = 2;

  constexpr int A_BYTES = PA * NW * 1024;
  constexpr int B_BYTES = BN * 128;           // 16384
  constexpr int PB = BN / 8 / NW;             // 2
  constexpr int D = BST - 1;                  // weight prefetch distance in steps
  constexpr int NFR = 4;                      // cout fragments per wave
  constexpr int EG0 = 128 / (16 * (int)sizeof(T));
  constexpr int EG = NFR < EG0 ? NFR : EG0;
  constexpr int CPP = EG * (int)sizeof(T);
  constexpr int EP_PITCH = 128 + 16;
  static_assert(16 * EP_PITCH * NW <= A_BYTES, "epilogue scratch must fit one halo stage");
  static_assert(2 * A_BYTES + BST * B_BYTES <= 160 * 1024, "LDS budget");

  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * A_BYTES + BST * B_BYTES];
  unsigned char* const smA = smem;
  unsigned char* const smB = smem + 2 * A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lq = lane >> 4, lr = lane & 15;
  const int prow = lane >> 3;
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
  const T* zp = reinterpret_cast<const T*>(g_vzero_page) + (lane & 7) * EPC;
  T* __restrict__ yg = reinterpret_cast<T*>(p.y);
  const T* __restrict__ rg = reinterpret_cast<const T*>(p.res);
  const int G = (int)gridDim.x;
  const int NC = g.nchunk;

  // real pixel index of virtual pixel v, or -1 for a pad slot / out of range
  auto real_px = [&](int v) -> int {
    if (v < 0 || v >= g.V) return -1;
    const unsigned t = fastdiv((unsigned)v, g.dWv);
    const int x = v - (int)t * g.Wv;
    const unsigned n = fastdiv(t, g.dHv);
    const int y = (int)t - (int)n * g.Hv;
    if (x >= p.W || y >= p.H) return -1;
    return ((int)n * p.H + y) * p.W + x;
  };

  // ---- issue side: halo (one chunk ahead) and weights (D steps ahead), each with its own tile cursor.  Every issue
  // point ALWAYS emits its full number of DMA instructions (from the zero page once the tiles are exhausted), so that
  // the counted s_waitcnt vmcnt(N) below stay exact ----
  int a_off[PA];          // element offset of this lane's source chunk for each of its pieces, -1 = zero page
  int ia_tile, ia_c;      // next halo chunk to issue
  auto setup_a = [&](int tile) {
    const int tM = tile / p.tilesN;
    const int vstart = tM * BM - g.Wv - 1;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int s = (i * NW + wave) * 8 + prow;
      const int px = s < g.S ? real_px(vstart + s) : -1;
      a_off[i] = px < 0 ? -1 : px * p.ldx + ((lane & 7) ^ (s & 6)) * EPC;  // halo swizzle: see the header (r04)
    }
  };
  auto issue_a = [&](int stage) {
    unsigned char* sa = smA + stage * A_BYTES;
    const bool live = ia_tile < p.nblk;
    const int cofs = ia_c * BKE;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const T* src = (!live || a_off[i] < 0) ? zp : xg + (size_t)(unsigned)(a_off[i] + cofs);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sa + (i * NW + wave) * 1024), 16, 0, 0);
    }
    if (live && ++ia_c == NC) {
      ia_c = 0;
      ia_tile += G;
      if (ia_tile < p.nblk) setup_a(ia_tile);
    }
  };
  const T* b_ptr[PB];
  int ib_tile, ib_c, ib_tap;  // next weight step to issue
  auto setup_b = [&](int tile) {
    const int tN = tile % p.tilesN;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int row = (j * NW + wave) * 8 + prow;
      b_ptr[j] = wg + (size_t)(tN * BN + row) * (size_t)p.Kpad + (size_t)(((lane & 7) ^ ((row >> 1) & 7)) * EPC);
    }
  };
  auto issue_b = [&](int stage) {
    unsigned char* sb = smB + stage * B_BYTES;
    const bool live = ib_tile < p.nblk;
    const int kofs = ib_tap * p.Cin + ib_c * BKE;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const T* src = live ? b_ptr[j] + kofs : zp;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sb + (j * NW + wave) * 1024), 16, 0, 0);
    }
    if (live && ++ib_tap == 9) {
      ib_tap = 0;
      if (++ib_c == NC) {
        ib_c = 0;
        ib_tile += G;
        if (ib_tile < p.nblk) setup_b(ib_tile);
      }
    }
  };

  int ga = 0, gb = 0;  // stages of the chunk / step being COMPUTED
  f32x4 acc[NFR][MI];
  const int sl0 = wm * 32 + lr;
  auto compute = [&](int astage, int tapoff, int bstage, int, int) {
    const int s = sl0 + tapoff;
    const int swzA = s & 6, swzB = lr >> 1;
    const unsigned char* sa = smA + astage * A_BYTES + s * 128;
    const unsigned char* sb = smB + bstage * B_BYTES + (wn * 64 + lr) * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ca = ((kk * 4 + lq) ^ swzA) * 16, cb = ((kk * 4 + lq) ^ swzB) * 16;
      u32x4 a[MI], b[NFR];
#pragma unroll
      for (int j = 0; j < NFR; ++j) b[j] = *reinterpret_cast<const u32x4*>(sb + j * 16 * 128 + cb);
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const u32x4*>(sa + i * 16 * 128 + ca);
#pragma unroll
      for (int j = 0; j < NFR; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = Elem<T>::mma(b[j], a[i], acc[j][i]);
    }
  };

  int tile = (int)blockIdx.x;
  if (tile >= p.nblk) return;
  ia_tile = tile, ia_c = 0;
  ib_tile = tile, ib_c = 0, ib_tap = 0;
  setup_a(tile);
  setup_b(tile);
  issue_a(0);
#pragma unroll
  for (int d = 0; d < D; ++d) issue_b(d);

  while (true) {
    const int tM = tile / p.tilesN, tN = tile % p.tilesN;
    const int n0 = tN * BN + wn * 64;
#pragma unroll
    for (int j = 0; j < NFR; ++j) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + n0 + j * 16 + lq * 4);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[j][i] = bb;
    }
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        // Outstanding DMA at the top of step s, oldest first: W(s) [halo if step s-D was a tap 0] W(s+1) ... W(s+D-1)
        // [halo if step s-1 was a tap 0].  W(s) must have landed: everything younger may stay in flight, i.e.
        // 2 (D - 1) weight pieces plus the 7 halo pieces when one of the last D steps was a tap 0 (tap in 1 .. D).
        // (raw s_barrier: a __syncthreads() carries its own vmcnt(0).)
        constexpr int young = PB * (D - 1);
        if (tap >= 1 && tap <= D)
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(young + PA) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(young) : "memory");
        __builtin_amdgcn_s_barrier();
        issue_b(gb + D >= BST ? gb + D - BST : gb + D);
        if (tap == 0) issue_a(ga ^ 1);
        if (tap < 8)
          compute(ga, (tap / 3) * g.Wv + (tap % 3), gb, ga, ((tap + 1) / 3) * g.Wv + ((tap + 1) % 3));
        else
          compute(ga, 2 * g.Wv + 2, gb, ga ^ 1, 0);  // next: first tap of the next chunk (requested at tap 0, published since)
        gb = gb + 1 == BST ? 0 : gb + 1;
      }
      ga ^= 1;
    }
    mfma_epilogue_fence<T>();
    __syncthreads();  // every wave is done reading the last chunk's halo stage (ga ^ 1 now): it becomes the scratch

    // ---- epilogue: one 16-pixel fragment per pass through the per-wave scratch ----
    unsigned char* escr = smA + (ga ^ 1) * A_BYTES + wave * (16 * EP_PITCH);
    const int vbase = tM * BM + wm * 32;
#pragma unroll
    for (int gq = 0; gq < NFR / EG; ++gq) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int rpx = rg != nullptr ? real_px(vbase + i * 16 + lr) : -1;
#pragma unroll
        for (int jj = 0; jj < EG; ++jj) {
          const int j = gq * EG + jj;
          float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
          if (p.act == DY_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_f32(v[e]);
          }
          if (rpx >= 0) {
            typedef __attribute__((ext_vector_type(4))) T t4;
            const t4 rv = *reinterpret_cast<const t4*>(rg + (size_t)rpx * (size_t)p.ldres + (size_t)(n0 + j * 16 + lq * 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f32(rv[e]);
          }
          unsigned char* sp = escr + lr * EP_PITCH + (jj * 16 + lq * 4) * (int)sizeof(T);
          if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<f32x4*>(sp) = f32x4{v[0], v[1], v[2], v[3]};
          } else {
            typedef __attribute__((ext_vector_type(4))) T t4;
            t4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f32(v[e]);
            *reinterpret_cast<u32x2*>(sp) = __builtin_bit_cast(u32x2, o);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < (16 * CPP + 63) / 64; ++k) {
          const int idx = k * 64 + lane;
          const int px = idx / CPP, cc = idx % CPP;
          const int m = px < 16 ? real_px(vbase + i * 16 + px) : -1;
          if (m >= 0) {
            const u32x4 val = *reinterpret_cast<const u32x4*>(escr + px * EP_PITCH + cc * 16);
            *reinterpret_cast<u32x4*>(yg + (size_t)m * (size_t)p.ldy + (size_t)(n0 + gq * EG * 16 + cc * EPC)) = val;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    tile += G;
    if (tile >= p.nblk) break;
    // the epilogue's global stores (and residual loads) share vmcnt with the DMA: settle them so the counts above hold
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PB * D + PA) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
}

template <typename T>
static int launch_vgemm(const ConvArgs& a, hipStream_t st) {
  ConvArgs p = a;
  VGeom g;
  const int n = a.M / a.HoWo;
  g.Wv = a.W + 1;
  g.Hv = a.H + 1;
  g.V = n * g.Hv * g.Wv;
  g.S = 256 + 2 * g.Wv + 2;
  g.tilesM = (g.V + 255) / 256;
  g.nchunk = a.Cin / (8 * Elem<T>::EPC);
  g.dWv = make_fastdiv((unsigned)g.Wv);
  g.dHv = make_fastdiv((unsigned)g.Hv);
  p.tilesN = a.Cout / 128;
  p.nblk = g.tilesM * p.tilesN;
  int grid = 256;
  if (grid > p.nblk) grid = p.nblk;
  // DYOLO_VGEMM_VAR (experiments): 8 = the 8-wave kernel instead of the 16-wave one, 1 = prefetch the next step's pixel fragments into registers (measured +-2 %),
  // 11 / 12 = timing probes without MFMAs / without LDS reads (wrong results)
  static const int var = dy_ablate("DYOLO_VGEMM_VAR");
  const bool narrow = g.S <= 6 * 64;  // maps up to 62 wide: 48 KiB halo stages leave room for a third weight stage
  const dim3 gr((unsigned)grid), bl(512);
  const char* name = "conv3x3_vgemm_kernel";
  if (!narrow) {
    if (var == 1)
      hipLaunchKernelGGL((conv3x3_vgemm_kernel<T, 7, 3, true>), gr, bl, 0, st, p, g);
    else
      hipLaunchKernelGGL((conv3x3_vgemm_kernel<T, 7, 3, false>), gr, bl, 0, st, p, g);
  } else if (var == 1) {
    hipLaunchKernelGGL((conv3x3_vgemm_kernel<T, 6, 3, true>), gr, bl, 0, st, p, g);
  } else if (var == 11) {
    hipLaunchKernelGGL((conv3x3_vgemm_kernel<T, 6, 3, false, 1>), gr, bl, 0, st, p, g);
  } else if (var == 12) {
    hipLaunchKernelGGL((conv3x3_vgemm_kernel<T, 6, 3, false, 2>), gr, bl, 0, st, p, g);
  } else if (var != 8 && sizeof(T) == 2) {
    // sixteen waves of 32 x 64 hide more of the per-step waits than eight of 64 x 64: 3-5 % faster on every shape in an
    // alternating A/B (tools/ab_conv.sh; 128->128 @40x40 160 -> 151 us, 256->256 @20x20 145 -> 138 us)
    name = "conv3x3_vgemm16_kernel";
    hipLaunchKernelGGL((conv3x3_vgemm16_kernel<T, 3, 3>), gr, dim3(1024), 0, st, p, g);
  } else {
    hipLaunchKernelGGL((conv3x3_vgemm_kernel<T, 6, 3, false>), gr, bl, 0, st, p, g);
  }
  return check_launch(name);
}

// Returns 1 when the shape is not one this kernel is built for, else the launch status.
int conv3x3_vgemm_try(const ConvArgs& a, int dtype, bool out_f32, hipStream_t st) {
  static const int off = dy_ablate("DYOLO_NO_VGEMM");
  if (dtype == DY_FP8) return 1;  // not built for fp8: the generic kernel runs
  const int es = dtype_size_no_fp8(dtype);
  const int bke = 8 * (16 / es);
  if (off || out_f32 || !a.vec_store) return 1;
  if (a.ks != 3 || a.stride != 1 || a.pad != 1 || a.up2x || a.split != a.Cin) return 1;
  if (a.Cin % bke || a.Cin < 128 || a.Cout % 128 || a.Kpad != 9 * a.Cin) return 1;
  if (a.H != a.Ho || a.W != a.Wo || 256 + 2 * (a.W + 1) + 2 > 448) return 1;
  if ((long long)a.M * (long long)a.ldx >= (1ll << 31) || (long long)(a.M / a.HoWo) * (a.H + 1) * (a.W + 1) >= (1ll << 30)) return 1;
  if (a.res && a.ldres % 4) return 1;
  switch (dtype) {
    case DY_BF16: return launch_vgemm<bf16_t>(a, st);
    case DY_F16: return launch_vgemm<f16_t>(a, st);
    default: return launch_vgemm<float>(a, st);
  }
}

}  // namespace DY_NS
