// Convolution weight gradient:  dW[co][r][q][ci] += sum_m dz[m][co] * x[pix(m; r, q)][ci]
// (autograd of F.conv2d w.r.t. the weight for Conv / RepVGGBlock / Detect convolutions in training,
// nn/modules/conv.py:37-55 called under engine/trainer.py:381-389).
//
// A GEMM whose contraction index is the PIXEL: M' = cout, N' = cin (one tap per workgroup), K' = batch*Ho*Wo.  Both
// operands are stored pixel-major (NHWC rows), i.e. "K-major": the MFMA operand of lane (row lr, k-chunk lq) needs 8
// different pixels of ONE channel.  gfx950's ds_read_b64_tr_b16 does that transpose for free: per 16-lane group it
// reads a 4-pixel x 16-channel block of the row-major LDS tile and hands lane i channel i of the 4 pixels.  The 8
// k-values of lane group lq are pixels {4lq..4lq+3} and {16+4lq..16+4lq+3} of the 32-pixel step (any k permutation is
// fine as long as both operands use it): the 32 lanes of a half then touch 8 consecutive tile rows, and with a row pitch
// of (columns*2 + 32) bytes those 8 x 32-byte segments fall into distinct banks.
// fp32 storage: v_mfma_f32_16x16x4_f32 takes one element per lane, so plain ds_read_b32 (no transpose needed).
//
// Workgroup = 4 waves, each wave owns a 64 (cout) x 64 (cin) tile: 2 x 2 waves when cout and cin >= 128, otherwise the
// waves split the pixel range (WK = 4 / (WCO*WCI) independent 32-pixel steps per iteration).  Pixel slabs are spread
// over gridDim.x; partial sums are added to dW with fp32 atomics (dW must be zero before the call).
#include "common_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace dy {

struct WgradArgs {
  const void* x;
  const void* dz;
  float* dw;  // [cout][ks*ks][cin]
  int H, W, Cin, ldx, Ho, Wo, Cout, lddz, ks, stride, pad;
  long long M;
  long long x_bytes, dz_bytes;  // extents of the views: ((pixels - 1) * pitch + channels rounded up to a chunk) * element size
  void* ws;                     // optional workspace of the 3x3 kernel (partial sums per pixel slab) and its size
  size_t ws_bytes;
  int HoWo, tilesCo, tilesCi, rows_per_block;
  FastDiv div_howo, div_wo;  // pixel index -> (image, row, column) without integer division (M < 2^31)
};

// The four 16-channel fragments of one operand for this lane: 8 transposed reads and their wait in ONE asm statement
// (the compiler does not track asm loads, so results must not be touched before the s_waitcnt inside the statement).
template <int PITCH, int ROW2 = 16 * PITCH>
__device__ __forceinline__ void tr_read_frags(const void* base, u32x4 (&out)[4]) {
  u32x2 r0, r1, r2, r3, r4, r5, r6, r7;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%9\n\t"
      "ds_read_b64_tr_b16 %1, %8 offset:%10\n\t"
      "ds_read_b64_tr_b16 %2, %8 offset:%11\n\t"
      "ds_read_b64_tr_b16 %3, %8 offset:%12\n\t"
      "ds_read_b64_tr_b16 %4, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %5, %8 offset:%14\n\t"
      "ds_read_b64_tr_b16 %6, %8 offset:%15\n\t"
      "ds_read_b64_tr_b16 %7, %8 offset:%16\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
      : "v"((unsigned)(uintptr_t)base), "n"(0), "n"(ROW2), "n"(32), "n"(32 + ROW2), "n"(64), "n"(64 + ROW2), "n"(96), "n"(96 + ROW2)
      : "memory");
  out[0] = u32x4{r0[0], r0[1], r1[0], r1[1]};
  out[1] = u32x4{r2[0], r2[1], r3[0], r3[1]};
  out[2] = u32x4{r4[0], r4[1], r5[0], r5[1]};
  out[3] = u32x4{r6[0], r6[1], r7[0], r7[1]};
}

// Two 16-channel fragments (channels 0..31 from `base`): the half-tile form of the above.
template <int PITCH, int ROW2 = 16 * PITCH>
__device__ __forceinline__ void tr_read_frags2(const void* base, u32x4 (&out)[2]) {
  u32x2 r0, r1, r2, r3;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %4 offset:%5\n\t"
      "ds_read_b64_tr_b16 %1, %4 offset:%6\n\t"
      "ds_read_b64_tr_b16 %2, %4 offset:%7\n\t"
      "ds_read_b64_tr_b16 %3, %4 offset:%8\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
      : "v"((unsigned)(uintptr_t)base), "n"(0), "n"(ROW2), "n"(32), "n"(32 + ROW2)
      : "memory");
  out[0] = u32x4{r0[0], r0[1], r1[0], r1[1]};
  out[1] = u32x4{r2[0], r2[1], r3[0], r3[1]};
}

// Pipelined form of the above for the 3x3 kernel: the reads of a group are ISSUED without a wait, the wait is a separate statement
// the destination registers are tied to ("+v": nothing that uses them can be scheduled above it), with the count of younger reads
// that may stay in flight (LDS returns in order; the counter has 4 bits).
template <int OFF, int ROW2>
__device__ __forceinline__ void tr_issue8(unsigned base, u32x2 (&r)[8]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%9\n\t"
      "ds_read_b64_tr_b16 %1, %8 offset:%10\n\t"
      "ds_read_b64_tr_b16 %2, %8 offset:%11\n\t"
      "ds_read_b64_tr_b16 %3, %8 offset:%12\n\t"
      "ds_read_b64_tr_b16 %4, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %5, %8 offset:%14\n\t"
      "ds_read_b64_tr_b16 %6, %8 offset:%15\n\t"
      "ds_read_b64_tr_b16 %7, %8 offset:%16"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
      : "v"(base), "n"(OFF), "n"(OFF + ROW2), "n"(OFF + 32), "n"(OFF + 32 + ROW2), "n"(OFF + 64), "n"(OFF + 64 + ROW2), "n"(OFF + 96), "n"(OFF + 96 + ROW2)
      : "memory");
}
template <int OFF, int ROW2>
__device__ __forceinline__ void tr_issue4(unsigned base, u32x2 (&r)[4]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %4 offset:%5\n\t"
      "ds_read_b64_tr_b16 %1, %4 offset:%6\n\t"
      "ds_read_b64_tr_b16 %2, %4 offset:%7\n\t"
      "ds_read_b64_tr_b16 %3, %4 offset:%8"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
      : "v"(base), "n"(OFF), "n"(OFF + ROW2), "n"(OFF + 32), "n"(OFF + 32 + ROW2)
      : "memory");
}
// three taps of one kernel row, one 16-channel fragment each: reads at OFF + q * STEP + {0, ROW2}
template <int OFF, int ROW2, int STEP>
__device__ __forceinline__ void tr_issue6(unsigned base, u32x2 (&r)[6]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %6 offset:%7\n\t"
      "ds_read_b64_tr_b16 %1, %6 offset:%8\n\t"
      "ds_read_b64_tr_b16 %2, %6 offset:%9\n\t"
      "ds_read_b64_tr_b16 %3, %6 offset:%10\n\t"
      "ds_read_b64_tr_b16 %4, %6 offset:%11\n\t"
      "ds_read_b64_tr_b16 %5, %6 offset:%12"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5])
      : "v"(base), "n"(OFF), "n"(OFF + ROW2), "n"(OFF + STEP), "n"(OFF + STEP + ROW2), "n"(OFF + 2 * STEP), "n"(OFF + 2 * STEP + ROW2)
      : "memory");
}
template <int N>
__device__ __forceinline__ void tr_wait(u32x2 (&r)[6]) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]) : "n"(N));
}
template <int N>
__device__ __forceinline__ void tr_wait(u32x2 (&r)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "n"(N));
}
template <int N>
__device__ __forceinline__ void tr_wait(u32x2 (&r)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N));
}
// reads a group of the 3x3 kernel's schedule issues (group g = (k-step g / 3, tap column g % 3): the tap's four x fragments, and the
// k-step's dz fragments with its first tap), and how many reads of the younger groups are in flight after issuing DEPTH groups ahead
constexpr int wg3_reads(int g, int cw) { return 6 + (g % 3 == 0 ? 2 * cw : 0); }
constexpr int wg3_writes(int g, int wpg, int per) {
  const int lo = g * wpg, hi = (g + 1) * wpg < per ? (g + 1) * wpg : per;
  return g >= 0 && hi > lo ? hi - lo : 0;
}
// LDS operations younger than group g's reads at the point where its MFMAs wait: the reads of groups g + 1, g + 2 and the staging
// stores placed behind the reads of groups g, g + 1, g + 2 (those were issued in the runs of groups g - 2, g - 1, g)
constexpr int wg3_pending(int g, int ng, int cw, int wpg, int per) {
  int n = 0;
  for (int j = g + 1; j <= g + 2 && j < ng; ++j) n += wg3_reads(j, cw);
  for (int j = (g >= 2 ? g - 2 : 0); j <= g; ++j) n += wg3_writes(j, wpg, per);
  return n > 15 ? 15 : n;
}

template <typename T, int WCO, int WCI>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs p) {
  constexpr int E = Elem<T>::EPC;
  constexpr int WK = 4 / (WCO * WCI);
  constexpr int TCO = 64 * WCO, TCI = 64 * WCI;
  constexpr int PA = TCO * (int)sizeof(T) + 32, PB = TCI * (int)sizeof(T) + 32;  // row pitches (bytes)
  constexpr int TILE = 32 * (PA + PB);                                            // one 32-pixel step of both operands
  constexpr int CA = TCO / E, CB = TCI / E;                                       // 16-byte chunks per row
  constexpr int NCH = WK * 32 * (CA + CB);                                        // chunks per iteration
  constexpr int PER = (NCH + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned char smem[WK * TILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int wk = wave / (WCO * WCI), wco = (wave / WCI) % WCO, wci = wave % WCI;
  int t = blockIdx.y;
  const int tci = t % p.tilesCi;
  t /= p.tilesCi;
  const int tco = t % p.tilesCo;
  const int tap = t / p.tilesCo;
  const int r_ = tap / p.ks, q_ = tap - r_ * p.ks;
  const int co0 = tco * TCO, ci0 = tci * TCI;

  const long long m_begin = (long long)blockIdx.x * p.rows_per_block;
  long long m_end = m_begin + p.rows_per_block;
  if (m_end > p.M) m_end = p.M;

  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ dg = reinterpret_cast<const T*>(p.dz);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 stage[PER];
  auto load_step = [&](long long m0) {  // chunk id -> (k-split, operand, row, chunk): global -> registers
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int id = k * 256 + tid;
      u32x4 v = zero_chunk();
      if (id < NCH) {
        const int ws = id / (32 * (CA + CB));
        const int rem = id - ws * 32 * (CA + CB);
        const long long m = m0 + ws * 32;
        if (rem < 32 * CA) {
          const int row = rem / CA, ch = rem - row * CA;
          const long long mm = m + row;
          const int co = co0 + ch * E;
          if (mm < m_end && co < p.Cout) v = *reinterpret_cast<const u32x4*>(dg + mm * p.lddz + co);
        } else {
          const int rem2 = rem - 32 * CA;
          const int row = rem2 / CB, ch = rem2 - row * CB;
          const long long mm = m + row;
          const int ci = ci0 + ch * E;
          if (mm < m_end && ci < p.Cin) {
            const int n = (int)fastdiv((unsigned)mm, p.div_howo);
            const int rr = (int)mm - n * p.HoWo;
            const int ho = (int)fastdiv((unsigned)rr, p.div_wo), wo = rr - ho * p.Wo;
            const int hi = ho * p.stride - p.pad + r_, wi = wo * p.stride - p.pad + q_;
            if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
              v = *reinterpret_cast<const u32x4*>(xg + ((long long)(n * p.H + hi) * p.W + wi) * p.ldx + ci);
          }
        }
      }
      stage[k] = v;
    }
  };
  auto store_step = [&]() {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int id = k * 256 + tid;
      if (id < NCH) {
        const int ws = id / (32 * (CA + CB));
        const int rem = id - ws * 32 * (CA + CB);
        unsigned char* base = smem + ws * TILE;
        if (rem < 32 * CA) {
          const int row = rem / CA, ch = rem - row * CA;
          *reinterpret_cast<u32x4*>(base + row * PA + ch * 16) = stage[k];
        } else {
          const int rem2 = rem - 32 * CA;
          const int row = rem2 / CB, ch = rem2 - row * CB;
          *reinterpret_cast<u32x4*>(base + 32 * PA + row * PB + ch * 16) = stage[k];
        }
      }
    }
  };

  const unsigned char* ta = smem + wk * TILE + wco * 64 * (int)sizeof(T);            // this wave's 64 cout columns
  const unsigned char* tb = smem + wk * TILE + 32 * PA + wci * 64 * (int)sizeof(T);  // this wave's 64 cin columns
  // operand fragments of this lane for k-group kg (16-bit: the whole 32-pixel step; fp32: 16 pixels)
  auto frags = [&](const unsigned char* tile, auto pitch_c, int kg, u32x4 (&out)[4]) {
    constexpr int PITCH = decltype(pitch_c)::value;
    if constexpr (sizeof(T) == 2) {
      // transposed reads: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of the 4 x 16 block;
      // lane group lq takes pixels 4lq..4lq+3 (first read) and 16+4lq.. (second read, +16 rows)
      const int q = lr >> 2, pp = lr & 3;
      tr_read_frags<PITCH>(tile + (lq * 4 + q) * PITCH + pp * 8, out);
      (void)kg;
    } else {
      // fp32: lane quarter lq holds pixels 4lq..4lq+3 of channel lr
      const unsigned char* a0 = tile + (kg * 16 + lq * 4) * PITCH + lr * 4;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) out[f][j] = *reinterpret_cast<const unsigned*>(a0 + f * 64 + j * PITCH);
    }
  };

  const long long step = 32LL * WK;
  if (m_begin < m_end) load_step(m_begin);
  for (long long m0 = m_begin; m0 < m_end; m0 += step) {
    __syncthreads();  // previous iteration's fragment reads are done
    store_step();
    __syncthreads();
    if (m0 + step < m_end) load_step(m0 + step);  // in flight during the MFMAs
    constexpr int KG = sizeof(T) == 2 ? 1 : 2;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      u32x4 a[4], b[4];
      frags(ta, std::integral_constant<int, PA>{}, kg, a);
      frags(tb, std::integral_constant<int, PB>{}, kg, b);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Elem<T>::mma(a[i], b[j], acc[i][j]);
    }
  }
  mfma_epilogue_fence<T>();

  // D[co][ci]: lane holds rows co = 4*lq + reg, column ci = lr of every 16 x 16 fragment
  const int kk = p.ks * p.ks;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ci = ci0 + wci * 64 + j * 16 + lr;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = co0 + wco * 64 + i * 16 + lq * 4 + e;
        if (co < p.Cout && ci < p.Cin) atomicAdd(p.dw + ((size_t)co * kk + tap) * p.Cin + ci, acc[i][j][e]);
      }
    }
}

// ---- 3x3 kernels, 16-bit storage: all nine taps from ONE staged halo --------------------------------------------------
// The kernel above stages x once per tap (nine passes over dz and x: 32 flop per byte moved into LDS).  Here a workgroup
// of three waves walks spatial steps of 2 output rows x 16 output columns; per step it stages the 32 dz pixels and the
// (S+3) x (15*S+3)... halo of x they touch ONCE, wave r owns kernel row r (taps (r,0), (r,1), (r,2): three 64 x 64
// accumulator tiles), the dz fragments are read once per step and every tap's x fragments are transposed reads at
// lane base + compile-time offset of the same halo tile: ~180 flop per staged byte.

struct Wgrad3Args {
  const void* x;
  const void* dz;
  float* dw;
  int H, W, Cin, ldx, Ho, Wo, Cout, lddz;
  int tilesCo, tilesCi, stepsX, stepsY, nSteps, steps_per_block;
  unsigned x_bytes, dz_bytes;  // extents of the x / dz views (buffer descriptors)
  float* part;       // workspace (see wgrad3_reduce_kernel): slab s stores its partial sums at part + s * Cout * 9 * Cin instead of atomics on dw; may be null
  int dbg;  // timing probes (-DDYOLO_ABLATE builds, DYOLO_WGRAD3_DBG): 1 no global loads, 32 no atomics
};

// Eight waves = (cout half h, 16-channel cin fragment c): a wave owns all nine taps x 32 couts x 16 cins = 72 accumulator registers,
// 18 MFMAs per 32-pixel k-step -- two waves on every SIMD (r02's six waves, one per (kernel row, cout half), left two SIMDs with one
// wave and two with two: the barrier waited for the pair).  ONE workgroup per CU (256 overall), so everything that hides latency is
// software:
//  * RS output rows per step (4 where the map's height allows, stride 1): two 32-pixel k-steps per staged halo and barrier;
//  * two register sets of global loads: the set stored during step t holds step t + 1 and is re-issued for step t + 3 as soon as it
//    is stored, so a load has two steps to arrive.  The loads are branch-free buffer loads and are issued on EVERY trip (past the
//    slab's end they read zeros) -- the number in flight is then the same on every path and the compiler's s_waitcnt vmcnt in front
//    of the stores leaves the other set flying.  (r02's loader chose between the dz and the x branch per lane and so read p.lddz /
//    p.ldx through a per-lane SELECTED ADDRESS into the kernel-argument segment: a global_load_dword + s_waitcnt vmcnt(0) in front of
//    every 16-byte load, i.e. a step's loads ran one after the other and nothing stayed in flight: 64 -> 64 @160 349 -> 195 us.)
//  * two LDS stages and ONE barrier per step: the stores of step t + 1's stage are spread over the MFMA groups of step t (their
//    stage was last read in step t - 1, behind the previous barrier) instead of standing between two barriers;
//  * the transposed fragment reads run two groups ahead of the MFMAs that use them (tr_issue* / tr_wait above; the staging stores
//    are part of the same in-order LDS queue and are counted in the waits).
template <typename T, int S, int RS>
__global__ __launch_bounds__(512, 1) void conv_wgrad3x3_kernel(const Wgrad3Args p) {
  constexpr int NT = 512, CW = 2;
  constexpr int E = Elem<T>::EPC;                  // 8
  constexpr int PITCH = 64 * (int)sizeof(T) + 32;  // 160 B rows: 64 channels + pad (conflict-free transposed reads)
  constexpr int HH = (RS - 1) * S + 3, HW = 15 * S + 3;  // x halo of an RS x 16 output step: 5 x 33 (S = 2, RS = 2), 6 x 18 (S = 1, RS = 4)
  constexpr int NPX = HH * HW;
  constexpr int NDZ = RS * 16;
  constexpr int NCHK = (NDZ + NPX) * 8;  // 16-byte chunks per step
  constexpr int PER = (NCHK + NT - 1) / NT;
  // the stage is padded to PER * NT chunks: every thread stores every chunk (no exec-masked store in the loop)
  constexpr int DZ_BYTES = NDZ * PITCH, STAGE = PER * NT / 8 * PITCH;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cf = wv & 3, chf = wv >> 2;  // wave = (cin fragment, cout half)
  const int lr = lane & 15, lq = lane >> 4;
  int t = blockIdx.y;
  const int tci = t % p.tilesCi, tco = t / p.tilesCi;
  const int co0 = tco * 64, ci0 = tci * 64;
  const int s_begin = blockIdx.x * p.steps_per_block;
  int s_end = s_begin + p.steps_per_block;
  if (s_end > p.nSteps) s_end = p.nSteps;

  f32x4 acc[9][CW];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int i = 0; i < CW; ++i) acc[tp][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Loader.  A chunk's role (dz or x) is uniform per (k, wave) -- NDZ * 8 is a multiple of 64 -- so the descriptor and the step's base
  // offset are scalars; the lane's byte offset inside the step is a launch constant; rows / columns outside the image and dead channel
  // chunks take an offset beyond num_records and read zeros.  The x descriptor starts (W + 1) pixels early so no offset is negative.
  constexpr unsigned kOob = 0xfffffff0u;
  const unsigned pre = (unsigned)((p.W + 1) * p.ldx) * 2u;
  const __amdgpu_buffer_rsrc_t xrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) - pre, 0, p.x_bytes + pre, 0x00020000);
  const __amdgpu_buffer_rsrc_t dzrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dz), 0, p.dz_bytes, 0x00020000);
  const int wvu = __builtin_amdgcn_readfirstlane(wv);
  unsigned rel[PER];  // launch constants
  int ryx[PER];       // row | column << 8 inside the step (dz: output pixel, x: halo pixel)
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int id = k * NT + tid;
    if (id < NDZ * 8) {
      const int px = id >> 3, ch = id & 7, row = px >> 4, col = px & 15, co = co0 + ch * E;
      rel[k] = co < p.Cout ? (unsigned)((row * p.Wo + col) * p.lddz + co) * 2u : kOob;
      ryx[k] = row | (col << 8);
    } else {
      const int idx = id - NDZ * 8, px = idx >> 3, ch = idx & 7, hy = px / HW, hx = px - hy * HW, ci = ci0 + ch * E;
      rel[k] = (id < NCHK && ci < p.Cin) ? (unsigned)((hy * p.W + hx) * p.ldx + ci) * 2u : kOob;
      ryx[k] = hy | (hx << 8);
    }
  }
  const unsigned st_lane = (unsigned)((tid >> 3) * PITCH + (tid & 7) * 16);  // chunk id -> stage byte: dz rows first, the halo rows follow in the same pitch

  u32x4 stage[2][PER];
  auto load_step = [&](int step, u32x4 (&stg)[PER]) {
    const int bx = step % p.stepsX;
    int rest = step / p.stepsX;
    const int by = rest % p.stepsY, n = rest / p.stepsY;
    const int y0 = by * RS, x0 = bx * 16;
    // steps past the slab's end read zeros: every offset out of range (a mask the compiler cannot see through -- a visible condition
    // is threaded into two copies of the loads behind a scalar branch)
    unsigned dead = step < s_end ? 0u : 0xffffffffu;
    asm volatile("" : "+v"(dead));
    dead = __builtin_amdgcn_readfirstlane(dead);  // (asm results count as divergent)
    const unsigned dz_base = (unsigned)(((n * p.Ho + y0) * p.Wo + x0) * p.lddz) * 2u & ~dead;
    const unsigned x_base = (unsigned)(((n * p.H + y0 * S) * p.W + x0 * S) * p.ldx) * 2u & ~dead;  // halo origin in the shifted descriptor's terms
#pragma unroll
    for (int k = 0; k < PER; ++k) {
#ifdef DYOLO_ABLATE
      if (p.dbg & 1) {
        stg[k] = u32x4{(unsigned)(k * NT + tid), 0x3c003c00u, (unsigned)step, 0x3c003c00u};
        continue;
      }
#endif
      // role of this (k, wave): known at compile time except for the one k that straddles the dz / x boundary; everything that
      // depends on it is a scalar select (no branch: the compiler then counts the loads in flight exactly)
      const bool role_dz = (k + 1) * NT <= NDZ * 8 ? true : (k * NT >= NDZ * 8 ? false : (k * NT + wvu * 64 < NDZ * 8));
      const int oy = role_dz ? y0 : y0 * S - 1, ox = role_dz ? x0 : x0 * S - 1;
      const unsigned lim_y = role_dz ? (unsigned)p.Ho : (unsigned)p.H, lim_x = role_dz ? (unsigned)p.Wo : (unsigned)p.W;
      const unsigned off = (((unsigned)(oy + (ryx[k] & 255)) < lim_y && (unsigned)(ox + (ryx[k] >> 8)) < lim_x) ? rel[k] : kOob) | dead;
      stg[k] = __builtin_amdgcn_raw_buffer_load_b128(role_dz ? dzrs : xrs, (int)off, (int)(role_dz ? dz_base : x_base), 0);
    }
  };
  auto store_chunk = [&](int buf, int k, const u32x4 (&stg)[PER]) {
    *reinterpret_cast<u32x4*>(smem + buf * STAGE + st_lane + k * (NT / 8 * PITCH)) = stg[k];
    asm volatile("" ::: "memory");  // stays where it is written: the LDS queue is counted by hand in this section
  };

  // lane part of the transposed-read addresses: lane 4q+p of a 16-lane group addresses pixel row q of the 4-pixel block,
  // channels 4p..4p+3; lane group lq takes output columns 4lq..4lq+3 of output row 0 (first read) and row 1 (second read)
  const int q4 = lr >> 2, pp = lr & 3;
  const int dz_lane = (lq * 4 + q4) * PITCH + pp * 8 + chf * 64;  // channels [32 * chf, 32 * chf + 32) of the dz rows
  const int x_lane = DZ_BYTES + ((lq * 4 + q4) * S) * PITCH + pp * 8 + cf * 32;  // channels [16 * cf, 16 * cf + 16) of the halo rows

  // One step: MFMAs on stage `buf` (step st) in NG = 3 * RS / 2 groups (k-step of 32 pixels = a pair of output rows, kernel row: three
  // taps, six MFMAs); `nxt` holds step st + 1: it is stored into the other stage, WPG chunks behind each of the first groups' reads,
  // and re-issued for step st + 3.
  constexpr int NHH = RS / 2, NG = 3 * NHH, WPG = (PER + NG - 1) / NG;
  auto step_body = [&](int st, int buf, u32x4 (&nxt)[PER]) {
    const unsigned bdz = (unsigned)(uintptr_t)(smem + buf * STAGE) + (unsigned)dz_lane;
    const unsigned bx = (unsigned)(uintptr_t)(smem + buf * STAGE) + (unsigned)x_lane;
    u32x2 ra[NHH][2 * CW];
    u32x2 rb[NG][6];
    u32x4 a[NHH][CW];
    auto issue = [&](auto G) {
      constexpr int g = decltype(G)::value, hh = g / 3, r = g % 3;
      if constexpr (r == 0) tr_issue4<hh * 32 * PITCH, 16 * PITCH>(bdz, ra[hh]);
      // tap (r, q): halo pixel of output (row, col) is ((row*S + r) * HW + col*S + q); the second output row is S halo rows below
      tr_issue6<((hh * 2 * S + r) * HW) * PITCH, S * HW * PITCH, PITCH>(bx, rb[g]);
    };
    auto run = [&](auto G) {
      constexpr int g = decltype(G)::value, hh = g / 3, r = g % 3;
      if constexpr (g + 2 < NG) issue(std::integral_constant<int, g + 2>{});
#pragma unroll
      for (int k = g * WPG; k < (g + 1) * WPG && k < PER; ++k) store_chunk(buf ^ 1, k, nxt);
      if constexpr ((g + 1) * WPG >= PER && g * WPG < PER) load_step(st + 3, nxt);  // the set is free: two steps to arrive
      constexpr int N = wg3_pending(g, NG, CW, WPG, PER);
      if constexpr (r == 0) {
        tr_wait<N>(ra[hh]);
#pragma unroll
        for (int i = 0; i < CW; ++i) a[hh][i] = u32x4{ra[hh][2 * i][0], ra[hh][2 * i][1], ra[hh][2 * i + 1][0], ra[hh][2 * i + 1][1]};
      }
      tr_wait<N>(rb[g]);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const u32x4 b = u32x4{rb[g][2 * q][0], rb[g][2 * q][1], rb[g][2 * q + 1][0], rb[g][2 * q + 1][1]};
#pragma unroll
        for (int i = 0; i < CW; ++i) acc[r * 3 + q][i] = Elem<T>::mma(a[hh][i], b, acc[r * 3 + q][i]);
      }
    };
    issue(std::integral_constant<int, 0>{});
    issue(std::integral_constant<int, 1>{});
    run(std::integral_constant<int, 0>{});
    run(std::integral_constant<int, 1>{});
    run(std::integral_constant<int, 2>{});
    if constexpr (NG == 6) {
      run(std::integral_constant<int, 3>{});
      run(std::integral_constant<int, 4>{});
      run(std::integral_constant<int, 5>{});
    }
    __syncthreads();  // stage buf ^ 1 is complete, stage buf is free for the stores of the next step
  };
  static_assert(NG * WPG >= PER, "every chunk of the next stage has a group to be stored behind");

  load_step(s_begin, stage[0]);
  load_step(s_begin + 1, stage[1]);
#pragma unroll
  for (int k = 0; k < PER; ++k) store_chunk(0, k, stage[0]);
  load_step(s_begin + 2, stage[0]);
  __syncthreads();
  for (int st = s_begin; st < s_end; st += 2) {  // two steps per trip: LDS stage and register set by the step's parity
    step_body(st, 0, stage[1]);
    step_body(st + 1, 1, stage[0]);  // an odd slab's last trip runs one step of zeros
  }

  // Epilogue.  Every slab ends with Cout x Cin x 9 partial sums.  fp32 atomics on dw take 30-50 us of a launch (9.4 M lane-atomics
  // with 256 workgroups, whatever the layer: ~1 per clock and L2 channel), more than the MFMAs of the small maps -- so with a
  // workspace the slab STORES its sums in its own copy and wgrad3_reduce_kernel adds the copies up (16 at a time, 1/16 of the atomics).
  float* __restrict__ dst = p.part ? p.part + (size_t)blockIdx.x * ((size_t)p.Cout * 9 * p.Cin) : p.dw;
  const bool plain = p.part != nullptr;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
    for (int i = 0; i < CW; ++i) {
      const int ci = ci0 + cf * 16 + lr;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = co0 + (chf * CW + i) * 16 + lq * 4 + e;
#ifdef DYOLO_ABLATE
        if (p.dbg & 32) continue;  // no atomics at all
#endif
        if (co < p.Cout && ci < p.Cin) {
          float* at = dst + ((size_t)co * 9 + tap) * p.Cin + ci;
          if (plain) *at = acc[tap][i][e];
          else atomicAdd(at, acc[tap][i][e]);
        }
      }
    }
  }
}

// dw[e] += sum over the slabs' copies, WITHOUT atomics: a workgroup owns 128 consecutive elements (32 quads); its 256 threads are
// 32 quads x 8 copy groups, the groups meet in LDS and thread (quad, 0) adds the total to dw -- nobody else touches those elements in
// this launch.  (The first form gave a thread 16 copies and ended with one fp32 atomic per element and group: the atomics, not the
// 37 MB of reads, were its time -- 8 copies per thread +0.5 ms per training step, 32 -0.2, 64 -0.25.)
__global__ __launch_bounds__(256) void wgrad3_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int n, int slabs) {
  __shared__ f32x4 red[8][32];
  const int qd = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int e = (blockIdx.x * 32 + qd) * 4;
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
  const bool vec = (n & 3) == 0;
  if (e < n) {
    if (vec) {
      const float* src = part + (size_t)g * n + e;
#pragma unroll 8
      for (int sl = g; sl < slabs; sl += 8, src += (size_t)8 * n) t += *reinterpret_cast<const f32x4*>(src);
    } else {
      for (int sl = g; sl < slabs; sl += 8)
        for (int k = 0; k < 4 && e + k < n; ++k) t[k] += part[(size_t)sl * n + e + k];
    }
  }
  red[g][qd] = t;
  __syncthreads();
  if (g == 0 && e < n) {
    f32x4 u = red[0][qd];
#pragma unroll
    for (int k = 1; k < 8; ++k) u += red[k][qd];
    if (vec && (reinterpret_cast<uintptr_t>(dw) & 15) == 0) {
      f32x4* d = reinterpret_cast<f32x4*>(dw + e);
      *d = *d + u;
    } else {
      for (int k = 0; k < 4 && e + k < n; ++k) dw[e + k] += u[k];
    }
  }
}

// ---- 1x1 stride-1 layers, 16-bit storage -------------------------------------------------------------------------------
// dW[co][ci] = sum_m dz[m][co] x[m][ci]: both operands are plain row-major (pixels x channels) matrices, so a lane's byte offsets
// are launch constants plus the slab's running row.  Same structure as the 3x3 kernel above: branch-free buffer loads into two
// register sets (two steps in flight, issued on every trip; rows past the slab read zeros), two LDS stages and ONE barrier per
// 32-pixel step, the slab's partial sums STORED in its own workspace copy (wgrad3_reduce_kernel adds the copies; the waves that
// split the pixel range inside a workgroup take one copy each).  conv_wgrad_kernel remains for fp32, strided and k x k layers.
struct Wgrad1Args {
  const void* x;
  const void* dz;
  float* dw;
  float* part;  // workspace or null (atomics on dw)
  int Cin, ldx, Cout, lddz, tilesCo, tilesCi, rows_per_block;
  unsigned x_bytes, dz_bytes;
  long long M;
};

template <typename T, int WCO, int WCI>
__global__ __launch_bounds__(256, 2) void conv_wgrad1x1_kernel(const Wgrad1Args p) {
  constexpr int E = Elem<T>::EPC;
  constexpr int WK = 4 / (WCO * WCI);
  constexpr int TCO = 64 * WCO, TCI = 64 * WCI;
  constexpr int PA = TCO * (int)sizeof(T) + 32, PB = TCI * (int)sizeof(T) + 32;  // row pitches (bytes)
  constexpr int TILE = 32 * (PA + PB);                                            // one 32-pixel step of both operands
  constexpr int CA = TCO / E, CB = TCI / E;                                       // 16-byte chunks per row
  constexpr int NCH = WK * 32 * (CA + CB);                                        // chunks per iteration: 1024 for every shape
  constexpr int PER = NCH / 256;
  static_assert(NCH % 256 == 0, "chunks must divide among the threads");
  constexpr int STEP = 32 * WK;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WK * TILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int wk = wave / (WCO * WCI), wco = (wave / WCI) % WCO, wci = wave % WCI;
  const int tci = blockIdx.y % p.tilesCi, tco = blockIdx.y / p.tilesCi;
  const int co0 = tco * TCO, ci0 = tci * TCI;
  const int m_begin = (int)((long long)blockIdx.x * p.rows_per_block);  // M < 2^31 (host check)
  const int m_end = (int)min((long long)m_begin + p.rows_per_block, p.M);

  constexpr unsigned kOob = 0xfffffff0u;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dzrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dz), 0, p.dz_bytes, 0x00020000);
  // chunk id -> (pixel split, operand, row, chunk): id = k * 256 + tid; the operand of a chunk is uniform per (k, wave) because
  // 32 * CA is a multiple of 64
  unsigned rel[PER], row_of[PER], lds_of[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int id = k * 256 + tid;
    const int ws = id / (32 * (CA + CB)), rem = id - ws * 32 * (CA + CB);
    if (rem < 32 * CA) {
      const int row = rem / CA, ch = rem - row * CA, co = co0 + ch * E;
      row_of[k] = ws * 32 + row;
      rel[k] = co < p.Cout ? (unsigned)(row_of[k] * p.lddz + co) * 2u : kOob;
      lds_of[k] = ws * TILE + row * PA + ch * 16;
    } else {
      const int rem2 = rem - 32 * CA, row = rem2 / CB, ch = rem2 - row * CB, ci = ci0 + ch * E;
      row_of[k] = ws * 32 + row;
      rel[k] = ci < p.Cin ? (unsigned)(row_of[k] * p.ldx + ci) * 2u : kOob;
      lds_of[k] = ws * TILE + 32 * PA + row * PB + ch * 16;
    }
  }
  const int wvu = __builtin_amdgcn_readfirstlane(wave);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 stage[2][PER];
  auto load_step = [&](int m0, u32x4 (&stg)[PER]) {
    // the whole offset goes into the VECTOR operand (range-checked against num_records whatever the hardware does with soffset);
    // rows of the next slab and of steps past the end are switched off per lane
    const unsigned base_dz = (unsigned)m0 * (unsigned)p.lddz * 2u, base_x = (unsigned)m0 * (unsigned)p.ldx * 2u;
    const unsigned left = m0 < m_end ? (unsigned)(m_end - m0) : 0u;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int rem0 = (k * 256 + wvu * 64) % (32 * (CA + CB));
      const bool role_dz = rem0 < 32 * CA;  // scalar
      const unsigned off = (rel[k] != kOob && row_of[k] < left) ? rel[k] + (role_dz ? base_dz : base_x) : kOob;
      stg[k] = __builtin_amdgcn_raw_buffer_load_b128(role_dz ? dzrs : xrs, (int)off, 0, 0);
    }
  };
  auto store_step = [&](int buf, const u32x4 (&stg)[PER]) {
#pragma unroll
    for (int k = 0; k < PER; ++k) *reinterpret_cast<u32x4*>(smem + buf * (WK * TILE) + lds_of[k]) = stg[k];
  };

  const int q4 = lr >> 2, pp = lr & 3;
  const unsigned lane_a = (unsigned)((lq * 4 + q4) * PA + pp * 8 + wco * 64 * (int)sizeof(T));
  const unsigned lane_b = (unsigned)(32 * PA + (lq * 4 + q4) * PB + pp * 8 + wci * 64 * (int)sizeof(T));
  auto step_body = [&](int m0, int buf, u32x4 (&stg)[PER]) {
    store_step(buf, stg);  // stage `buf` was last read two steps ago: the barrier of the previous step covers it
    __syncthreads();
    load_step(m0 + 2 * STEP, stg);
    const unsigned tile = (unsigned)(uintptr_t)(smem + buf * (WK * TILE) + wk * TILE);
    u32x2 ra[8], rb[8];
    tr_issue8<0, 16 * PA>(tile + lane_a, ra);
    tr_issue8<0, 16 * PB>(tile + lane_b, rb);
    tr_wait<8>(ra);
    u32x4 a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = u32x4{ra[2 * i][0], ra[2 * i][1], ra[2 * i + 1][0], ra[2 * i + 1][1]};
    tr_wait<0>(rb);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 b = u32x4{rb[2 * j][0], rb[2 * j][1], rb[2 * j + 1][0], rb[2 * j + 1][1]};
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = Elem<T>::mma(a[i], b, acc[i][j]);
    }
  };
  load_step(m_begin, stage[0]);
  load_step(m_begin + STEP, stage[1]);
  for (int m0 = m_begin; m0 < m_end; m0 += 2 * STEP) {
    step_body(m0, 0, stage[0]);
    step_body(m0 + STEP, 1, stage[1]);  // an odd slab's last trip runs one step of zeros
  }

  // D[co][ci]: lane holds rows co = 4*lq + reg, column ci = lr of every 16 x 16 fragment
  float* __restrict__ dst = p.part ? p.part + ((size_t)blockIdx.x * WK + wk) * ((size_t)p.Cout * p.Cin) : p.dw;
  const bool plain = p.part != nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ci = ci0 + wci * 64 + j * 16 + lr;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = co0 + wco * 64 + i * 16 + lq * 4 + e;
        if (co < p.Cout && ci < p.Cin) {
          float* at = dst + (size_t)co * p.Cin + ci;
          if (plain) *at = acc[i][j][e];
          else atomicAdd(at, acc[i][j][e]);
        }
      }
    }
}

// Launch geometry of the 3x3 kernel: output rows per step, steps, pixel slabs (= gridDim.x = copies in the workspace).
struct Wgrad3Plan {
  int rs, stepsX, stepsY, nSteps, ny, steps_per_block, gx;
};
static Wgrad3Plan wgrad3_plan(const WgradArgs& a, int batch, int stride) {
  Wgrad3Plan g{};
  // four output rows per step where the map is tall enough to fill them (stride 1: the 6 x 18 halo + 64 dz pixels are 27 KB a stage);
  // stride 2 keeps two rows (its halo is 5 x 33 already; four rows -- 9 x 33, 58 KB a stage -- measured 25-60 % slower on every stride-2 layer)
  static const int rs2 = dy_ablate("DYOLO_WGRAD3_RS2");
  g.rs = (stride == 1 && !rs2 && a.Ho % 4 == 0) ? 4 : 2;
  g.stepsX = (a.Wo + 15) / 16, g.stepsY = (a.Ho + g.rs - 1) / g.rs;
  g.nSteps = batch * g.stepsY * g.stepsX;
  g.ny = ((a.Cout + 63) / 64) * ((a.Cin + 63) / 64);
  // Pixel slabs: ONE six-wave workgroup per CU (256 overall) with two steps of loads in flight (PF = 2), every slab at least 8 steps.
  // (r02 ran stride-2 layers with 512 workgroups and one step in flight: that was the loader's serialised loads, see the kernel;
  // B = 64, 512 + PF 1 -> 256 + PF 2: 64 -> 128 s2 @160 222 -> 182 us, 128 -> 256 s2 235 -> 190, 64 -> 64 s2 176 -> 131.)
  const int target = 256;
  int slabs = (target + g.ny - 1) / g.ny;
  const int max_slabs = (g.nSteps + 7) / 8;
  if (slabs > max_slabs) slabs = max_slabs;
  if (slabs < 1) slabs = 1;
  g.steps_per_block = (g.nSteps + slabs - 1) / slabs;
#ifdef DYOLO_ABLATE
  if (const int sl = dy_ablate("DYOLO_WGRAD3_SLABS")) g.steps_per_block = (g.nSteps + sl - 1) / sl;  // probe: another pixel-slab count
#endif
  g.gx = (g.nSteps + g.steps_per_block - 1) / g.steps_per_block;
  return g;
}
static size_t wgrad3_workspace_bytes(const WgradArgs& a, int batch, int stride) {
  const Wgrad3Plan g = wgrad3_plan(a, batch, stride);
  return g.gx > 1 ? (size_t)g.gx * a.Cout * 9 * a.Cin * sizeof(float) : 0;
}

template <typename T, int S, int RS>
static int launch_wgrad3_rs(const WgradArgs& a, const Wgrad3Plan& g, hipStream_t st) {
  Wgrad3Args p{};
  p.x = a.x, p.dz = a.dz, p.dw = a.dw, p.H = a.H, p.W = a.W, p.Cin = a.Cin, p.ldx = a.ldx, p.Ho = a.Ho, p.Wo = a.Wo, p.Cout = a.Cout, p.lddz = a.lddz;
  p.tilesCo = (p.Cout + 63) / 64, p.tilesCi = (p.Cin + 63) / 64;
  p.stepsX = g.stepsX, p.stepsY = g.stepsY, p.nSteps = g.nSteps, p.steps_per_block = g.steps_per_block;
  p.x_bytes = (unsigned)a.x_bytes, p.dz_bytes = (unsigned)a.dz_bytes;
  const size_t n = (size_t)a.Cout * 9 * a.Cin;
  static const int no_part = dy_ablate("DYOLO_WGRAD3_ATOMICS");  // probe: atomics on dw even with a workspace
  p.part = (!no_part && g.gx > 1 && a.ws && a.ws_bytes >= (size_t)g.gx * n * sizeof(float)) ? reinterpret_cast<float*>(a.ws) : nullptr;
  p.dbg = dy_ablate("DYOLO_WGRAD3_DBG");
  const dim3 grid((unsigned)g.gx, (unsigned)g.ny);
  hipLaunchKernelGGL((conv_wgrad3x3_kernel<T, S, RS>), grid, dim3(512), 0, st, p);
  if (const int rc = check_launch("conv_wgrad3x3_kernel")) return rc;
  if (p.part) {
    hipLaunchKernelGGL(wgrad3_reduce_kernel, dim3((unsigned)((n + 127) / 128)), dim3(256), 0, st, p.part, a.dw, (int)n, g.gx);
    return check_launch("wgrad3_reduce_kernel");
  }
  return 0;
}

template <typename T, int S>
static int launch_wgrad3(const WgradArgs& a, int batch, hipStream_t st) {
  const Wgrad3Plan g = wgrad3_plan(a, batch, S);
  if constexpr (S == 1) {
    if (g.rs == 4) return launch_wgrad3_rs<T, 1, 4>(a, g, st);
  }
  return launch_wgrad3_rs<T, S, 2>(a, g, st);
}

template <typename T, int WCO, int WCI>
static int launch_wgrad(const WgradArgs& a, hipStream_t st) {
  WgradArgs p = a;
  p.tilesCo = (p.Cout + 64 * WCO - 1) / (64 * WCO);
  p.tilesCi = (p.Cin + 64 * WCI - 1) / (64 * WCI);
  const int ny = p.ks * p.ks * p.tilesCo * p.tilesCi;
  constexpr int STEP = 32 * (4 / (WCO * WCI));
  // pixel slabs: ~2 workgroups per CU overall, every slab at least 8 steps long.  Every slab ends with its tile's fp32 atomics, and with 1024 workgroups those cost more
  // than the extra latency hiding bought (B = 64, tools/bench_wgrad.py, 1024 -> 512 -> 256 workgroups: 192 -> 128 @80 124 / 104 / 136 us,
  // 384 -> 256 @40 106 / 79 / 106, 64 -> 64 @160 163 / 131 / 134, 768 -> 512 @20 96 / 80 / 100)
  static const int tgt = dy_ablate("DYOLO_WGRAD_TARGET");  // probe: workgroups overall
  long long slabs = ((tgt ? tgt : 512) + ny - 1) / ny;
  const long long max_slabs = (p.M + 8LL * STEP - 1) / (8LL * STEP);
  if (slabs > max_slabs) slabs = max_slabs;
  if (slabs < 1) slabs = 1;
  long long rpb = (p.M + slabs - 1) / slabs;
  rpb = (rpb + STEP - 1) / STEP * STEP;
  p.rows_per_block = (int)rpb;
  const unsigned gx = (unsigned)((p.M + rpb - 1) / rpb);
  hipLaunchKernelGGL((conv_wgrad_kernel<T, WCO, WCI>), dim3(gx, (unsigned)ny), dim3(256), 0, st, p);
  return check_launch("conv_wgrad_kernel");
}

template <typename T>
static int launch_wgrad_dtype(const WgradArgs& a, hipStream_t st) {
  const bool bco = a.Cout > 64, bci = a.Cin > 64;
  if (bco && bci) return launch_wgrad<T, 2, 2>(a, st);
  if (bco) return launch_wgrad<T, 2, 1>(a, st);
  if (bci) return launch_wgrad<T, 1, 2>(a, st);
  return launch_wgrad<T, 1, 1>(a, st);
}

// column sums of a (rows, c) view: the bias gradient of the plain Detect convolutions (head.py:43-57).
// 256 threads = R rows x NCH 16-byte chunks; every thread sums its chunk over its rows of the slab (coalesced 16-byte
// loads), the R partials meet in LDS, one fp32 atomic per channel and workgroup.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* z, float* out, long long rows, int c, int ld, int rows_per_block, int nch, int R) {
  constexpr int E = Elem<T>::EPC;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  float* red = reinterpret_cast<float*>(dyn_smem);  // [R][nch*E]
  const int tid = threadIdx.x, ch = tid % nch, rr = tid / nch;
  const int cw = nch * E;
  if (rr < R) {
    float s[E];
#pragma unroll
    for (int e = 0; e < E; ++e) s[e] = 0.f;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    for (long long r = r0 + rr; r < r1; r += R) {
      float f[E];
      Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(z + r * ld + ch * E), f);
#pragma unroll
      for (int e = 0; e < E; ++e) s[e] += f[e];
    }
#pragma unroll
    for (int e = 0; e < E; ++e) red[rr * cw + ch * E + e] = s[e];
  }
  __syncthreads();
  for (int cc = tid; cc < c; cc += 256) {
    float t = 0.f;
    for (int k = 0; k < R; ++k) t += red[k * cw + cc];
    atomicAdd(out + cc, t);
  }
}

}  // namespace dy

using namespace dy;

// Launch geometry of the 1x1 kernel: waves per tile side, pixel slabs, workspace copies (slabs x pixel splits inside a workgroup).
struct Wgrad1Plan {
  int wco, wci, wk, ny, rows_per_block, gx;
};
static Wgrad1Plan wgrad1_plan(const WgradArgs& a) {
  Wgrad1Plan g{};
  g.wco = a.Cout > 64 ? 2 : 1, g.wci = a.Cin > 64 ? 2 : 1, g.wk = 4 / (g.wco * g.wci);
  g.ny = ((a.Cout + 64 * g.wco - 1) / (64 * g.wco)) * ((a.Cin + 64 * g.wci - 1) / (64 * g.wci));
  const int step = 32 * g.wk;
  static const int tgt = dy_ablate("DYOLO_WGRAD1_TARGET");  // probe: workgroups overall
  long long slabs = ((tgt ? tgt : 512) + g.ny - 1) / g.ny;  // two workgroups per CU, every slab at least 8 steps
  const long long max_slabs = (a.M + 8LL * step - 1) / (8LL * step);
  if (slabs > max_slabs) slabs = max_slabs;
  if (slabs < 1) slabs = 1;
  long long rpb = (a.M + slabs - 1) / slabs;
  rpb = (rpb + step - 1) / step * step;
  g.rows_per_block = (int)rpb;
  g.gx = (int)((a.M + rpb - 1) / rpb);
  return g;
}
static size_t wgrad1_workspace_bytes(const WgradArgs& a) {
  const Wgrad1Plan g = wgrad1_plan(a);
  return g.gx * g.wk > 1 ? (size_t)g.gx * g.wk * a.Cout * a.Cin * sizeof(float) : 0;
}
template <typename T>
static int launch_wgrad1(const WgradArgs& a, hipStream_t st) {
  const Wgrad1Plan g = wgrad1_plan(a);
  Wgrad1Args p{};
  p.x = a.x, p.dz = a.dz, p.dw = a.dw, p.Cin = a.Cin, p.ldx = a.ldx, p.Cout = a.Cout, p.lddz = a.lddz, p.M = a.M;
  p.tilesCo = (a.Cout + 64 * g.wco - 1) / (64 * g.wco), p.tilesCi = (a.Cin + 64 * g.wci - 1) / (64 * g.wci);
  p.rows_per_block = g.rows_per_block, p.x_bytes = (unsigned)a.x_bytes, p.dz_bytes = (unsigned)a.dz_bytes;
  const size_t n = (size_t)a.Cout * a.Cin;
  const int copies = g.gx * g.wk;
  static const int no_part = dy_ablate("DYOLO_WGRAD3_ATOMICS");
  p.part = (!no_part && copies > 1 && a.ws && a.ws_bytes >= (size_t)copies * n * sizeof(float)) ? reinterpret_cast<float*>(a.ws) : nullptr;
  const dim3 grid((unsigned)g.gx, (unsigned)g.ny);
  if (g.wco == 2 && g.wci == 2) hipLaunchKernelGGL((conv_wgrad1x1_kernel<T, 2, 2>), grid, dim3(256), 0, st, p);
  else if (g.wco == 2) hipLaunchKernelGGL((conv_wgrad1x1_kernel<T, 2, 1>), grid, dim3(256), 0, st, p);
  else if (g.wci == 2) hipLaunchKernelGGL((conv_wgrad1x1_kernel<T, 1, 2>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((conv_wgrad1x1_kernel<T, 1, 1>), grid, dim3(256), 0, st, p);
  if (const int rc = check_launch("conv_wgrad1x1_kernel")) return rc;
  if (p.part) {
    hipLaunchKernelGGL(wgrad3_reduce_kernel, dim3((unsigned)((n + 127) / 128)), dim3(256), 0, st, p.part, a.dw, (int)n, copies);
    return check_launch("wgrad3_reduce_kernel");
  }
  return 0;
}

// ---- the image stem: 3x3 stride 2, at most 8 input channels (the 3-channel image padded to one 16-byte chunk), at most 32 couts ----
// On the tiled kernel above this layer uses 3 of the 64 x 64 tile's input channels and half of its couts (6 % of the MFMAs), and seven
// of every eight staged x chunks are padding: 0.7 ms per training step.  Here the x rows sit in LDS at their natural 16-byte pixel
// pitch, and ONE transposed fragment read covers TWO taps: output pixel k of a 32-pixel row segment sees taps (q, q + 1) of kernel row r
// as the 32 contiguous bytes of input pixels (2k + q, 2k + q + 1) -- B[k][n = t * 8 + ci] -- with a row stride of two pixels (32 B: eight
// rows x 32 B are 64 distinct banks).  Per segment and kernel row: fragments (q = 0, 1) and (q = 2, [3 = unused]) x two cout fragments
// = 12 MFMAs on 5 KB of staged data -- the kernel is a stream.  A WAVE is the unit: its own LDS region (no workgroup barriers), the next
// segment's loads in registers while the current one is multiplied, 48 accumulator registers summed over the workgroup's four waves
// at the end and stored as ONE partial per workgroup (wgrad3_reduce_kernel adds them up).
struct WgradStemArgs {
  const void* x;
  const void* dz;
  float* dw;    // [cout][9][cin]
  float* part;  // workspace or null (atomics on dw)
  int N, H, W, Cin, ldx, Ho, Wo, Cout, lddz, segX, nSeg;
  unsigned x_bytes, dz_bytes;
};

template <typename T>
__global__ __launch_bounds__(256, 3) void conv_wgrad_stem_kernel(const WgradStemArgs p) {
  constexpr int PDZ = 96;                    // dz tile: 32 pixels x (32 couts = 64 B + 32 B pad): conflict-free transposed reads
  constexpr int XROW = 68 * 16;              // one halo row: 68 pixels x 16 B (pixels 2 xo0 - 1 .. 2 xo0 + 66)
  constexpr int DZ_BYTES = 32 * PDZ, WAVE_LDS = DZ_BYTES + 3 * XROW;  // 3,072 + 3,264
  constexpr int RED_BYTES = 4 * 12 * 4 * 64 * 4;  // the final sum over the four waves (49 KB) reuses the staging space
  __shared__ __attribute__((aligned(16))) unsigned char smem[RED_BYTES > 4 * WAVE_LDS ? RED_BYTES : 4 * WAVE_LDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  unsigned char* const my = smem + wave * WAVE_LDS;
  constexpr unsigned kOob = 0xfffffff0u;
  const unsigned pre = (unsigned)((p.W + 1) * p.ldx) * 2u;  // the x descriptor starts one row + one pixel early: no negative offsets
  const __amdgpu_buffer_rsrc_t xrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) - pre, 0, p.x_bytes + pre, 0x00020000);
  const __amdgpu_buffer_rsrc_t dzrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dz), 0, p.dz_bytes, 0x00020000);

  // lane constants of the loader.  dz: chunks c = lane, lane + 64 -> pixel c >> 2, 16-byte part c & 3 (8 couts).
  // x: ids lane + 64 i (i < 4) of 3 rows x 68 pixels -> (row, pixel); one chunk (8 channels) per pixel.
  unsigned dz_rel[2], dz_lds[2];
  int dz_px[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane + 64 * i, px = c >> 2, part = c & 3;
    dz_px[i] = px;
    dz_rel[i] = part * 8 < p.Cout ? (unsigned)(px * p.lddz + part * 8) * 2u : kOob;
    dz_lds[i] = (unsigned)(px * PDZ + part * 16);
  }
  unsigned x_rel[4], x_lds[4];
  int x_rc[4];  // row | pixel << 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = lane + 64 * i, row = id / 68, px = id - row * 68;
    x_rc[i] = row | (px << 8);
    x_rel[i] = id < 3 * 68 ? (unsigned)((row * p.W + px) * p.ldx) * 2u : kOob;
    x_lds[i] = (unsigned)(DZ_BYTES + (id < 3 * 68 ? row * XROW + px * 16 : 0));
  }

  f32x4 acc[3][2][2];  // [kernel row][tap pair][cout fragment]
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int f = 0; f < 2; ++f) acc[r][j][f] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int gw = (int)blockIdx.x * 4 + wave, GW = (int)gridDim.x * 4;
  u32x4 rdz[2], rx[4];
  auto load_seg = [&](int seg) {  // seg = (image, output row, 32-pixel segment): scalars of the wave
    unsigned dead = seg < p.nSeg ? 0u : 0xffffffffu;
    asm volatile("" : "+v"(dead));
    dead = __builtin_amdgcn_readfirstlane(dead);
    const int sx = seg % p.segX;
    const int t = seg / p.segX;
    const int yo = t % p.Ho, n = t / p.Ho;
    const int xo0 = sx * 32;
    const unsigned dz_base = (unsigned)(((n * p.Ho + yo) * p.Wo + xo0) * p.lddz) * 2u & ~dead;
    const unsigned x_base = (unsigned)(((n * p.H + 2 * yo) * p.W + 2 * xo0) * p.ldx) * 2u & ~dead;  // halo origin (2 yo - 1, 2 xo0 - 1) in the shifted descriptor's terms
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned off = (xo0 + dz_px[i] < p.Wo ? dz_rel[i] : kOob) | dead;
      rdz[i] = __builtin_amdgcn_raw_buffer_load_b128(dzrs, (int)off, (int)dz_base, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gy = 2 * yo - 1 + (x_rc[i] & 255), gx = 2 * xo0 - 1 + (x_rc[i] >> 8);
      const unsigned off = (((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) ? x_rel[i] : kOob) | dead;
      rx[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)off, (int)x_base, 0);
    }
  };

  const int q4 = lr >> 2, pp = lr & 3;
  const unsigned la = (unsigned)(uintptr_t)my + (unsigned)((lq * 4 + q4) * PDZ + pp * 8);                 // dz: pixel rows of 96 B
  const unsigned lb = (unsigned)(uintptr_t)my + (unsigned)(DZ_BYTES + (lq * 4 + q4) * 32 + pp * 8);       // x: output pixel k at 2 k input pixels = 32 B

  load_seg(gw);
  for (int seg = gw; seg < p.nSeg; seg += GW) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(my + dz_lds[i]) = rdz[i];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (lane + 64 * i < 3 * 68) *reinterpret_cast<u32x4*>(my + x_lds[i]) = rx[i];
    load_seg(seg + GW);  // in flight during this segment's MFMAs (zeros past the end)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own stores are in LDS (one wave: no barrier)
    // one call = four transposed reads at {0, 16 rows, +32 B, +32 B + 16 rows}: for dz the two cout fragments, for an x row the tap pairs
    // (q = 0, 1) and (q = 2, [3]) -- "+32 B" is the next 16 couts there and the next two input pixels here
    u32x2 ra[4], rb[3][4];
    tr_issue4<0, 16 * PDZ>(la, ra);
    tr_issue4<0 * XROW, 16 * 32>(lb, rb[0]);
    tr_issue4<1 * XROW, 16 * 32>(lb, rb[1]);
    tr_issue4<2 * XROW, 16 * 32>(lb, rb[2]);
    tr_wait<12>(ra);
    u32x4 a[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) a[f] = u32x4{ra[2 * f][0], ra[2 * f][1], ra[2 * f + 1][0], ra[2 * f + 1][1]};
    auto taps = [&](auto R) {
      constexpr int r = decltype(R)::value;
      tr_wait<(2 - r) * 4>(rb[r]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u32x4 b = u32x4{rb[r][2 * j][0], rb[r][2 * j][1], rb[r][2 * j + 1][0], rb[r][2 * j + 1][1]};
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[r][j][f] = Elem<T>::mma(a[f], b, acc[r][j][f]);
      }
    };
    taps(std::integral_constant<int, 0>{});
    taps(std::integral_constant<int, 1>{});
    taps(std::integral_constant<int, 2>{});
  }

  // sum over the workgroup's waves (through the staging space), then one partial per workgroup
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);  // [wave][12 tiles][4][64]
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[((wave * 12 + (r * 2 + j) * 2 + f) * 4 + e) * 64 + lane] = acc[r][j][f][e];
  __syncthreads();
  float* __restrict__ dst = p.part ? p.part + (size_t)blockIdx.x * ((size_t)p.Cout * 9 * p.Cin) : p.dw;
  for (int i = tid; i < 12 * 4 * 64; i += 256) {
    const int ln = i & 63, e = (i >> 6) & 3, tile = i >> 8;
    const int f = tile & 1, j = (tile >> 1) & 1, r = tile >> 2;
    const int n16 = ln & 15, t = n16 >> 3, ci = n16 & 7;
    const int co = f * 16 + (ln >> 4) * 4 + e, q = 2 * j + t;
    if (q > 2 || co >= p.Cout || ci >= p.Cin) continue;
    const float v = red[i] + red[i + 12 * 256] + red[i + 2 * 12 * 256] + red[i + 3 * 12 * 256];
    float* at = dst + ((size_t)co * 9 + r * 3 + q) * p.Cin + ci;
    if (p.part) *at = v;
    else atomicAdd(at, v);
  }
}

static int wgrad_stem_blocks(const WgradArgs& a, int batch) {
  const long long segs = (long long)batch * a.Ho * ((a.Wo + 31) / 32);
  const long long want = (segs + 4 * 8 - 1) / (4 * 8);  // at least 8 segments per wave
  return (int)(want < 768 ? (want < 1 ? 1 : want) : 768);  // three 256-thread workgroups per CU
}
static size_t wgrad_stem_workspace_bytes(const WgradArgs& a, int batch) {
  const int g = wgrad_stem_blocks(a, batch);
  return g > 1 ? (size_t)g * a.Cout * 9 * a.Cin * sizeof(float) : 0;
}
template <typename T>
static int launch_wgrad_stem(const WgradArgs& a, int batch, hipStream_t st) {
  WgradStemArgs p{};
  p.x = a.x, p.dz = a.dz, p.dw = a.dw, p.N = batch, p.H = a.H, p.W = a.W, p.Cin = a.Cin, p.ldx = a.ldx, p.Ho = a.Ho, p.Wo = a.Wo, p.Cout = a.Cout, p.lddz = a.lddz;
  p.segX = (a.Wo + 31) / 32, p.nSeg = batch * a.Ho * p.segX;
  p.x_bytes = (unsigned)a.x_bytes, p.dz_bytes = (unsigned)a.dz_bytes;
  const int grid = wgrad_stem_blocks(a, batch);
  const size_t n = (size_t)a.Cout * 9 * a.Cin;
  p.part = (grid > 1 && a.ws && a.ws_bytes >= (size_t)grid * n * sizeof(float)) ? reinterpret_cast<float*>(a.ws) : nullptr;
  hipLaunchKernelGGL((conv_wgrad_stem_kernel<T>), dim3((unsigned)grid), dim3(256), 0, st, p);
  if (const int rc = check_launch("conv_wgrad_stem_kernel")) return rc;
  if (p.part) {
    hipLaunchKernelGGL(wgrad3_reduce_kernel, dim3((unsigned)((n + 127) / 128)), dim3(256), 0, st, p.part, a.dw, (int)n, grid);
    return check_launch("wgrad3_reduce_kernel");
  }
  return 0;
}

// validation + geometry shared by the entry points; `uses3` = the 3x3 kernel takes this call
static int wgrad_setup(const dy_conv_desc* d, const void* dz, int32_t ld_dz, float* dw, bool need_ptrs, WgradArgs& a, bool& uses3, bool& uses1, bool& uses_stem) {
  DY_REQUIRE(d && (!need_ptrs || (d->x && dz && dw)), DY_ERR_INVALID_ARG, "dy_conv2d_wgrad_nhwc: null pointer");
  const int es = dtype_size_no_fp8(d->dtype);
  DY_REQUIRE(es != 0 && d->batch > 0 && d->h > 0 && d->w_in > 0 && d->cin > 0 && d->cout > 0 && d->ksize >= 1 && d->stride >= 1 && d->pad >= 0,
             DY_ERR_INVALID_ARG, "dy_conv2d_wgrad_nhwc: bad dims");
  DY_REQUIRE(d->groups <= 1 && !d->up2x && !d->x2, DY_ERR_UNSUPPORTED, "dy_conv2d_wgrad_nhwc: dense single-source convolutions only");
  const int epc = 16 / es;
  // channels are read in whole 16-byte chunks: the pitches must cover cin / cout ROUNDED UP to a chunk (what lies in the
  // padding only reaches gradient rows / columns that are never written)
  DY_REQUIRE((!need_ptrs || (aligned16(d->x) && aligned16(dz))) && (d->ld_x * es) % 16 == 0 && (ld_dz * es) % 16 == 0 &&
                 d->ld_x >= (d->cin + epc - 1) / epc * epc && ld_dz >= (d->cout + epc - 1) / epc * epc,
             DY_ERR_INVALID_ARG, "dy_conv2d_wgrad_nhwc: x / dz views must be 16-byte aligned with pitches covering the channels rounded up to %d", epc);
  const int ho = (d->h + 2 * d->pad - d->ksize) / d->stride + 1, wo = (d->w_in + 2 * d->pad - d->ksize) / d->stride + 1;
  DY_REQUIRE(ho == d->ho && wo == d->wo, DY_ERR_INVALID_ARG, "dy_conv2d_wgrad_nhwc: ho/wo (%d,%d) != expected (%d,%d)", d->ho, d->wo, ho, wo);
  a.x = d->x, a.dz = dz, a.dw = dw;
  a.H = d->h, a.W = d->w_in, a.Cin = d->cin, a.ldx = d->ld_x, a.Ho = ho, a.Wo = wo, a.Cout = d->cout, a.lddz = ld_dz;
  a.ks = d->ksize, a.stride = d->stride, a.pad = d->pad;
  a.M = (long long)d->batch * ho * wo;
  a.x_bytes = (((long long)d->batch * d->h * d->w_in - 1) * d->ld_x + (d->cin + epc - 1) / epc * epc) * es;
  a.dz_bytes = ((a.M - 1) * ld_dz + (d->cout + epc - 1) / epc * epc) * es;
  DY_REQUIRE(a.M < (1ll << 31), DY_ERR_UNSUPPORTED, "dy_conv2d_wgrad_nhwc: pixel count exceeds int32");
  a.HoWo = ho * wo;
  a.div_howo = make_fastdiv((unsigned)a.HoWo);
  a.div_wo = make_fastdiv((unsigned)wo);
  static const int no3 = dy_ablate("DYOLO_NO_WGRAD3");
  // all nine taps from one staged halo (32-bit byte offsets in its buffer loads: views of 2 GiB and more take the per-tap kernel)
  uses3 = !no3 && d->ksize == 3 && d->pad == 1 && (d->stride == 1 || d->stride == 2) && es == 2 && a.x_bytes < (1ll << 31) && a.dz_bytes < (1ll << 31);
  static const int no_stem = dy_ablate("DYOLO_NO_WGRAD_STEM");
  uses_stem = uses3 && !no_stem && d->stride == 2 && d->cin <= 8 && d->cout <= 32;  // the image stem (conv_wgrad_stem_kernel)
  if (uses_stem) uses3 = false;
  static const int no1 = dy_ablate("DYOLO_NO_WGRAD1");
  uses1 = !no1 && d->ksize == 1 && d->pad == 0 && d->stride == 1 && es == 2 && a.x_bytes < (1ll << 31) && a.dz_bytes < (1ll << 31);
  return 0;
}

extern "C" int64_t dy_conv2d_wgrad_workspace_bytes(const dy_conv_desc* d, int32_t ld_dz) {
  WgradArgs a{};
  bool uses3 = false, uses1 = false, uses_stem = false;
  if (const int rc = wgrad_setup(d, nullptr, ld_dz, nullptr, false, a, uses3, uses1, uses_stem)) return rc;
  if (uses_stem) return (int64_t)wgrad_stem_workspace_bytes(a, d->batch);
  return uses3 ? (int64_t)wgrad3_workspace_bytes(a, d->batch, d->stride) : uses1 ? (int64_t)wgrad1_workspace_bytes(a) : 0;
}

extern "C" int32_t dy_conv2d_wgrad_nhwc_ws(const dy_conv_desc* d, const void* dz, int32_t ld_dz, float* dw, void* workspace, int64_t workspace_bytes,
                                           dy_stream_t stream) {
  WgradArgs a{};
  bool uses3 = false, uses1 = false, uses_stem = false;
  if (const int rc = wgrad_setup(d, dz, ld_dz, dw, true, a, uses3, uses1, uses_stem)) return rc;
  DY_REQUIRE(!workspace || (aligned16(workspace) && workspace_bytes >= 0), DY_ERR_INVALID_ARG, "dy_conv2d_wgrad_nhwc_ws: workspace must be 16-byte aligned");
  a.ws = workspace, a.ws_bytes = workspace ? (size_t)workspace_bytes : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (uses3) {
    if (d->dtype == DY_BF16) return d->stride == 1 ? launch_wgrad3<bf16_t, 1>(a, d->batch, st) : launch_wgrad3<bf16_t, 2>(a, d->batch, st);
    return d->stride == 1 ? launch_wgrad3<f16_t, 1>(a, d->batch, st) : launch_wgrad3<f16_t, 2>(a, d->batch, st);
  }
  if (uses_stem) return d->dtype == DY_BF16 ? launch_wgrad_stem<bf16_t>(a, d->batch, st) : launch_wgrad_stem<f16_t>(a, d->batch, st);
  if (uses1) return d->dtype == DY_BF16 ? launch_wgrad1<bf16_t>(a, st) : launch_wgrad1<f16_t>(a, st);
  switch (d->dtype) {
    case DY_BF16: return launch_wgrad_dtype<bf16_t>(a, st);
    case DY_F16: return launch_wgrad_dtype<f16_t>(a, st);
    default: return launch_wgrad_dtype<float>(a, st);
  }
}

extern "C" int32_t dy_conv2d_wgrad_nhwc(const dy_conv_desc* d, const void* dz, int32_t ld_dz, float* dw, dy_stream_t stream) {
  return dy_conv2d_wgrad_nhwc_ws(d, dz, ld_dz, dw, nullptr, 0, stream);
}

extern "C" int32_t dy_colsum(const void* z, float* out, int64_t rows, int32_t c, int32_t ld, int32_t dtype, dy_stream_t stream) {
  const int es = dtype_size_no_fp8(dtype);
  DY_REQUIRE(z && out && rows > 0 && c > 0 && es, DY_ERR_INVALID_ARG, "dy_colsum: bad arguments");
  const int epc = 16 / es, nch = (c + epc - 1) / epc;
  // channels are read in whole 16-byte chunks: the pitch must cover c rounded up (the padding only reaches sums that are dropped)
  DY_REQUIRE(aligned16(z) && (ld * es) % 16 == 0 && ld >= nch * epc && nch <= 256, DY_ERR_INVALID_ARG,
             "dy_colsum: view must be 16-byte aligned with a pitch covering c rounded up to %d (c <= %d)", epc, 256 * epc);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int R = 256 / nch;
  long long blocks = (rows + 8LL * R - 1) / (8LL * R);
  if (blocks > 256) blocks = 256;  // every workgroup ends with one atomic per channel on the same addresses: they serialise (1024: 20 us a call)
  const int rpb = (int)((rows + blocks - 1) / blocks);
  const unsigned gx = (unsigned)((rows + rpb - 1) / rpb);
  const size_t smem = (size_t)R * nch * epc * 4;
  switch (dtype) {
    case DY_BF16: hipLaunchKernelGGL((colsum_kernel<bf16_t>), dim3(gx), dim3(256), smem, st, reinterpret_cast<const bf16_t*>(z), out, (long long)rows, c, ld, rpb, nch, R); break;
    case DY_F16: hipLaunchKernelGGL((colsum_kernel<f16_t>), dim3(gx), dim3(256), smem, st, reinterpret_cast<const f16_t*>(z), out, (long long)rows, c, ld, rpb, nch, R); break;
    default: hipLaunchKernelGGL((colsum_kernel<float>), dim3(gx), dim3(256), smem, st, reinterpret_cast<const float*>(z), out, (long long)rows, c, ld, rpb, nch, R); break;
  }
  return check_launch("dy_colsum");
}
