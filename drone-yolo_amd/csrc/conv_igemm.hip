// NHWC implicit-GEMM convolution on CDNA4 MFMA (gfx950).
//
// GEMM view:  C[M][N] = A[M][K] * W[N][K]^T,  M = batch*Ho*Wo output pixels, N = Cout,
// K = ksize*ksize*Cin flattened in (r, q, c) order (c fastest), so that a 16-byte
// chunk of K is 16 bytes of contiguous NHWC channels of one input pixel.
//
// One 256-thread workgroup (4 waves) owns a BM x BN output tile; waves are stacked
// along M.  Every K-step (8 chunks = 128 bytes of K per row) the A tile is gathered
// from global memory (zero outside the image / beyond K), the W tile is a plain 2-D
// tile of the packed weights, both are register-staged into a double-buffered LDS
// image (global loads for step s+1 are issued before the MFMAs of step s and written
// after them: one barrier per step).  The LDS image is fragment-ordered:
//   slot(row, chunk) = ((kgroup*ROWFRAGS + row/16)*4 + chunk%4)*16 + ((row%16) ^ chunk)
// so that the ds_read_b128 of an MFMA operand (lane = quarter*16 + row%16) is
// lane-linear up to an XOR inside the 16-slot bank row (conflict free), and the
// ds_write_b128 of 8 consecutive lanes (same row, chunks 0..7) hits 8 distinct 16-byte
// slots mod 8 (conflict free on the 32-bank write path).
// Epilogue: accumulators -> fp32 LDS tile -> bias + SiLU (+ residual) in fp32 -> 16-byte
// channel-contiguous stores.
//
// Reference semantics: nn/modules/conv.py:37-55 (Conv), block.py:337-350 (Bottleneck
// residual), block.py:1480-1490 (RepVGGBlock, folded), head.py:43-57 (Detect convs).
#include "common_hip.h"
#include "conv_args.h"
#include <type_traits>

namespace DY_NS {


template <typename T, int BM, int BN, bool OUTF32>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
  constexpr int EPC = Elem<T>::EPC;      // elements per 16-byte chunk
  constexpr int BKE = 8 * EPC;           // K elements per step
  constexpr int NA = BM / 32;            // A chunks per thread per step
  constexpr int NB = (BN >= 32) ? BN / 32 : 1;
  constexpr int MFR = BM / 16;           // row fragments in the A image
  constexpr int NFR = BN / 16;           // row fragments in the W image
  constexpr int MF = BM / 64;            // m fragments per wave
  constexpr int NF = BN / 16;            // n fragments per wave
  constexpr int A_BYTES = BM * 128;
  constexpr int B_BYTES = BN * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int CLD = BN + 4;            // fp32 C tile pitch
  constexpr int C_BYTES = BM * CLD * 4;
  constexpr int SMEM = (2 * STAGE > C_BYTES) ? 2 * STAGE : C_BYTES;
  typedef typename std::conditional<OUTF32, float, T>::type OutT;
  constexpr int VEC = 16 / sizeof(OutT);  // output channels per 16-byte store

  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const unsigned L = xcd_remap(blockIdx.x, (unsigned)p.nblk);
  const int tileN = (int)(L % (unsigned)p.tilesN);
  const int tileM = (int)(L / (unsigned)p.tilesN);

  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ x2g = reinterpret_cast<const T*>(p.x2);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);

  // ---- per-thread gather bookkeeping -------------------------------------------
  const int chunk = tid & 7;
  const int rowbase = tid >> 3;  // 0..31
  int a_pix[NA], a_pix2[NA], a_hi0[NA], a_wi0[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int m = tileM * BM + rowbase + 32 * i;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int n = mm / p.HoWo;
    const int rem = mm - n * p.HoWo;
    const int ho = rem / p.Wo;
    const int wo = rem - ho * p.Wo;
    a_pix[i] = n * p.HB * p.WB;
    a_pix2[i] = n * p.H * p.W;
    a_hi0[i] = ok ? ho * p.stride - p.pad : -(1 << 28);
    a_wi0[i] = wo * p.stride - p.pad;
  }
  // position of this thread's chunk inside the flattened K axis: (r, q, c)
  int kc, kr, kq;
  {
    const int kk0 = chunk * EPC;
    const int tap = kk0 / p.Cin;
    kc = kk0 - tap * p.Cin;
    kr = tap / p.ks;
    kq = tap - kr * p.ks;
  }
  const size_t w_row0 = (size_t)(tileN * BN + rowbase) * (size_t)p.Kpad + (size_t)(chunk * EPC);

  u32x4 ra[NA], rb[NB];

  auto load_global = [&](int step) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int hi = a_hi0[i] + kr;
      const int wi = a_wi0[i] + kq;
      const bool ok = (kr < p.ks) && ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W) && !(p.up2x == 2 && ((hi | wi) & 1));
      u32x4 v = zero_chunk();
      if (ok) {
        if (kc < p.split) {
          const int hb = p.up2x ? (hi >> 1) : hi;
          const int wb = p.up2x ? (wi >> 1) : wi;
          const size_t off = (size_t)(a_pix[i] + hb * p.WB + wb) * (size_t)p.ldx + (size_t)kc;
          v = *reinterpret_cast<const u32x4*>(xg + off);
        } else {
          const size_t off = (size_t)(a_pix2[i] + hi * p.W + wi) * (size_t)p.ldx2 + (size_t)(kc - p.split);
          v = *reinterpret_cast<const u32x4*>(x2g + off);
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (BN >= 32 || rowbase < BN) {
        const size_t off = w_row0 + (size_t)(32 * j) * (size_t)p.Kpad + (size_t)step * BKE;
        rb[j] = *reinterpret_cast<const u32x4*>(wg + off);
      }
    }
    // advance (r, q, c) by one K-step
    kc += BKE;
    while (kc >= p.Cin) {
      kc -= p.Cin;
      if (++kq == p.ks) {
        kq = 0;
        ++kr;
      }
    }
  };

  const int kgrp = chunk >> 2, cq = chunk & 3;
  auto store_lds = [&](int stage) {
    unsigned char* sa = smem + stage * STAGE;
    unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = rowbase + 32 * i;
      const int slot = ((kgrp * MFR + (row >> 4)) * 4 + cq) * 16 + ((row & 15) ^ chunk);
      *reinterpret_cast<u32x4*>(sa + slot * 16) = ra[i];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (BN >= 32 || rowbase < BN) {
        const int row = rowbase + 32 * j;
        const int slot = ((kgrp * NFR + (row >> 4)) * 4 + cq) * 16 + ((row & 15) ^ chunk);
        *reinterpret_cast<u32x4*>(sb + slot * 16) = rb[j];
      }
    }
  };

  f32x4 acc[MF][NF];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lq = lane >> 4, lr = lane & 15;
  auto compute = [&](int stage) {
    const unsigned char* sa = smem + stage * STAGE;
    const unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int sw = lr ^ (4 * s + lq);
      u32x4 a[MF], b[NF];
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const int slot = ((s * MFR + wave * MF + i) * 4 + lq) * 16 + sw;
        a[i] = *reinterpret_cast<const u32x4*>(sa + slot * 16);
      }
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int slot = ((s * NFR + j) * 4 + lq) * 16 + sw;
        b[j] = *reinterpret_cast<const u32x4*>(sb + slot * 16);
      }
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = Elem<T>::mma(a[i], b[j], acc[i][j]);
    }
  };

  // ---- main loop: one barrier per K-step ------------------------------------------
  const int nsteps = p.Kpad / BKE;
  load_global(0);
  store_lds(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const bool more = (s + 1) < nsteps;
    if (more) load_global(s + 1);
    compute(s & 1);
    if (more) store_lds((s + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue ----------------------------------------------------------------------
  mfma_epilogue_fence<T>();
  float* cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (wave * MF + i) * 16 + lq * 4 + r;
        const int col = j * 16 + lr;
        cs[row * CLD + col] = acc[i][j][r];
      }
  __syncthreads();

  OutT* __restrict__ yg = reinterpret_cast<OutT*>(p.y);
  const T* __restrict__ rg = reinterpret_cast<const T*>(p.res);
  constexpr int CV = BN / VEC;  // vectors per tile row
  for (int it = tid; it < BM * CV; it += 256) {
    const int row = it / CV;
    const int col = (it - row * CV) * VEC;
    const int m = tileM * BM + row;
    const int gcol = tileN * BN + col;
    if (m >= p.M || gcol >= p.Cout) continue;
    float v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e += 4) {
      f32x4 t = *reinterpret_cast<const f32x4*>(cs + row * CLD + col + e);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + gcol + e);
      if constexpr (std::is_same<T, fp8_t>::value) {  // integer-like products of quanta -> real units
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.wscale + gcol + e);
        t = f32x4{t[0] * sc[0], t[1] * sc[1], t[2] * sc[2], t[3] * sc[3]};
      }
      v[e + 0] = t[0] + bb[0];
      v[e + 1] = t[1] + bb[1];
      v[e + 2] = t[2] + bb[2];
      v[e + 3] = t[3] + bb[3];
    }
    if (p.act == DY_ACT_SILU) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = silu_f32(v[e]);
    }
    const int nvalid = (p.Cout - gcol) < VEC ? (p.Cout - gcol) : VEC;
    if constexpr (!OUTF32) if (rg != nullptr) {
      const T* rp = rg + (size_t)m * (size_t)p.ldres + (size_t)gcol;
      if (nvalid == VEC) {
        float rf[VEC];
        Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(rp), reinterpret_cast<float(&)[Elem<T>::EPC]>(rf));
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] += rf[e] * (std::is_same<T, fp8_t>::value ? p.act_scale : 1.f);
      } else {
        for (int e = 0; e < nvalid; ++e) v[e] += Elem<T>::to_f32(rp[e]) * (std::is_same<T, fp8_t>::value ? p.act_scale : 1.f);
      }
    }
    if constexpr (std::is_same<T, fp8_t>::value && !OUTF32) {
      const float inv = 1.f / p.act_scale;
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] *= inv;
    }
    OutT* yp = yg + (size_t)m * (size_t)p.ldy + (size_t)gcol;
    if (p.vec_store && nvalid == VEC) {
      if constexpr (OUTF32 || sizeof(T) == 4) {
        *reinterpret_cast<f32x4*>(yp) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
        *reinterpret_cast<u32x4*>(yp) = Chunk<T>::pack(reinterpret_cast<const float(&)[Elem<T>::EPC]>(v));
      }
    } else {
      for (int e = 0; e < nvalid; ++e) {
        if constexpr (OUTF32)
          yp[e] = v[e];
        else
          yp[e] = Elem<T>::from_f32(v[e]);
      }
    }
  }
}

// ---- grouped / depthwise direct kernel (DWConv, conv.py:102-107) ----------------------------
// One thread per (pixel, output channel); bandwidth bound and tiny in this model
// (yolov8-p2-repvgg-sf.yaml:32,38,44), so no MFMA.  w: [cout][ks*ks*(cin/groups)] in (r,q,c) order.
template <typename T>
__global__ __launch_bounds__(256) void conv_grouped_kernel(const ConvArgs p, int groups) {
  const long long total = (long long)p.M * p.Cout;
  const int cpg_in = p.Cin / groups, cpg_out = p.Cout / groups;
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
  const T* __restrict__ rg = reinterpret_cast<const T*>(p.res);
  T* __restrict__ yg = reinterpret_cast<T*>(p.y);
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int co = (int)(idx % p.Cout);
    const int m = (int)(idx / p.Cout);
    const int n = m / p.HoWo;
    const int rem = m - n * p.HoWo;
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    const int g = co / cpg_out;
    float acc = p.bias[co];
    const T* wr = wg + (size_t)co * (size_t)(p.ks * p.ks * cpg_in);
    for (int r = 0; r < p.ks; ++r) {
      const int hi = ho * p.stride - p.pad + r;
      if ((unsigned)hi >= (unsigned)p.H) continue;
      for (int q = 0; q < p.ks; ++q) {
        const int wi = wo * p.stride - p.pad + q;
        if ((unsigned)wi >= (unsigned)p.W) continue;
        const T* xp = xg + (size_t)((n * p.H + hi) * p.W + wi) * (size_t)p.ldx + (size_t)(g * cpg_in);
        const T* wp = wr + (r * p.ks + q) * cpg_in;
        for (int c = 0; c < cpg_in; ++c) acc += Elem<T>::to_f32(xp[c]) * Elem<T>::to_f32(wp[c]);
      }
    }
    if (p.act == DY_ACT_SILU) acc = silu_f32(acc);
    if (rg) acc += Elem<T>::to_f32(rg[(size_t)m * (size_t)p.ldres + co]);
    yg[(size_t)m * (size_t)p.ldy + co] = Elem<T>::from_f32(acc);
  }
}

// The DWConv layers of the -sf YAML are Conv(c1, c2, 3, 2, g = gcd(c1, c2)) with c2 = c1 / 2 (yolov8-p2-repvgg-sf.yaml:32,38,44): every output
// channel reads CPG = 2 input channels (1 or 4 at other widths).  The kernel above walks them one output element per thread with 2-byte
// loads (r05 bench: three launches, 2.1 ms of a 15 ms pass at 1.6 TFLOP/s).  Here a thread owns ONE 16-byte chunk of output channels of a
// pixel: per tap it loads the CPG input chunks those channels read (contiguous), the weights of the workgroup's channel range sit in LDS as
// [tap][cin-in-group][cout] so that a lane's operands are unit-stride; fp32 accumulate, bias + SiLU, one 16-byte store: HBM-bound work at HBM speed.
template <typename T, int CPG>
__global__ __launch_bounds__(256) void conv_smallgroup_kernel(const ConvArgs p) {
  constexpr int EPC = Elem<T>::EPC;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  float* wl = reinterpret_cast<float*>(dyn_smem);  // [ks*ks][CPG][Cout]
  const int taps = p.ks * p.ks;
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
  for (int i = threadIdx.x; i < taps * CPG * p.Cout; i += 256) {
    const int co = i % p.Cout, t2 = i / p.Cout;
    const int ci = t2 % CPG, tap = t2 / CPG;
    wl[i] = Elem<T>::to_f32(wg[(size_t)co * (size_t)(taps * CPG) + tap * CPG + ci]);
  }
  __syncthreads();
  const int cch = p.Cout / EPC;  // output chunks per pixel
  const long long total = (long long)p.M * cch;
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  T* __restrict__ yg = reinterpret_cast<T*>(p.y);
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int cc = (int)(idx % cch);
    const int m = (int)(idx / cch);
    const int n = m / p.HoWo;
    const int rem = m - n * p.HoWo;
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    const int co0 = cc * EPC;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = p.bias[co0 + e];
    for (int r = 0; r < p.ks; ++r) {
      const int hi = ho * p.stride - p.pad + r;
      if ((unsigned)hi >= (unsigned)p.H) continue;
      for (int q = 0; q < p.ks; ++q) {
        const int wi = wo * p.stride - p.pad + q;
        if ((unsigned)wi >= (unsigned)p.W) continue;
        const T* xp = xg + (size_t)((n * p.H + hi) * p.W + wi) * (size_t)p.ldx + (size_t)co0 * CPG;  // input channels co * CPG .. of this chunk: CPG chunks
        const float* wt = wl + (size_t)((r * p.ks + q) * CPG) * p.Cout + co0;
        float xin[CPG * EPC];
#pragma unroll
        for (int k = 0; k < CPG; ++k) {
          float f[EPC];
          Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(xp + k * EPC), f);
#pragma unroll
          for (int e = 0; e < EPC; ++e) xin[k * EPC + e] = f[e];
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e)
#pragma unroll
          for (int ci = 0; ci < CPG; ++ci) acc[e] += xin[e * CPG + ci] * wt[(size_t)ci * p.Cout + e];
      }
    }
    if (p.act == DY_ACT_SILU) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] = silu_f32(acc[e]);
    }
    *reinterpret_cast<u32x4*>(yg + (size_t)m * (size_t)p.ldy + co0) = Chunk<T>::pack(acc);
  }
}

// The same for DY_F16X2 (split float16; the -sf YAML at the default precision of YOLO.predict, r05): a thread owns one group of 8 output
// channels of a pixel — a (hi, lo) chunk pair, 32 bytes — joins the CPG input pairs those channels read to fp32 (exact), multiplies by fp32
// weights (kept whole in LDS: the layer's rows are handed over in fp32, [cout][taps][CPG]) and splits the result again.  Pitches in elements
// of 4 bytes, as everywhere for this type.
#ifndef DYOLO_L2E_BUILD
template <int CPG>
__global__ __launch_bounds__(256) void conv_smallgroup_split_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  float* wl = reinterpret_cast<float*>(dyn_smem);  // [ks*ks][CPG][Cout]
  const int taps = p.ks * p.ks;
  const float* __restrict__ wg = reinterpret_cast<const float*>(p.w);
  for (int i = threadIdx.x; i < taps * CPG * p.Cout; i += 256) {
    const int co = i % p.Cout, t2 = i / p.Cout;
    const int ci = t2 % CPG, tap = t2 / CPG;
    wl[i] = wg[(size_t)co * (size_t)(taps * CPG) + tap * CPG + ci];
  }
  __syncthreads();
  const int cg = p.Cout / 8;  // output groups per pixel
  const long long total = (long long)p.M * cg;
  const unsigned* __restrict__ xg = reinterpret_cast<const unsigned*>(p.x);
  unsigned* __restrict__ yg = reinterpret_cast<unsigned*>(p.y);
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int cc = (int)(idx % cg);
    const int m = (int)(idx / cg);
    const int n = m / p.HoWo;
    const int rem = m - n * p.HoWo;
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    const int co0 = cc * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = p.bias[co0 + e];
    for (int r = 0; r < p.ks; ++r) {
      const int hi = ho * p.stride - p.pad + r;
      if ((unsigned)hi >= (unsigned)p.H) continue;
      for (int q = 0; q < p.ks; ++q) {
        const int wi = wo * p.stride - p.pad + q;
        if ((unsigned)wi >= (unsigned)p.W) continue;
        const unsigned* xp = xg + (size_t)((n * p.H + hi) * p.W + wi) * (size_t)p.ldx + (size_t)co0 * CPG;  // input channels co * CPG ..: CPG groups of 8
        const float* wt = wl + (size_t)((r * p.ks + q) * CPG) * p.Cout + co0;
        float xin[CPG * 8];
#pragma unroll
        for (int k = 0; k < CPG; ++k) {
          float f[8];
          join8(*reinterpret_cast<const u32x4*>(xp + k * 8), *reinterpret_cast<const u32x4*>(xp + k * 8 + 4), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) xin[k * 8 + e] = f[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int ci = 0; ci < CPG; ++ci) acc[e] += xin[e * CPG + ci] * wt[(size_t)ci * p.Cout + e];
      }
    }
    if (p.act == DY_ACT_SILU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = silu_f32(acc[e]);
    }
    u32x4 oh, ol;
    split8(acc, oh, ol);
    unsigned* yp = yg + (size_t)m * (size_t)p.ldy + co0;
    *reinterpret_cast<u32x4*>(yp) = oh;
    *reinterpret_cast<u32x4*>(yp + 4) = ol;
  }
}

static int launch_smallgroup_split(const ConvArgs& a, int cpg, hipStream_t st) {
  const long long total = (long long)a.M * (a.Cout / 8);
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  const size_t smem = (size_t)a.ks * a.ks * cpg * a.Cout * 4;
  switch (cpg) {
    case 1: hipLaunchKernelGGL((conv_smallgroup_split_kernel<1>), dim3(blocks), dim3(256), smem, st, a); break;
    case 2: hipLaunchKernelGGL((conv_smallgroup_split_kernel<2>), dim3(blocks), dim3(256), smem, st, a); break;
    default: hipLaunchKernelGGL((conv_smallgroup_split_kernel<4>), dim3(blocks), dim3(256), smem, st, a); break;
  }
  return check_launch("conv_smallgroup_split_kernel");
}
#endif

template <typename T>
static int launch_smallgroup(const ConvArgs& a, int cpg, hipStream_t st) {
  const long long total = (long long)a.M * (a.Cout / Elem<T>::EPC);
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  const size_t smem = (size_t)a.ks * a.ks * cpg * a.Cout * 4;
  switch (cpg) {
    case 1: hipLaunchKernelGGL((conv_smallgroup_kernel<T, 1>), dim3(blocks), dim3(256), smem, st, a); break;
    case 2: hipLaunchKernelGGL((conv_smallgroup_kernel<T, 2>), dim3(blocks), dim3(256), smem, st, a); break;
    default: hipLaunchKernelGGL((conv_smallgroup_kernel<T, 4>), dim3(blocks), dim3(256), smem, st, a); break;
  }
  return check_launch("conv_smallgroup_kernel");
}

template <typename T, int BM, int BN, bool OUTF32>
static int launch_tile(const ConvArgs& a, hipStream_t st) {
  ConvArgs p = a;
  const int tilesM = (p.M + BM - 1) / BM;
  p.tilesN = (p.Cout + BN - 1) / BN;
  p.nblk = tilesM * p.tilesN;
  hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, OUTF32>), dim3((unsigned)p.nblk), dim3(256), 0, st, p);
  return check_launch("conv_igemm_kernel");
}

template <typename T, bool OUTF32>
static int launch_dtype(const ConvArgs& a, hipStream_t st) {
  // Tile choice: BN follows Cout; BM drops to 64 when a 128-row tiling would leave
  // fewer than ~2 workgroups per CU (256 CUs), the deep/small-M layers (P4/P5).
  const int bn = a.Cout > 32 ? 64 : (a.Cout > 16 ? 32 : 16);
  const long long blocks128 = (long long)((a.M + 127) / 128) * ((a.Cout + bn - 1) / bn);
  const bool small = blocks128 < 512;
  if (bn == 64) return small ? launch_tile<T, 64, 64, OUTF32>(a, st) : launch_tile<T, 128, 64, OUTF32>(a, st);
  if (bn == 32) return small ? launch_tile<T, 64, 32, OUTF32>(a, st) : launch_tile<T, 128, 32, OUTF32>(a, st);
  return small ? launch_tile<T, 64, 16, OUTF32>(a, st) : launch_tile<T, 128, 16, OUTF32>(a, st);
}

int conv3x3_halo_dispatch(const dy_conv_desc* d, hipStream_t st);    // conv3x3_halo.hip
int conv_gemm_fk_try(const dy_conv_desc* d, hipStream_t st);         // conv_gemm_fk.hip: flat-K kernel (any channel count / fp8 on the block-scaled MFMA)
int conv_gemm_fk_split(const dy_conv_desc* d, hipStream_t st);       // conv_gemm_fk.hip: DY_F16X2 (split float16), the only kernel of that type
int conv1x1_stream_dispatch(const dy_conv_desc* d, hipStream_t st);  // conv1x1_stream.hip

}  // namespace DY_NS

using namespace DY_NS;

#ifndef DYOLO_L2E_BUILD
extern "C" int32_t dy_conv_k_pad(int32_t cin, int32_t ksize, int32_t dtype) {
  const int es = dy_dtype_size(dtype);
  if (es == 0 || cin <= 0 || ksize <= 0) return -1;
  const int bke = 8 * (16 / es);
  const int k = ksize * ksize * cin;
  return (k + bke - 1) / bke * bke;
}

extern "C" int32_t dy_conv_cout_pad(int32_t cout) { return cout <= 0 ? -1 : (cout + 63) / 64 * 64; }

namespace dy_l2e {
int32_t conv2d_entry(const dy_conv_desc* d, dy_stream_t stream);
}
namespace dy {
int32_t conv2d_entry(const dy_conv_desc* d, dy_stream_t stream);
}
// DY_ACT_SILU_L2E runs the second compilation of the kernels (namespace dy_l2e: silu_f32 is the scaled-domain formula there)
extern "C" int32_t dy_conv2d_nhwc(const dy_conv_desc* d, dy_stream_t stream) {
  dy::note_stats(0);
  if (d != nullptr && d->act == DY_ACT_SILU_L2E) {
    dy_conv_desc c = *d;
    c.act = DY_ACT_SILU;
    return dy_l2e::conv2d_entry(&c, stream);
  }
  return dy::conv2d_entry(d, stream);
}
#endif

#ifndef DYOLO_L2E_BUILD
namespace dy {
int conv3x3_hsplit_try(const dy_conv_desc* d, hipStream_t st);  // conv3x3_hsplit.hip: DY_F16X2, 3x3 stride 1, cin 32 / 64
}
#endif
namespace DY_NS {
int32_t conv2d_entry(const dy_conv_desc* d, dy_stream_t stream) {
  DY_REQUIRE(d != nullptr, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: null descriptor");
  DY_REQUIRE(d->x && d->w && d->bias && d->y, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: null x/w/bias/y");
  const int es = dy_dtype_size(d->dtype);
  DY_REQUIRE(es != 0, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: bad dtype %d", d->dtype);
  const int epc = 16 / es;
  DY_REQUIRE(d->batch > 0 && d->h > 0 && d->w_in > 0 && d->cin > 0 && d->cout > 0, DY_ERR_INVALID_ARG,
             "dy_conv2d_nhwc: non-positive dims");
  DY_REQUIRE(d->ksize >= 1 && d->stride >= 1 && d->pad >= 0, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: bad ksize/stride/pad");
  const int ho = (d->h + 2 * d->pad - d->ksize) / d->stride + 1;
  const int wo = (d->w_in + 2 * d->pad - d->ksize) / d->stride + 1;
  DY_REQUIRE(ho == d->ho && wo == d->wo, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: ho/wo (%d,%d) != expected (%d,%d)", d->ho,
             d->wo, ho, wo);
  DY_REQUIRE((long long)d->batch * ho * wo < (1ll << 31) && (long long)d->batch * d->h * d->w_in < (1ll << 31),
             DY_ERR_UNSUPPORTED, "dy_conv2d_nhwc: pixel count exceeds int32");
  DY_REQUIRE(d->ld_y >= d->cout && d->ld_x > 0, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: bad pitches");
  DY_REQUIRE(!(d->residual && d->out_f32), DY_ERR_UNSUPPORTED, "dy_conv2d_nhwc: residual with out_f32");
  DY_REQUIRE(!d->residual || d->ld_res >= d->cout, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: bad ld_res");

  ConvArgs a{};
  a.x = d->x;
  a.x2 = d->x;
  a.w = d->w;
  a.bias = d->bias;
  a.res = d->residual;
  a.y = d->y;
  a.H = d->h;
  a.W = d->w_in;
  a.Cin = d->cin;
  a.ldx = d->ld_x;
  a.ldx2 = d->ld_x;
  a.split = d->cin;
  a.HB = d->h;
  a.WB = d->w_in;
  a.Ho = ho;
  a.Wo = wo;
  a.Cout = d->cout;
  a.ldy = d->ld_y;
  a.ldres = d->ld_res;
  a.ks = d->ksize;
  a.stride = d->stride;
  a.pad = d->pad;
  a.M = d->batch * ho * wo;
  a.HoWo = ho * wo;
  a.act = d->act;
  a.up2x = d->up2x;  // 0 plain, 1 nearest-upsampled source, 2 zero-dilated source (stride-2 transposed conv)
  a.stats = (d->out_f32 || d->y_dtype1 || d->bnb_z) ? nullptr : d->bn_stats;  // (kernels without a statistics epilogue ignore it: dy_conv_stats_written() stays 0)
  if (d->bnb_z) {
    DY_REQUIRE(d->bn_stats && d->bnb_mean && d->bnb_rstd && d->bnb_gamma && d->bnb_beta, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: bnb_z needs bn_stats and bnb_mean / rstd / gamma / beta");
    DY_REQUIRE(aligned16(d->bnb_z) && d->bnb_ld_z >= d->cout && (d->bnb_ld_z * es) % 16 == 0 && !d->residual && !d->out_f32, DY_ERR_INVALID_ARG,
               "dy_conv2d_nhwc: bnb_z view misaligned, pitch below cout, or combined with a residual / out_f32");
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);

#ifndef DYOLO_L2E_BUILD
  if (d->dtype == DY_F16X2 && d->groups > 1) {
    // DWConv of the -sf YAML (conv.py:102-107) in split float16: one output channel per group, 1 / 2 / 4 input channels each; w = fp32 rows
    const int cpg = d->groups > 0 && d->cin % d->groups == 0 ? d->cin / d->groups : 0;
    DY_REQUIRE(d->cout == d->groups && (cpg == 1 || cpg == 2 || cpg == 4) && d->cout % 8 == 0 && !d->residual && !d->out_f32 && !d->up2x && !d->x2 && !d->y_dtype1 && !d->bn_stats &&
                   d->w_layout == DY_WLAYOUT_ROWS, DY_ERR_UNSUPPORTED,
               "dy_conv2d_nhwc: a grouped DY_F16X2 convolution is built for cout == groups, 1 / 2 / 4 input channels per group, cout %% 8 == 0, plain call");
    DY_REQUIRE((size_t)d->ksize * d->ksize * cpg * d->cout * 4 <= 48 * 1024, DY_ERR_UNSUPPORTED, "dy_conv2d_nhwc: grouped DY_F16X2 weights exceed 48 KB of LDS");
    DY_REQUIRE(aligned16(d->x) && aligned16(d->y) && d->ld_x % 4 == 0 && d->ld_y % 4 == 0 && d->ld_x >= d->cin && (reinterpret_cast<uintptr_t>(d->w) & 3) == 0, DY_ERR_INVALID_ARG,
               "dy_conv2d_nhwc: grouped DY_F16X2 views must be 16-byte aligned with pitches in whole chunks");
    return launch_smallgroup_split(a, cpg, st);
  }
#endif
  if (d->dtype == DY_F16X2) {
    DY_REQUIRE(d->k_pad == dy_conv_k_pad(d->cin, d->ksize, d->dtype) && d->cout_pad == dy_conv_cout_pad(d->cout), DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: k_pad / cout_pad");
    DY_REQUIRE(aligned16(d->w) && aligned16(d->bias), DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: w/bias not 16-byte aligned");
    DY_REQUIRE(!d->up2x || (d->h % 2 == 0 && d->w_in % 2 == 0), DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: up2x needs even h, w");
#ifndef DYOLO_L2E_BUILD
    {
      const int rh = ::dy::conv3x3_hsplit_try(d, st);  // narrow 3x3 layers: halo staged once, weights in registers (conv3x3_hsplit.hip)
      if (rh <= 0) return rh;
    }
#endif
    return conv_gemm_fk_split(d, st);
  }
  if (d->dtype == DY_FP8) {
    DY_REQUIRE(d->w_layout == DY_WLAYOUT_ROWS && d->groups <= 1, DY_ERR_UNSUPPORTED, "dy_conv2d_nhwc: DY_FP8 is built for dense convolutions in DY_WLAYOUT_ROWS");
    DY_REQUIRE(d->w_scale && d->act_scale > 0.f && aligned16(d->w_scale), DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: DY_FP8 needs w_scale (fp32[cout_pad], 16-byte aligned) and act_scale > 0");
    a.wscale = d->w_scale;
    a.act_scale = d->act_scale;
  } else {
    a.wscale = nullptr;
    a.act_scale = 1.f;
  }
  DY_REQUIRE(!d->y_dtype1 || d->y_dtype1 - 1 == d->dtype || (d->w_layout == DY_WLAYOUT_ROWS && !d->out_f32), DY_ERR_UNSUPPORTED,
             "dy_conv2d_nhwc: an output type other than the input's (y_dtype1) needs DY_WLAYOUT_ROWS and no out_f32");
  if (d->w_layout == DY_WLAYOUT_HALO3X3) return conv3x3_halo_dispatch(d, st);
  if (d->w_layout == DY_WLAYOUT_FRAG1X1) return conv1x1_stream_dispatch(d, st);
  DY_REQUIRE(d->w_layout == DY_WLAYOUT_ROWS, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: unknown w_layout %d", d->w_layout);

  if (d->groups > 1) {
    DY_REQUIRE(d->cin % d->groups == 0 && d->cout % d->groups == 0, DY_ERR_INVALID_ARG,
               "dy_conv2d_nhwc: groups %d does not divide cin/cout", d->groups);
    DY_REQUIRE(!d->out_f32 && !d->up2x && !d->x2, DY_ERR_UNSUPPORTED, "dy_conv2d_nhwc: grouped conv option unsupported");
    DY_REQUIRE(d->ld_x >= d->cin, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: bad ld_x");
    {
      // one output channel per group, 1 / 2 / 4 input channels each, whole 16-byte chunks everywhere, weights of the layer within 48 KB of LDS
      const int cpg = d->cin / d->groups;
      const int epc_g = 16 / es;
      if (d->cout == d->groups && (cpg == 1 || cpg == 2 || cpg == 4) && !d->residual && d->cout % epc_g == 0 && (d->ld_x * es) % 16 == 0 && (d->ld_y * es) % 16 == 0 &&
          aligned16(d->x) && aligned16(d->y) && (size_t)d->ksize * d->ksize * cpg * d->cout * 4 <= 48 * 1024 && (d->dtype == DY_BF16 || d->dtype == DY_F16 || d->dtype == DY_F32)) {
        if (d->dtype == DY_BF16) return launch_smallgroup<bf16_t>(a, cpg, st);
        if (d->dtype == DY_F16) return launch_smallgroup<f16_t>(a, cpg, st);
        return launch_smallgroup<float>(a, cpg, st);
      }
    }
    const long long total = (long long)a.M * a.Cout;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (d->dtype == DY_BF16)
      hipLaunchKernelGGL((conv_grouped_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, a, d->groups);
    else if (d->dtype == DY_F16)
      hipLaunchKernelGGL((conv_grouped_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, a, d->groups);
    else
      hipLaunchKernelGGL((conv_grouped_kernel<float>), dim3(blocks), dim3(256), 0, st, a, d->groups);
    return check_launch("conv_grouped_kernel");
  }

  // dense MFMA path
  int c1 = d->cin;  // channels served by x
  if (d->x2) {
    DY_REQUIRE(d->cin_split > 0 && d->cin_split < d->cin && d->cin_split % epc == 0, DY_ERR_INVALID_ARG,
               "dy_conv2d_nhwc: bad cin_split %d", d->cin_split);
    DY_REQUIRE(aligned16(d->x2) && (d->ld_x2 * es) % 16 == 0 && d->ld_x2 >= d->cin - d->cin_split, DY_ERR_INVALID_ARG,
               "dy_conv2d_nhwc: x2 view misaligned or pitch too small");
    a.x2 = d->x2;
    a.ldx2 = d->ld_x2;
    a.split = d->cin_split;
    c1 = d->cin_split;
  }
  DY_REQUIRE(d->ld_x >= c1, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: ld_x %d < channels %d", d->ld_x, c1);
  DY_REQUIRE(d->cin % epc == 0, DY_ERR_UNSUPPORTED, "dy_conv2d_nhwc: cin %d not a multiple of %d", d->cin, epc);
  DY_REQUIRE(aligned16(d->x) && (d->ld_x * es) % 16 == 0, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: x view not 16-byte aligned");
  DY_REQUIRE(aligned16(d->w) && aligned16(d->bias), DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: w/bias not 16-byte aligned");
  DY_REQUIRE(d->k_pad == dy_conv_k_pad(d->cin, d->ksize, d->dtype), DY_ERR_INVALID_ARG,
             "dy_conv2d_nhwc: k_pad %d != dy_conv_k_pad()", d->k_pad);
  DY_REQUIRE(d->cout_pad == dy_conv_cout_pad(d->cout), DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: cout_pad %d != dy_conv_cout_pad()",
             d->cout_pad);
  if (d->up2x) {
    DY_REQUIRE(d->h % 2 == 0 && d->w_in % 2 == 0 && (d->up2x == 1 || d->up2x == 2), DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: up2x must be 1 or 2 and needs even h,w");
    DY_REQUIRE(d->up2x == 1 || !d->x2, DY_ERR_UNSUPPORTED, "dy_conv2d_nhwc: a zero-dilated source cannot be combined with x2");
    a.HB = d->h / 2;
    a.WB = d->w_in / 2;
  }
  if (d->residual)
    DY_REQUIRE(aligned16(d->residual) && (d->ld_res * es) % 16 == 0, DY_ERR_INVALID_ARG,
               "dy_conv2d_nhwc: residual view not 16-byte aligned");
  a.Kpad = d->k_pad;
  const int oes = d->out_f32 ? 4 : es;
  a.vec_store = (aligned16(d->y) && (d->ld_y * oes) % 16 == 0) ? 1 : 0;

  // cout tiles of 80 / 160 (the x scale: 160, 320, 800 ... outputs) beat the 64-cout persistent tiles of the tap-aligned kernel:
  // a layer whose Cout is no multiple of 128 but fills 160-wide tiles exactly goes to the flat-K kernel first
  const bool fk_first = d->cout % 128 != 0 && d->cout % 80 == 0 && (d->dtype == DY_BF16 || d->dtype == DY_F16) && !d->out_f32 && d->up2x != 2;
  if (fk_first) {
    const int rf = conv_gemm_fk_try(d, st);
    if (rf <= 0) return rf;
  }
  if (!d->y_dtype1 || d->y_dtype1 - 1 == d->dtype) {
    const int rv = conv3x3_vgemm_try(a, d->dtype, d->out_f32 != 0, st);  // deep 3x3 stride-1 layers: halo staged once per chunk
    if (rv <= 0) return rv;
    const int rc = conv_gemm_glds_try(a, d->dtype, d->out_f32 != 0, st);  // big-tile LDS-DMA kernel where it is built
    if (rc <= 0) return rc;
  }
  {
    const int rf = conv_gemm_fk_try(d, st);  // channel counts that are not whole tap-aligned K-steps / cout tiles; fp8 on the block-scaled MFMA
    if (rf <= 0) return rf;
  }
  DY_REQUIRE(!d->y_dtype1 || d->y_dtype1 - 1 == d->dtype, DY_ERR_UNSUPPORTED,
             "dy_conv2d_nhwc: y_dtype1 %d with dtype %d is built in the flat-K kernel only (dense 1x1 / 3x3, DY_WLAYOUT_ROWS, 16-byte aligned views)", d->y_dtype1, d->dtype);
  switch (d->dtype) {
    case DY_BF16:
      return d->out_f32 ? launch_dtype<bf16_t, true>(a, st) : launch_dtype<bf16_t, false>(a, st);
    case DY_F16:
      return d->out_f32 ? launch_dtype<f16_t, true>(a, st) : launch_dtype<f16_t, false>(a, st);
    case DY_FP8:
      return d->out_f32 ? launch_dtype<fp8_t, true>(a, st) : launch_dtype<fp8_t, false>(a, st);
    default:
      return launch_dtype<float, false>(a, st);
  }
}
}  // namespace DY_NS
