// Train-mode BatchNorm2d (+ SiLU) forward and backward on NHWC activations.
// Reference: Conv.forward = act(bn(conv(x))) in training mode (nn/modules/conv.py:37-55), BatchNorm2d with
// eps 1e-3 / momentum 0.03 (utils/torch_utils.py:423-433); RepVGGBlock.forward sums two BN branches before the
// activation (nn/modules/block.py:1480-1490) — served by `addend` (forward) and by running the backward twice on the
// same incoming gradient.
//
// Layout: z is (rows, c) with pitch ld — channels fastest, so a per-channel reduction walks rows.  A 256-thread
// workgroup is arranged as R rows x NCH 16-byte channel chunks; each thread keeps fp32 partial sums of its chunk over
// its rows of the slab, the R partials of a chunk are combined through LDS and added to per-channel DOUBLE
// accumulators with atomics (sum and sum of squares in double: var = E[z^2] - mean^2 stays accurate at 3.3 M rows).
// All kernels are bandwidth bound: forward reads z twice (stats, apply) and writes y once; backward reads dy and z
// twice and writes dz once.
#include "common_hip.h"

namespace dy {

constexpr int kBnMaxSlabs = kStatSlots;
constexpr int kBnMaxC = 2048;  // widest layer one workgroup row covers: 256 threads x one 16-byte chunk of 16-bit elements  // workgroups of a reduction pass (dy_bn_workspace_bytes reserves one partial each)

struct BnArgs {
  const void* z;
  void* y;
  const void* addend;
  const void* dy;
  void* dz;
  long long rows;
  int c, ld_z, ld_y, ld_add, ld_dy, ld_dz, act;
  const float* gamma;
  const float* beta;
  float* mean;
  float* rstd;
  float* running_mean;
  float* running_var;
  float eps, momentum;
  float* dgamma;
  float* dbeta;
  double* acc;  // [2][c]
  int partial_slabs;  // forward: > 0 = that many slab partials are already in the workspace (a convolution epilogue's)
  int nch, R, rows_per_block;
  double invn;  // 1 / rows (host side: a double division per thread is ~40 quarter-rate instructions)
};

__device__ __forceinline__ float silu_grad(float u) {  // d/du u*sigmoid(u)
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
  return s * (1.0f + u * (1.0f - s));
}

// MODE 0: sum z, sum z^2.   MODE 1: sum du, sum du*xhat with du = dy * act'(u), u = gamma*xhat + beta.
template <typename T, int MODE, int UNR>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const BnArgs p) {
  constexpr int E = Elem<T>::EPC;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  float* red = reinterpret_cast<float*>(dyn_smem);  // [2][R][c]
  const int tid = threadIdx.x;
  const int ch = tid % p.nch, rr = tid / p.nch;
  if (blockIdx.x == 0)  // the totals bn_sum_partials_kernel (the next launch) adds into: zeroed here instead of by a memset node per call
    for (int i = tid; i < 2 * p.c; i += 256) p.acc[i] = 0.0;
  float s0[E], s1[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s0[e] = 0.f, s1[e] = 0.f;
  if (rr < p.R) {
    float mu[E], rs[E], ga[E], be[E];
    if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        mu[e] = p.mean[ch * E + e], rs[e] = p.rstd[ch * E + e], ga[e] = p.gamma[ch * E + e], be[e] = p.beta[ch * E + e];
      }
    }
    const long long r0 = (long long)blockIdx.x * p.rows_per_block;
    long long r1 = r0 + p.rows_per_block;
    if (r1 > p.rows) r1 = p.rows;
    const T* zb = reinterpret_cast<const T*>(p.z) + ch * E;
    const T* db = reinterpret_cast<const T*>(p.dy) + ch * E;
    auto accumulate = [&](const u32x4 zraw, const u32x4 draw) {
      float zf[E];
      Chunk<T>::unpack(zraw, zf);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < E; ++e) s0[e] += zf[e], s1[e] += zf[e] * zf[e];
      } else {
        float df[E];
        Chunk<T>::unpack(draw, df);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const float xh = (zf[e] - mu[e]) * rs[e];
          float du = df[e];
          if (p.act == DY_ACT_SILU) du *= silu_grad(ga[e] * xh + be[e]);
          s0[e] += du, s1[e] += du * xh;
        }
      }
    };
    // UNR rows in flight per thread: with one load per iteration the pass ran at 1.2 TB/s (latency bound)
    long long r = r0 + rr;
    for (; r + (long long)(UNR - 1) * p.R < r1; r += (long long)UNR * p.R) {
      u32x4 zr[UNR], dr[UNR];
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
        zr[k] = *reinterpret_cast<const u32x4*>(zb + (r + (long long)k * p.R) * p.ld_z);
        if (MODE == 1) dr[k] = *reinterpret_cast<const u32x4*>(db + (r + (long long)k * p.R) * p.ld_dy);
      }
#pragma unroll
      for (int k = 0; k < UNR; ++k) accumulate(zr[k], MODE == 1 ? dr[k] : zr[k]);
    }
    for (; r < r1; r += p.R)
      accumulate(*reinterpret_cast<const u32x4*>(zb + r * p.ld_z), MODE == 1 ? *reinterpret_cast<const u32x4*>(db + r * p.ld_dy) : u32x4{0u, 0u, 0u, 0u});
#pragma unroll
    for (int e = 0; e < E; ++e) {
      red[(0 * p.R + rr) * p.c + ch * E + e] = s0[e];
      red[(1 * p.R + rr) * p.c + ch * E + e] = s1[e];
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * p.c; i += 256) {
    const int which = i / p.c, cc = i - which * p.c;
    double t = 0.0;
    for (int k = 0; k < p.R; ++k) t += (double)red[(which * p.R + k) * p.c + cc];
    p.acc[(size_t)(1 + blockIdx.x) * 2 * p.c + i] = t;  // this slab's partial; summed by bn_sum_partials_kernel
  }
}

// acc[0 .. 2c) += sum over slabs of their partials (slab b at acc[(1 + b) * 2c ...)).  The first version had every slab end
// with 2c double atomics on the same addresses (1024 slabs x 128 atomics queued on 128 addresses per launch: the reduce
// pass ran at 1.2 TB/s).  Here 16 x 8 threads share the slabs of an output: 16 atomics per address.
constexpr int kBnSumY = 16;
__global__ __launch_bounds__(256) void bn_sum_partials_kernel(const BnArgs p, int slabs) {
  __shared__ double red[8][32];
  const int il = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + il;
  const int chunk = (slabs + kBnSumY - 1) / kBnSumY;
  const int b0 = blockIdx.y * chunk, b1 = (b0 + chunk < slabs) ? b0 + chunk : slabs;
  double t = 0.0;
  if (i < 2 * p.c)
    for (int b = b0 + g; b < b1; b += 8) t += p.acc[(size_t)(1 + b) * 2 * p.c + i];
  red[g][il] = t;
  __syncthreads();
  if (g == 0 && i < 2 * p.c) {
    double u = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) u += red[k][il];
    atomicAdd(p.acc + i, u);
  }
}

// y = act(gamma * (z - mean) * rstd + beta (+ addend)).  Threads keep their channel chunk (tid % nch) and walk rows, so the
// per-channel constants are loaded once into registers.  r04: the constants come straight from the double sums (mean, biased variance,
// rstd = 1 / sqrt(var + eps) in fp32 as torch's batch_norm computes invstd), and workgroup 0 leaves mean / rstd for the backward pass
// and updates the running statistics (unbiased variance, torch semantics) -- until r04 a launch of its own (bn_finalize_kernel: 74
// launches of ~5 us per training step).  UNR rows are in flight per thread.
template <typename T, int UNR>
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnArgs p) {
  constexpr int E = Elem<T>::EPC;
  const int tid = threadIdx.x;
  const int ch = tid % p.nch, rr = tid / p.nch;
  // one thread per channel computes the constants (in every thread the double-precision prologue cost more VALU time than the pass:
  // 8 waves per SIMD x ~4,000 issue cycles each), the workgroup shares them through LDS
  __shared__ float s_sc[kBnMaxC], s_sh[kBnMaxC];
  {
    const double n = (double)p.rows, invn = p.invn;
    for (int cc = tid; cc < p.c; cc += 256) {
      const double m = p.acc[cc] * invn;
      double var = p.acc[p.c + cc] * invn - m * m;
      if (var < 0.0) var = 0.0;
      const float mean = (float)m, rstd = 1.0f / sqrtf((float)var + p.eps);
      const float scv = p.gamma[cc] * rstd;
      s_sc[cc] = scv;
      s_sh[cc] = p.beta[cc] - mean * scv;
      if (blockIdx.x == 0) {
        p.mean[cc] = mean, p.rstd[cc] = rstd;
        if (p.running_mean) {
          const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
          p.running_mean[cc] = (1.f - p.momentum) * p.running_mean[cc] + p.momentum * mean;
          p.running_var[cc] = (1.f - p.momentum) * p.running_var[cc] + p.momentum * (float)unb;
        }
      }
    }
  }
  __syncthreads();
  if (rr >= p.R) return;
  float sc[E], sh[E];
#pragma unroll
  for (int e = 0; e < E; ++e) sc[e] = s_sc[ch * E + e], sh[e] = s_sh[ch * E + e];
  const T* zb = reinterpret_cast<const T*>(p.z) + ch * E;
  const T* ab = reinterpret_cast<const T*>(p.addend) + ch * E;
  T* yb = reinterpret_cast<T*>(p.y) + ch * E;
  auto one = [&](long long r, const u32x4 zraw, const u32x4 araw) {
    float zf[E], o[E];
    Chunk<T>::unpack(zraw, zf);
#pragma unroll
    for (int e = 0; e < E; ++e) o[e] = zf[e] * sc[e] + sh[e];
    if (p.addend) {
      float af[E];
      Chunk<T>::unpack(araw, af);
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] += af[e];
    }
    if (p.act == DY_ACT_SILU) {
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = silu_f32(o[e]);
    }
    *reinterpret_cast<u32x4*>(yb + r * p.ld_y) = Chunk<T>::pack(o);
  };
  const long long G = (long long)gridDim.x * p.R;
  long long r = (long long)blockIdx.x * p.R + rr;
  for (; r + (UNR - 1) * G < p.rows; r += UNR * G) {
    u32x4 zr[UNR], ar[UNR];
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
      zr[k] = *reinterpret_cast<const u32x4*>(zb + (r + k * G) * p.ld_z);
      ar[k] = p.addend ? *reinterpret_cast<const u32x4*>(ab + (r + k * G) * p.ld_add) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int k = 0; k < UNR; ++k) one(r + k * G, zr[k], ar[k]);
  }
  for (; r < p.rows; r += G)
    one(r, *reinterpret_cast<const u32x4*>(zb + r * p.ld_z), p.addend ? *reinterpret_cast<const u32x4*>(ab + r * p.ld_add) : u32x4{0u, 0u, 0u, 0u});
}

// dz = gamma * rstd * (du - mean(du) - xhat * mean(du * xhat));  dgamma = sum du*xhat, dbeta = sum du.  UNR rows in flight per thread.
template <typename T, int UNR>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnArgs p) {
  constexpr int E = Elem<T>::EPC;
  const int tid = threadIdx.x;
  if (blockIdx.x == 0) {
    for (int cc = tid; cc < p.c; cc += 256) {
      if (p.dbeta) p.dbeta[cc] = (float)p.acc[cc];
      if (p.dgamma) p.dgamma[cc] = (float)p.acc[p.c + cc];
    }
  }
  const int ch = tid % p.nch, rr = tid / p.nch;
  if (rr >= p.R) return;
  const double invn = p.invn;
  float mu[E], rs[E], ga[E], be[E], gr[E], mdu[E], mdx[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int cc = ch * E + e;
    mu[e] = p.mean[cc], rs[e] = p.rstd[cc], ga[e] = p.gamma[cc], be[e] = p.beta[cc];
    gr[e] = ga[e] * rs[e];
    mdu[e] = (float)(p.acc[cc] * invn), mdx[e] = (float)(p.acc[p.c + cc] * invn);
  }
  const T* zb = reinterpret_cast<const T*>(p.z) + ch * E;
  const T* db = reinterpret_cast<const T*>(p.dy) + ch * E;
  T* ob = reinterpret_cast<T*>(p.dz) + ch * E;
  auto one = [&](long long r, const u32x4 zraw, const u32x4 draw) {
    float zf[E], df[E], o[E];
    Chunk<T>::unpack(zraw, zf);
    Chunk<T>::unpack(draw, df);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float xh = (zf[e] - mu[e]) * rs[e];
      float du = df[e];
      if (p.act == DY_ACT_SILU) du *= silu_grad(ga[e] * xh + be[e]);
      o[e] = gr[e] * (du - mdu[e] - xh * mdx[e]);
    }
    *reinterpret_cast<u32x4*>(ob + r * p.ld_dz) = Chunk<T>::pack(o);
  };
  const long long G = (long long)gridDim.x * p.R;
  long long r = (long long)blockIdx.x * p.R + rr;
  for (; r + (UNR - 1) * G < p.rows; r += UNR * G) {
    u32x4 zr[UNR], dr[UNR];
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
      zr[k] = *reinterpret_cast<const u32x4*>(zb + (r + k * G) * p.ld_z);
      dr[k] = *reinterpret_cast<const u32x4*>(db + (r + k * G) * p.ld_dy);
    }
#pragma unroll
    for (int k = 0; k < UNR; ++k) one(r + k * G, zr[k], dr[k]);
  }
  for (; r < p.rows; r += G) one(r, *reinterpret_cast<const u32x4*>(zb + r * p.ld_z), *reinterpret_cast<const u32x4*>(db + r * p.ld_dy));
}

// y = silu(u)  /  du = dy * silu'(u)   (RepVGG: the activation after the sum of two BN branches)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void silu_kernel(const void* u_, const void* dy_, void* out_, long long rows, int nch, int ld_u, int ld_dy, int ld_o) {
  constexpr int E = Elem<T>::EPC;
  const long long total = rows * nch;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / nch;
    const int ch = (int)(i - r * nch);
    float uf[E], o[E];
    Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(u_) + r * ld_u + ch * E), uf);
    if (BWD) {
      float df[E];
      Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(dy_) + r * ld_dy + ch * E), df);
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = df[e] * silu_grad(uf[e]);
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = uf[e] / (1.0f + __expf(-uf[e]));
    }
    *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(out_) + r * ld_o + ch * E) = Chunk<T>::pack(o);
  }
}

static int bn_prepare(const dy_bn_desc* d, BnArgs* a, const char* who, bool bwd) {
  DY_REQUIRE(d && d->z && d->gamma && d->beta && d->mean && d->rstd && d->workspace, DY_ERR_INVALID_ARG, "%s: null pointer", who);
  const int es = dtype_size_no_fp8(d->dtype);
  DY_REQUIRE(es != 0 && d->rows > 0 && d->c > 0, DY_ERR_INVALID_ARG, "%s: bad dtype/rows/c", who);
  const int epc = 16 / es;
  DY_REQUIRE(d->c % epc == 0 && d->c <= 8192, DY_ERR_UNSUPPORTED, "%s: c %d must be a multiple of %d (one 16-byte chunk) and <= 8192", who, d->c, epc);
  DY_REQUIRE(aligned16(d->z) && d->ld_z >= d->c && (d->ld_z * es) % 16 == 0, DY_ERR_INVALID_ARG, "%s: z view misaligned", who);
  DY_REQUIRE(d->workspace_bytes >= dy_bn_workspace_bytes(d->c) && (reinterpret_cast<uintptr_t>(d->workspace) & 7) == 0, DY_ERR_WORKSPACE,
             "%s: workspace needs %lld bytes (dy_bn_workspace_bytes)", who, (long long)dy_bn_workspace_bytes(d->c));
  if (bwd) {
    DY_REQUIRE(d->dy && d->dz && aligned16(d->dy) && aligned16(d->dz) && d->ld_dy >= d->c && d->ld_dz >= d->c && (d->ld_dy * es) % 16 == 0 &&
                   (d->ld_dz * es) % 16 == 0, DY_ERR_INVALID_ARG, "%s: dy/dz views null or misaligned", who);
  } else {
    DY_REQUIRE(d->y && aligned16(d->y) && d->ld_y >= d->c && (d->ld_y * es) % 16 == 0, DY_ERR_INVALID_ARG, "%s: y view null or misaligned", who);
    DY_REQUIRE(!d->addend || (aligned16(d->addend) && d->ld_add >= d->c && (d->ld_add * es) % 16 == 0), DY_ERR_INVALID_ARG, "%s: addend misaligned", who);
  }
  a->z = d->z, a->y = d->y, a->addend = d->addend, a->dy = d->dy, a->dz = d->dz;
  a->rows = d->rows, a->c = d->c, a->ld_z = d->ld_z, a->ld_y = d->ld_y, a->ld_add = d->ld_add, a->ld_dy = d->ld_dy, a->ld_dz = d->ld_dz;
  a->act = d->act, a->gamma = d->gamma, a->beta = d->beta, a->mean = d->mean, a->rstd = d->rstd;
  a->running_mean = d->running_mean, a->running_var = d->running_mean ? d->running_var : nullptr;
  a->eps = d->eps, a->momentum = d->momentum, a->dgamma = d->dgamma, a->dbeta = d->dbeta;
  a->acc = reinterpret_cast<double*>(d->workspace);
  a->partial_slabs = d->partial_slabs;
  DY_REQUIRE(a->partial_slabs >= 0 && a->partial_slabs <= kBnMaxSlabs, DY_ERR_INVALID_ARG, "%s: partial_slabs out of range", who);
  a->nch = d->c / epc;
  a->invn = 1.0 / (double)d->rows;
  int R = 256 / a->nch;
  if (R < 1) R = 1;
  a->R = R;
  DY_REQUIRE(a->nch <= 256, DY_ERR_UNSUPPORTED, "%s: c %d too wide for one workgroup row (max %d)", who, d->c, 256 * epc);
  // slabs: enough workgroups to fill the chip (four per CU), at least 4 passes of R rows each
  long long blocks = (d->rows + 4LL * R - 1) / (4LL * R);
  if (blocks > kBnMaxSlabs) blocks = kBnMaxSlabs;
  a->rows_per_block = (int)((d->rows + blocks - 1) / blocks);
  return 0;
}

// rows per pass of an apply grid = blocks * R; at most `cap` workgroups (the grid-stride loop takes the rest).  Measured per kernel over
// the eight BatchNorm shapes of the B = 64 training step (tools/bn_prof.sh, buffers rotating through 1 GB): 1 / 2 / 4 rows in flight and
// 1024 / 2048 / 4096 workgroups all land within 5 % of each other (apply 20.2-21.4 us, backward apply 30.0-31.8 us on average): both
// passes sit at 4.3-4.7 TB/s of the box's 5.4 TB/s copy rate, the backward one co-limited by its two quarter-rate transcendentals per
// element.  What did matter (49 -> 21 us): the statistics prologue in ONE thread per channel instead of every thread.
static unsigned apply_blocks(const BnArgs& a, int unr, int cap) {
  const long long nb = (a.rows + (long long)a.R * unr - 1) / ((long long)a.R * unr);
  return (unsigned)(nb < cap ? (nb < 1 ? 1 : nb) : cap);
}

template <typename T>
static int bn_fwd_t(const BnArgs& a, hipStream_t st) {
  const unsigned blocks = (unsigned)((a.rows + a.rows_per_block - 1) / a.rows_per_block);
  const size_t smem = (size_t)2 * a.R * a.c * 4;
  if (a.partial_slabs <= 0) hipLaunchKernelGGL((bn_reduce_kernel<T, 0, 8>), dim3(blocks), dim3(256), smem, st, a);
  hipLaunchKernelGGL(bn_sum_partials_kernel, dim3((unsigned)((2 * a.c + 31) / 32), kBnSumY), dim3(256), 0, st, a, a.partial_slabs > 0 ? a.partial_slabs : (int)blocks);
  const int unr = dy_ablate("DYOLO_BN_UNR") ? dy_ablate("DYOLO_BN_UNR") : 2;
  const int cap = dy_ablate("DYOLO_BN_GRID") ? dy_ablate("DYOLO_BN_GRID") : 2048;
  const unsigned ab = apply_blocks(a, unr, cap);
  if (unr == 1) hipLaunchKernelGGL((bn_apply_kernel<T, 1>), dim3(ab), dim3(256), 0, st, a);
  else if (unr == 2) hipLaunchKernelGGL((bn_apply_kernel<T, 2>), dim3(ab), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((bn_apply_kernel<T, 4>), dim3(ab), dim3(256), 0, st, a);
  return check_launch("dy_bn_train_fwd");
}

template <typename T>
static int bn_bwd_t(const BnArgs& a, hipStream_t st) {
  const unsigned blocks = (unsigned)((a.rows + a.rows_per_block - 1) / a.rows_per_block);
  const size_t smem = (size_t)2 * a.R * a.c * 4;
  if (a.partial_slabs > 0) {
    // the input-gradient convolution behind left the slots of du and du * xhat (dy_conv_desc.bnb_z): no pass over dy and z
  } else if (dy_ablate("DYOLO_BN_RUNR") == 8) {
    hipLaunchKernelGGL((bn_reduce_kernel<T, 1, 8>), dim3(blocks), dim3(256), smem, st, a);
  } else {
    hipLaunchKernelGGL((bn_reduce_kernel<T, 1, 4>), dim3(blocks), dim3(256), smem, st, a);
  }
  hipLaunchKernelGGL(bn_sum_partials_kernel, dim3((unsigned)((2 * a.c + 31) / 32), kBnSumY), dim3(256), 0, st, a, a.partial_slabs > 0 ? a.partial_slabs : (int)blocks);
  const int unr = dy_ablate("DYOLO_BN_BUNR") ? dy_ablate("DYOLO_BN_BUNR") : 2;
  const int cap = dy_ablate("DYOLO_BN_GRID") ? dy_ablate("DYOLO_BN_GRID") : 2048;
  const unsigned ab = apply_blocks(a, unr, cap);
  if (unr == 1) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 1>), dim3(ab), dim3(256), 0, st, a);
  else if (unr == 2) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 2>), dim3(ab), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 4>), dim3(ab), dim3(256), 0, st, a);
  return check_launch("dy_bn_train_bwd");
}

}  // namespace dy

using namespace dy;

extern "C" int64_t dy_bn_workspace_bytes(int32_t c) { return c <= 0 ? -1 : (int64_t)(1 + kBnMaxSlabs) * 2 * c * 8; }  // sums + one partial per slab

extern "C" int32_t dy_bn_train_fwd(const dy_bn_desc* d, dy_stream_t stream) {
  BnArgs a{};
  const int rc = bn_prepare(d, &a, "dy_bn_train_fwd", false);
  if (rc) return rc;
  DY_REQUIRE((size_t)2 * a.R * a.c * 4 <= 160 * 1024, DY_ERR_UNSUPPORTED, "dy_bn_train_fwd: reduction scratch exceeds LDS");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (d->dtype) {
    case DY_BF16: return bn_fwd_t<bf16_t>(a, st);
    case DY_F16: return bn_fwd_t<f16_t>(a, st);
    default: return bn_fwd_t<float>(a, st);
  }
}

extern "C" int32_t dy_bn_train_bwd(const dy_bn_desc* d, dy_stream_t stream) {
  BnArgs a{};
  const int rc = bn_prepare(d, &a, "dy_bn_train_bwd", true);
  if (rc) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (d->dtype) {
    case DY_BF16: return bn_bwd_t<bf16_t>(a, st);
    case DY_F16: return bn_bwd_t<f16_t>(a, st);
    default: return bn_bwd_t<float>(a, st);
  }
}

template <bool BWD>
static int silu_launch(const void* u, const void* dy, void* out, int64_t rows, int32_t c, int32_t ld_u, int32_t ld_dy, int32_t ld_o, int32_t dtype,
                       hipStream_t st, const char* who) {
  const int es = dtype_size_no_fp8(dtype);
  DY_REQUIRE(es && u && out && (!BWD || dy) && rows > 0 && c > 0, DY_ERR_INVALID_ARG, "%s: null pointer or bad dims", who);
  const int epc = 16 / es;
  DY_REQUIRE(c % epc == 0 && aligned16(u) && aligned16(out) && (ld_u * es) % 16 == 0 && (ld_o * es) % 16 == 0 && ld_u >= c && ld_o >= c &&
                 (!BWD || (aligned16(dy) && (ld_dy * es) % 16 == 0 && ld_dy >= c)), DY_ERR_INVALID_ARG, "%s: views must be 16-byte aligned chunks", who);
  const int nch = c / epc;
  const long long total = rows * nch;
  const unsigned ab = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  switch (dtype) {
    case DY_BF16: hipLaunchKernelGGL((silu_kernel<bf16_t, BWD>), dim3(ab), dim3(256), 0, st, u, dy, out, (long long)rows, nch, ld_u, ld_dy, ld_o); break;
    case DY_F16: hipLaunchKernelGGL((silu_kernel<f16_t, BWD>), dim3(ab), dim3(256), 0, st, u, dy, out, (long long)rows, nch, ld_u, ld_dy, ld_o); break;
    default: hipLaunchKernelGGL((silu_kernel<float, BWD>), dim3(ab), dim3(256), 0, st, u, dy, out, (long long)rows, nch, ld_u, ld_dy, ld_o); break;
  }
  return check_launch(who);
}

extern "C" int32_t dy_silu_fwd(const void* u, void* y, int64_t rows, int32_t c, int32_t ld_u, int32_t ld_y, int32_t dtype, dy_stream_t stream) {
  return silu_launch<false>(u, nullptr, y, rows, c, ld_u, 0, ld_y, dtype, reinterpret_cast<hipStream_t>(stream), "dy_silu_fwd");
}

extern "C" int32_t dy_silu_bwd(const void* u, const void* dy, void* du, int64_t rows, int32_t c, int32_t ld_u, int32_t ld_dy, int32_t ld_du,
                               int32_t dtype, dy_stream_t stream) {
  return silu_launch<true>(u, dy, du, rows, c, ld_u, ld_dy, ld_du, dtype, reinterpret_cast<hipStream_t>(stream), "dy_silu_bwd");
}
