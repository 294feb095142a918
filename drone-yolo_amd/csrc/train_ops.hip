// Small bandwidth-bound kernels of the training path: gradients of nn.Upsample(None, 2, 'nearest')
// (yolov8-p2-repvgg.yaml:30,34,38), of SPPF's MaxPool2d(k, 1, k//2) (nn/modules/block.py:185-191), the element-wise add
// of Bottleneck's shortcut (block.py:348-350), and the optimizer / EMA / clipping updates of the trainer
// (engine/trainer.py:591-599, 764-825; utils/torch_utils.py:515-545).
#include "common_hip.h"

namespace dy {

// ---- dx[n,h,w,:] = sum of the 2x2 block of g it was copied to -----------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const T* __restrict__ g, T* __restrict__ dx, int n, int h, int w, int cchunks, int ldg, int ldx) {
  constexpr int E = Elem<T>::EPC;
  const long long total = (long long)n * h * w * cchunks;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % cchunks);
    long long t = i / cchunks;
    const int x = (int)(t % w);
    t /= w;
    const int y = (int)(t % h);
    const int img = (int)(t / h);
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
#pragma unroll
    for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
      for (int dxx = 0; dxx < 2; ++dxx) {
        float f[E];
        Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(g + ((size_t)(img * 2 * h + 2 * y + dyy) * (2 * w) + 2 * x + dxx) * (size_t)ldg + cc * E), f);
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += f[e];
      }
    *reinterpret_cast<u32x4*>(dx + ((size_t)(img * h + y) * w + x) * (size_t)ldx + cc * E) = Chunk<T>::pack(acc);
  }
}

// ---- max-pool k x k, stride 1, pad k/2: gradient goes to the FIRST maximum of each window in (row, column) scan order,
// as torch's max_pool2d does.  One workgroup per (image, 16-byte channel chunk); the plane of x and of the incoming gradient sit in LDS
// as they come -- one 16-byte chunk (8 or 4 channels) per position -- and a thread handles a POSITION with its channels in registers:
//   1a. per position the maximum of its ROW segment (columns xx - r .. xx + r) and the first column that reaches it (4 bits a channel);
//   1b. the window's arg-max from the row segments: the first ROW whose segment maximum equals the window maximum, at that segment's
//       first column -- 2 (2r + 1) chunk reads per output instead of (2r + 1)^2 scalar ones; then the output's gradient ADDED at that
//       input position (LDS float atomics; outputs that share a target add in arrival order);
//   2.  sums (+ the gradient already held) -> g_in.
// (r02: scalar fp32 planes, a thread per (position, channel) with dependent LDS reads per tap, 54 KB of LDS: two workgroups per CU and
// 126-190 us for a 13 MB map.)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ go, T* __restrict__ gi, int h, int w, int cchunks,
                                                          int ldx, int ldgo, int ldgi, int r, int accumulate) {
  constexpr int E = Elem<T>::EPC;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  const int hw = h * w;
  u32x4* xs = reinterpret_cast<u32x4*>(dyn_smem);  // [hw] chunks of x
  u32x4* gs = xs + hw;                             // [hw] chunks of the incoming gradient
  u32x4* rm = gs + hw;                             // [hw] row-segment maxima (values of x: exact in T)
  float* sums = reinterpret_cast<float*>(rm + hw); // [hw][E]
  unsigned* ra = reinterpret_cast<unsigned*>(sums + (size_t)hw * E);  // [hw] 4 bits per channel: column offset dx + r of the segment's first maximum
  const int img = blockIdx.x / cchunks, cc = blockIdx.x - img * cchunks;
  for (int p = threadIdx.x; p < hw; p += 256) {
    xs[p] = *reinterpret_cast<const u32x4*>(x + ((size_t)img * hw + p) * (size_t)ldx + cc * E);
    gs[p] = *reinterpret_cast<const u32x4*>(go + ((size_t)img * hw + p) * (size_t)ldgo + cc * E);
#pragma unroll
    for (int e = 0; e < E; ++e) sums[p * E + e] = 0.f;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < hw; p += 256) {
    const int yy = p / w, xx = p - yy * w;
    float best[E];
    unsigned arg = 0;
    bool first = true;
    for (int dxx = -r; dxx <= r; ++dxx) {
      const int x2 = xx + dxx;
      if ((unsigned)x2 >= (unsigned)w) continue;
      float v[E];
      Chunk<T>::unpack(xs[yy * w + x2], v);
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (first || v[e] > best[e]) best[e] = v[e], arg = (arg & ~(15u << (4 * e))) | ((unsigned)(dxx + r) << (4 * e));
      first = false;
    }
    rm[p] = Chunk<T>::pack(best);
    ra[p] = arg;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < hw; p += 256) {
    const int yy = p / w, xx = p - yy * w;
    float best[E], g[E];
    int tgt[E];  // input position of the channel's arg-max
    bool first = true;
    for (int dyy = -r; dyy <= r; ++dyy) {
      const int y2 = yy + dyy;
      if ((unsigned)y2 >= (unsigned)h) continue;
      float v[E];
      Chunk<T>::unpack(rm[y2 * w + xx], v);
      const unsigned arg = ra[y2 * w + xx];
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (first || v[e] > best[e]) best[e] = v[e], tgt[e] = y2 * w + xx + (int)((arg >> (4 * e)) & 15u) - r;
      first = false;
    }
    Chunk<T>::unpack(gs[p], g);
#pragma unroll
    for (int e = 0; e < E; ++e) atomicAdd(sums + tgt[e] * E + e, g[e]);
  }
  __syncthreads();
  for (int p = threadIdx.x; p < hw; p += 256) {
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = sums[p * E + e];
    T* dst = gi + ((size_t)img * hw + p) * (size_t)ldgi + cc * E;
    if (accumulate) {
      float f[E];
      Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(dst), f);
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] += f[e];
    }
    *reinterpret_cast<u32x4*>(dst) = Chunk<T>::pack(acc);
  }
}

// ---- gradient of the Detect head's fp32 training map, split for the two 1x1 convolutions behind it ------------------------------
// g: (rows, nb + nc) fp32, pitch ld_g -- what the loss hands back for cat(box logits, class logits) (head.py:69-72).  One pass writes
// the 16-bit operands of the weight / input gradient kernels: dzb = scale * g[:, :nb], dzc = scale * g[:, nb : nb + nc] zero-padded
// to ncp channels.  scale: optional DEVICE scalar (the seed of backward(): the loss scale under fp16).  Replaces a torch multiply,
// two casting copies and a zero fill per level.
template <typename T>
__global__ __launch_bounds__(256) void head_grad_split_kernel(const float* __restrict__ g, int ld_g, long long rows, int nb, int nc, int ncp, const float* __restrict__ scale,
                                                              T* __restrict__ dzb, int ld_b, T* __restrict__ dzc, int ld_c) {
  constexpr int E = Elem<T>::EPC;
  const int cb = nb / E, cc = ncp / E, per_row = cb + cc;
  const float sc = scale ? *scale : 1.0f;
  const bool vec = ld_g >= nb + ncp && ld_g % 4 == 0 && nb % 4 == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0;
  const long long total = rows * per_row;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / per_row;
    const int k = (int)(i - r * per_row);
    const bool box = k < cb;
    const int c0 = box ? k * E : (k - cb) * E;       // first channel of this chunk inside its slot
    const int lim = box ? nb : nc;                   // channels of the slot that exist
    const float* src = g + r * ld_g + (box ? 0 : nb) + c0;
    float f[E];
    if (vec) {  // whole 16-byte loads: the slot's padding lies inside the row pitch (host check)
#pragma unroll
      for (int e = 0; e < E; e += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) f[e + j] = c0 + e + j < lim ? v[j] * sc : 0.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) f[e] = c0 + e < lim ? src[e] * sc : 0.f;
    }
    T* dst = box ? dzb + r * ld_b + c0 : dzc + r * ld_c + c0;
    *reinterpret_cast<u32x4*>(dst) = Chunk<T>::pack(f);
  }
}

// ---- dx[n][2y][2x][:] += t[n][y][x][:]: the input gradient of a 1x1 STRIDE-2 convolution (RepVGGBlock's side branch, block.py:1480-1490)
// reaches only the even positions.  t = the 1x1 stride-1 convolution of dz with the transposed weights (a quarter of the pixels of
// dx); the zero-dilated gather it replaces ran the generic implicit-GEMM kernel over ALL pixels of dx, three quarters of them for zeros.
template <typename T>
__global__ __launch_bounds__(256) void add_dilated2_kernel(const T* __restrict__ t, T* __restrict__ dx, int n, int h, int w, int H2, int W2, int cchunks, int ld_t, int ld_dx) {
  constexpr int E = Elem<T>::EPC;
  const long long total = (long long)n * h * w * cchunks;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % cchunks);
    long long r = i / cchunks;
    const int x = (int)(r % w);
    r /= w;
    const int y = (int)(r % h), img = (int)(r / h);
    float a[E], b[E];
    Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(t + (((long long)img * h + y) * w + x) * ld_t + cc * E), a);
    T* dst = dx + (((long long)img * H2 + 2 * y) * W2 + 2 * x) * ld_dx + cc * E;
    Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(dst), b);
#pragma unroll
    for (int e = 0; e < E; ++e) a[e] += b[e];
    *reinterpret_cast<u32x4*>(dst) = Chunk<T>::pack(a);
  }
}

// ---- out = a + b on (rows, c) views ----------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long long rows, int cchunks, int lda, int ldb, int ldo) {
  constexpr int E = Elem<T>::EPC;
  const long long total = rows * cchunks;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / cchunks;
    const int cc = (int)(i - r * cchunks);
    float fa[E], fb[E];
    Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(a + r * lda + cc * E), fa);
    Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(b + r * ldb + cc * E), fb);
#pragma unroll
    for (int e = 0; e < E; ++e) fa[e] += fb[e];
    *reinterpret_cast<u32x4*>(o + r * ldo + cc * E) = Chunk<T>::pack(fa);
  }
}

// ---- optimizer over ONE flat fp32 tensor ------------------------------------------------------------------------------
// sum of squares (double) for clip_grad_norm_
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, double* out) {
  __shared__ double red[4];
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += (double)g[i] * (double)g[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  // ONE atomic per workgroup, 256 workgroups: 4,096 wave atomics on the one address took most of the kernel's 60 us
  if (threadIdx.x == 0) {
    const double t = red[0] + red[1] + red[2] + red[3];
    if (t != 0.0) atomicAdd(out, t);
  }
}

// Loss-scaling state of torch.cuda.amp.GradScaler kept on the device (reference: engine/trainer.py:271, 389, 591-599):
//   amp[0] scale, amp[1] growth tracker (unskipped steps since the last change), amp[2] found_inf of the last step, amp[3] skipped steps.
// With `amp` the optimizer kernels fold `scaler.unscale_` into the step (grad / scale; the clip norm is sqrt(sumsq) / scale) and skip
// the step when the squared-gradient sum is not finite, as `scaler.step` does; dy_amp_update is `scaler.update()`.
struct StepScale {
  float inv, clip;
  bool skip;
};
__device__ __forceinline__ StepScale step_scale(const double* sumsq, float max_norm, const float* amp) {
  StepScale r{1.f, 1.f, false};
  if (amp) r.inv = 1.f / amp[0];
  if (sumsq) {
    const double ss = *sumsq;
    if (amp && !(ss == ss && ss < 1.7e308)) {
      r.skip = true;
      return r;
    }
    const float c = max_norm / ((float)sqrt(ss) * r.inv + 1e-6f);
    r.clip = c < 1.f ? c : 1.f;
  }
  return r;
}

// torch.optim.SGD (momentum, nesterov, weight decay) with the clip coefficient folded in:
//   g = clip * grad + wd * p;  buf = first ? g : mom * buf + g;  p -= lr * (nesterov ? g + mom * buf : buf)
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ grad, float* __restrict__ buf, long long n, float lr, float mom,
                                                  float wd, int nesterov, int first, const double* sumsq, float max_norm, const float* amp) {
  const StepScale sc = step_scale(sumsq, max_norm, amp);
  if (sc.skip) return;
  // (a skipped first step leaves buf at its zeros, so the next step's mom * buf + g equals the "first" form: no extra state)
  const float k = sc.inv * sc.clip;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float g = grad[i] * k + wd * p[i];
    const float b = first ? g : mom * buf[i] + g;
    buf[i] = b;
    g = nesterov ? g + mom * b : b;
    p[i] -= lr * g;
  }
}

// torch.optim.AdamW: p *= 1 - lr*wd;  m, v moments;  p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v, long long n,
                                                    float lr, float b1, float b2, float eps, float wd, int step, const double* sumsq, float max_norm, const float* amp) {
  const StepScale sc = step_scale(sumsq, max_norm, amp);
  if (sc.skip) return;
  const float t = (float)(step - (amp ? (int)amp[3] : 0));  // optimizer.step() calls that really ran (skipped ones leave state['step'] alone)
  const float bc1 = 1.f - powf(b1, t), bc2s = sqrtf(1.f - powf(b2, t));
  const float k = sc.inv * sc.clip;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float g = grad[i] * k;
    float pp = p[i] * (1.f - lr * wd);
    const float mm = m[i] * b1 + (1.f - b1) * g;
    const float vv = v[i] * b2 + (1.f - b2) * g * g;
    m[i] = mm;
    v[i] = vv;
    pp -= (lr / bc1) * mm / (sqrtf(vv) / bc2s + eps);
    p[i] = pp;
  }
}

// GradScaler.update(): found_inf -> scale *= backoff, tracker = 0; else tracker += 1 and scale *= growth every `interval` clean steps
__global__ void amp_update_kernel(float* amp, const double* sumsq, float growth, float backoff, int interval) {
  if (threadIdx.x || blockIdx.x) return;
  const double ss = *sumsq;
  const bool bad = !(ss == ss && ss < 1.7e308);
  amp[2] = bad ? 1.f : 0.f;
  if (bad) {
    amp[0] *= backoff;
    amp[1] = 0.f;
    amp[3] += 1.f;
  } else {
    const float t = amp[1] + 1.f;
    if ((int)t >= interval) {
      amp[0] *= growth;
      amp[1] = 0.f;
    } else {
      amp[1] = t;
    }
  }
}

// ModelEMA.update: e = d * e + (1 - d) * p
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ e, const float* __restrict__ p, long long n, float d) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) e[i] = e[i] * d + (1.f - d) * p[i];
}

static inline unsigned grid1(long long items) {
  long long b = (items + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}


// ---- grouped convolution gradients (DWConv of the -sf YAML, nn/modules/conv.py:102-107: g = gcd(c1, c2), k 3, stride 2) ----------
// Tiny (0.026 GFLOP / image forward) and bandwidth bound, so direct kernels.  Weights and their gradient are the fp32 master
// tensors in torch's own (cout, cin/g, k, k) layout: no packing step, the gradient lands where the optimizer reads it.
// wgrad: one thread per (pixel slab, output channel): consecutive lanes = consecutive channels, so the dz row (64 channels x 2 B) and
// the x row of the matching groups are read as whole lines; CPG_IN*K*K partial sums in registers, added with fp32 atomics.
template <typename T, int CPG_IN, int K>
__global__ __launch_bounds__(256) void conv_grouped_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dz, float* __restrict__ dw, int n, int h, int w, int cin,
                                                                 int ldx, int ho, int wo, int cout, int lddz, int stride, int pad, int groups, int slab) {
  const int co = blockIdx.y * 64 + (threadIdx.x & 63);
  const long long m_total = (long long)n * ho * wo;
  const long long m0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * slab;
  if (co >= cout || m0 >= m_total) return;
  const int cpg_out = cout / groups;
  const int cbase = (co / cpg_out) * CPG_IN;
  float acc[K * K * CPG_IN];
#pragma unroll
  for (int i = 0; i < K * K * CPG_IN; ++i) acc[i] = 0.f;
  const long long m1 = m0 + slab < m_total ? m0 + slab : m_total;
  for (long long m = m0; m < m1; ++m) {
    const int img = (int)(m / ((long long)ho * wo));
    const int rem = (int)(m - (long long)img * ho * wo);
    const int oy = rem / wo, ox = rem - oy * wo;
    const float g = Elem<T>::to_f32(dz[(size_t)m * (size_t)lddz + co]);
#pragma unroll
    for (int r = 0; r < K; ++r) {
      const int iy = oy * stride - pad + r;
      if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const int ix = ox * stride - pad + q;
        if ((unsigned)ix >= (unsigned)w) continue;
        const T* xp = x + ((size_t)(img * h + iy) * w + ix) * (size_t)ldx + cbase;
#pragma unroll
        for (int c = 0; c < CPG_IN; ++c) acc[(c * K + r) * K + q] += g * Elem<T>::to_f32(xp[c]);
      }
    }
  }
  float* out = dw + (size_t)co * (CPG_IN * K * K);
#pragma unroll
  for (int i = 0; i < K * K * CPG_IN; ++i) atomicAdd(out + i, acc[i]);
}

// dgrad: one thread per (input pixel, input channel): dx = sum over the taps whose output pixel exists of dz[out pixel][co] * w[co][ci_local][r][q]
template <typename T>
__global__ __launch_bounds__(256) void conv_grouped_dgrad_kernel(const T* __restrict__ dz, const float* __restrict__ wt, const T* __restrict__ acc_in, T* __restrict__ dx, int n,
                                                                 int h, int w, int cin, int lddx, int ldacc, int ho, int wo, int cout, int lddz, int k, int stride, int pad,
                                                                 int groups) {
  const long long total = (long long)n * h * w * cin;
  const int cpg_in = cin / groups, cpg_out = cout / groups;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ci = (int)(i % cin);
    long long t = i / cin;
    const int ix = (int)(t % w);
    t /= w;
    const int iy = (int)(t % h);
    const int img = (int)(t / h);
    const int g = ci / cpg_in, cil = ci - g * cpg_in;
    float acc = 0.f;
    for (int r = 0; r < k; ++r) {
      const int ty = iy + pad - r;
      if (ty < 0 || ty % stride) continue;
      const int oy = ty / stride;
      if (oy >= ho) continue;
      for (int q = 0; q < k; ++q) {
        const int tx = ix + pad - q;
        if (tx < 0 || tx % stride) continue;
        const int ox = tx / stride;
        if (ox >= wo) continue;
        const T* gp = dz + ((size_t)(img * ho + oy) * wo + ox) * (size_t)lddz + g * cpg_out;
        for (int j = 0; j < cpg_out; ++j)
          acc += Elem<T>::to_f32(gp[j]) * wt[(((size_t)(g * cpg_out + j) * cpg_in + cil) * k + r) * k + q];
      }
    }
    const size_t pix = ((size_t)(img * h + iy) * w + ix);
    if (acc_in) acc += Elem<T>::to_f32(acc_in[pix * (size_t)ldacc + ci]);
    dx[pix * (size_t)lddx + ci] = Elem<T>::from_f32(acc);
  }
}

}  // namespace dy

using namespace dy;

#define DY_VIEW_OK(p, ld, c, es) (aligned16(p) && (ld) >= (c) && ((ld) * (es)) % 16 == 0)

extern "C" int32_t dy_upsample2x_bwd_nhwc(const void* g, void* dx, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ld_g, int32_t ld_dx, int32_t dtype,
                                          dy_stream_t stream) {
  const int es = dtype_size_no_fp8(dtype);
  DY_REQUIRE(es && g && dx && n > 0 && h > 0 && w > 0 && c > 0, DY_ERR_INVALID_ARG, "dy_upsample2x_bwd_nhwc: bad arguments");
  const int epc = 16 / es;
  DY_REQUIRE(c % epc == 0 && DY_VIEW_OK(g, ld_g, c, es) && DY_VIEW_OK(dx, ld_dx, c, es), DY_ERR_INVALID_ARG, "dy_upsample2x_bwd_nhwc: views must be whole 16-byte chunks");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int cch = c / epc;
  const unsigned grid = grid1((long long)n * h * w * cch);
  if (dtype == DY_BF16) hipLaunchKernelGGL((upsample2x_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)g, (bf16_t*)dx, n, h, w, cch, ld_g, ld_dx);
  else if (dtype == DY_F16) hipLaunchKernelGGL((upsample2x_bwd_kernel<f16_t>), dim3(grid), dim3(256), 0, st, (const f16_t*)g, (f16_t*)dx, n, h, w, cch, ld_g, ld_dx);
  else hipLaunchKernelGGL((upsample2x_bwd_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)g, (float*)dx, n, h, w, cch, ld_g, ld_dx);
  return check_launch("dy_upsample2x_bwd_nhwc");
}

extern "C" int32_t dy_maxpool_bwd_nhwc(const void* x, const void* g_out, void* g_in, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ld_x, int32_t ld_go,
                                       int32_t ld_gi, int32_t k, int32_t accumulate, int32_t dtype, dy_stream_t stream) {
  const int es = dtype_size_no_fp8(dtype);
  DY_REQUIRE(es && x && g_out && g_in && n > 0 && h > 0 && w > 0 && c > 0 && k >= 1 && (k & 1) && k <= 15, DY_ERR_INVALID_ARG, "dy_maxpool_bwd_nhwc: bad arguments");
  const int epc = 16 / es;
  DY_REQUIRE(c % epc == 0 && DY_VIEW_OK(x, ld_x, c, es) && DY_VIEW_OK(g_out, ld_go, c, es) && DY_VIEW_OK(g_in, ld_gi, c, es), DY_ERR_INVALID_ARG,
             "dy_maxpool_bwd_nhwc: views must be whole 16-byte chunks");
  const size_t smem = (size_t)h * w * (3 * 16 + epc * 4 + 4);  // chunks of x, g, row maxima; fp32 sums; packed row arg-max
  DY_REQUIRE(k <= 15, DY_ERR_UNSUPPORTED, "dy_maxpool_bwd_nhwc: k <= 15 (4-bit column offsets)");
  DY_REQUIRE(smem <= 160 * 1024, DY_ERR_UNSUPPORTED, "dy_maxpool_bwd_nhwc: plane %dx%d too large for LDS", h, w);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned)(n * (c / epc));
#define DY_MPB(T)                                                                                                                               \
  do {                                                                                                                                          \
    static const hipError_t once = hipFuncSetAttribute((const void*)maxpool_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    (void)once;                                                                                                                                 \
    hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(grid), dim3(256), smem, st, (const T*)x, (const T*)g_out, (T*)g_in, h, w, c / epc, ld_x, ld_go, ld_gi, k / 2, accumulate); \
  } while (0)
  if (dtype == DY_BF16) DY_MPB(bf16_t);
  else if (dtype == DY_F16) DY_MPB(f16_t);
  else DY_MPB(float);
#undef DY_MPB
  return check_launch("dy_maxpool_bwd_nhwc");
}

extern "C" int32_t dy_head_grad_split(const float* g, int32_t ld_g, int64_t rows, int32_t nb, int32_t nc, int32_t ncp, const float* scale, void* dzb, int32_t ld_b,
                                      void* dzc, int32_t ld_c, int32_t dtype, dy_stream_t stream) {
  const int es = dtype_size_no_fp8(dtype);
  DY_REQUIRE(g && dzb && dzc && rows > 0 && nb > 0 && nc > 0 && ncp >= nc && (es == 2 || es == 4), DY_ERR_INVALID_ARG, "dy_head_grad_split: bad arguments");
  const int epc = 16 / es;
  DY_REQUIRE(nb % epc == 0 && ncp % epc == 0 && ld_g >= nb + nc && DY_VIEW_OK(dzb, ld_b, nb, es) && DY_VIEW_OK(dzc, ld_c, ncp, es), DY_ERR_INVALID_ARG,
             "dy_head_grad_split: slots must be whole 16-byte chunks");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned grid = grid1(rows * ((nb + ncp) / epc));
  if (dtype == DY_BF16) hipLaunchKernelGGL((head_grad_split_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, g, ld_g, (long long)rows, nb, nc, ncp, scale, (bf16_t*)dzb, ld_b, (bf16_t*)dzc, ld_c);
  else if (dtype == DY_F16) hipLaunchKernelGGL((head_grad_split_kernel<f16_t>), dim3(grid), dim3(256), 0, st, g, ld_g, (long long)rows, nb, nc, ncp, scale, (f16_t*)dzb, ld_b, (f16_t*)dzc, ld_c);
  else hipLaunchKernelGGL((head_grad_split_kernel<float>), dim3(grid), dim3(256), 0, st, g, ld_g, (long long)rows, nb, nc, ncp, scale, (float*)dzb, ld_b, (float*)dzc, ld_c);
  return check_launch("dy_head_grad_split");
}

extern "C" int32_t dy_add_dilated2_nhwc(const void* t, void* dx, int32_t n, int32_t h, int32_t w, int32_t H2, int32_t W2, int32_t c, int32_t ld_t, int32_t ld_dx,
                                        int32_t dtype, dy_stream_t stream) {
  const int es = dtype_size_no_fp8(dtype);
  DY_REQUIRE(es && t && dx && n > 0 && h > 0 && w > 0 && c > 0 && H2 >= 2 * h - 1 && W2 >= 2 * w - 1, DY_ERR_INVALID_ARG, "dy_add_dilated2_nhwc: bad arguments");
  const int epc = 16 / es;
  DY_REQUIRE(c % epc == 0 && DY_VIEW_OK(t, ld_t, c, es) && DY_VIEW_OK(dx, ld_dx, c, es), DY_ERR_INVALID_ARG, "dy_add_dilated2_nhwc: views must be whole 16-byte chunks");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned grid = grid1((long long)n * h * w * (c / epc));
  if (dtype == DY_BF16) hipLaunchKernelGGL((add_dilated2_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)t, (bf16_t*)dx, n, h, w, H2, W2, c / epc, ld_t, ld_dx);
  else if (dtype == DY_F16) hipLaunchKernelGGL((add_dilated2_kernel<f16_t>), dim3(grid), dim3(256), 0, st, (const f16_t*)t, (f16_t*)dx, n, h, w, H2, W2, c / epc, ld_t, ld_dx);
  else hipLaunchKernelGGL((add_dilated2_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)t, (float*)dx, n, h, w, H2, W2, c / epc, ld_t, ld_dx);
  return check_launch("dy_add_dilated2_nhwc");
}

extern "C" int32_t dy_add_nhwc(const void* a, const void* b, void* out, int64_t rows, int32_t c, int32_t ld_a, int32_t ld_b, int32_t ld_o, int32_t dtype, dy_stream_t stream) {
  const int es = dtype_size_no_fp8(dtype);
  DY_REQUIRE(es && a && b && out && rows > 0 && c > 0, DY_ERR_INVALID_ARG, "dy_add_nhwc: bad arguments");
  const int epc = 16 / es;
  DY_REQUIRE(c % epc == 0 && DY_VIEW_OK(a, ld_a, c, es) && DY_VIEW_OK(b, ld_b, c, es) && DY_VIEW_OK(out, ld_o, c, es), DY_ERR_INVALID_ARG,
             "dy_add_nhwc: views must be whole 16-byte chunks");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int cch = c / epc;
  const unsigned grid = grid1(rows * cch);
  if (dtype == DY_BF16) hipLaunchKernelGGL((add_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, (long long)rows, cch, ld_a, ld_b, ld_o);
  else if (dtype == DY_F16) hipLaunchKernelGGL((add_kernel<f16_t>), dim3(grid), dim3(256), 0, st, (const f16_t*)a, (const f16_t*)b, (f16_t*)out, (long long)rows, cch, ld_a, ld_b, ld_o);
  else hipLaunchKernelGGL((add_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, (long long)rows, cch, ld_a, ld_b, ld_o);
  return check_launch("dy_add_nhwc");
}


extern "C" int32_t dy_conv2d_grouped_bwd_nhwc(const dy_conv_desc* d, const void* dz, int32_t ld_dz, const float* w_oihw, float* dw_oihw, void* dx, int32_t ld_dx,
                                              const void* dx_accumulate, int32_t ld_acc, dy_stream_t stream) {
  DY_REQUIRE(d && dz && d->x && (dw_oihw || (dx && w_oihw)), DY_ERR_INVALID_ARG, "dy_conv2d_grouped_bwd_nhwc: null argument");
  const int es = dtype_size_no_fp8(d->dtype);
  DY_REQUIRE(es && d->groups > 1 && d->cin % d->groups == 0 && d->cout % d->groups == 0, DY_ERR_INVALID_ARG, "dy_conv2d_grouped_bwd_nhwc: bad dtype / groups");
  DY_REQUIRE(d->batch > 0 && d->h > 0 && d->w_in > 0 && d->ho == (d->h + 2 * d->pad - d->ksize) / d->stride + 1 && d->wo == (d->w_in + 2 * d->pad - d->ksize) / d->stride + 1,
             DY_ERR_INVALID_ARG, "dy_conv2d_grouped_bwd_nhwc: geometry");
  DY_REQUIRE(d->ld_x >= d->cin && ld_dz >= d->cout && (d->stride == 1 || d->stride == 2) && d->ksize >= 1 && d->ksize <= 7, DY_ERR_INVALID_ARG,
             "dy_conv2d_grouped_bwd_nhwc: pitches / stride / kernel size");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int cpg_in = d->cin / d->groups;
  if (dw_oihw) {
    DY_REQUIRE((cpg_in == 1 || cpg_in == 2 || cpg_in == 4) && d->ksize == 3, DY_ERR_UNSUPPORTED,
               "dy_conv2d_grouped_bwd_nhwc: weight gradient is built for 3x3 kernels with 1, 2 or 4 input channels per group (got k %d, %d)", d->ksize, cpg_in);
    const long long m_total = (long long)d->batch * d->ho * d->wo;
    const int slab = 128;
    dim3 grid((unsigned)((m_total + 4LL * slab - 1) / (4LL * slab)), (unsigned)((d->cout + 63) / 64));
#define DY_GW(T, CI)                                                                                                                                      \
  hipLaunchKernelGGL((conv_grouped_wgrad_kernel<T, CI, 3>), grid, dim3(256), 0, st, (const T*)d->x, (const T*)dz, dw_oihw, d->batch, d->h, d->w_in, d->cin, d->ld_x, d->ho, \
                     d->wo, d->cout, ld_dz, d->stride, d->pad, d->groups, slab)
#define DY_GWT(T)                        \
  do {                                   \
    if (cpg_in == 1) DY_GW(T, 1);        \
    else if (cpg_in == 2) DY_GW(T, 2);   \
    else DY_GW(T, 4);                    \
  } while (0)
    if (d->dtype == DY_BF16) DY_GWT(bf16_t);
    else if (d->dtype == DY_F16) DY_GWT(f16_t);
    else DY_GWT(float);
#undef DY_GWT
#undef DY_GW
    const int rc = check_launch("conv_grouped_wgrad_kernel");
    if (rc) return rc;
  }
  if (dx) {
    DY_REQUIRE(w_oihw && ld_dx >= d->cin && (!dx_accumulate || ld_acc >= d->cin), DY_ERR_INVALID_ARG, "dy_conv2d_grouped_bwd_nhwc: dx arguments");
    const unsigned grid = grid1((long long)d->batch * d->h * d->w_in * d->cin);
#define DY_GD(T)                                                                                                                                              \
  hipLaunchKernelGGL((conv_grouped_dgrad_kernel<T>), dim3(grid), dim3(256), 0, st, (const T*)dz, w_oihw, (const T*)dx_accumulate, (T*)dx, d->batch, d->h, d->w_in, d->cin, ld_dx, \
                     ld_acc, d->ho, d->wo, d->cout, ld_dz, d->ksize, d->stride, d->pad, d->groups)
    if (d->dtype == DY_BF16) DY_GD(bf16_t);
    else if (d->dtype == DY_F16) DY_GD(f16_t);
    else DY_GD(float);
#undef DY_GD
    return check_launch("conv_grouped_dgrad_kernel");
  }
  return DY_OK;
}

extern "C" int32_t dy_sumsq_f32(const float* g, int64_t n, double* out, dy_stream_t stream) {
  DY_REQUIRE(g && out && n > 0, DY_ERR_INVALID_ARG, "dy_sumsq_f32: bad arguments");
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid1(n) > 256 ? 256 : grid1(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, (long long)n, out);
  return check_launch("dy_sumsq_f32");
}

extern "C" int32_t dy_sgd_step(float* p, const float* grad, float* buf, int64_t n, float lr, float momentum, float weight_decay, int32_t nesterov, int32_t first_step,
                               const double* grad_sumsq, float max_norm, const float* amp_state, dy_stream_t stream) {
  DY_REQUIRE(p && grad && buf && n > 0, DY_ERR_INVALID_ARG, "dy_sgd_step: bad arguments");
  DY_REQUIRE(!amp_state || grad_sumsq, DY_ERR_INVALID_ARG, "dy_sgd_step: amp_state needs grad_sumsq (the overflow check reads it)");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid1(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, grad, buf, (long long)n, lr, momentum, weight_decay, nesterov,
                     first_step, grad_sumsq, max_norm, amp_state);
  return check_launch("dy_sgd_step");
}

extern "C" int32_t dy_adamw_step(float* p, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                 int32_t step, const double* grad_sumsq, float max_norm, const float* amp_state, dy_stream_t stream) {
  DY_REQUIRE(p && grad && m && v && n > 0 && step >= 1, DY_ERR_INVALID_ARG, "dy_adamw_step: bad arguments");
  DY_REQUIRE(!amp_state || grad_sumsq, DY_ERR_INVALID_ARG, "dy_adamw_step: amp_state needs grad_sumsq (the overflow check reads it)");
  hipLaunchKernelGGL(adamw_kernel, dim3(grid1(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, grad, m, v, (long long)n, lr, beta1, beta2, eps, weight_decay,
                     step, grad_sumsq, max_norm, amp_state);
  return check_launch("dy_adamw_step");
}

extern "C" int32_t dy_amp_update(float* amp_state, const double* grad_sumsq, float growth_factor, float backoff_factor, int32_t growth_interval, dy_stream_t stream) {
  DY_REQUIRE(amp_state && grad_sumsq && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval >= 1, DY_ERR_INVALID_ARG,
             "dy_amp_update: bad arguments");
  hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), amp_state, grad_sumsq, growth_factor, backoff_factor, growth_interval);
  return check_launch("dy_amp_update");
}

// grad += sink (sink -> 0), weight blocks transposed from the weight-gradient kernels' (cout, k, k, cin) to torch's (cout, cin, k, k).
// blockIdx.y = parameter, blockIdx.x strides over its elements in the DESTINATION order (coalesced read-modify-write of grad; the
// gather from sink has stride cin and stays in L2: the whole sink is 44 MB for scale s).
__global__ __launch_bounds__(256) void grad_sink_flush_kernel(const long long* __restrict__ entries, float* __restrict__ grad, float* __restrict__ sink) {
  const long long off = entries[4 * blockIdx.y], cout = entries[4 * blockIdx.y + 1], cin = entries[4 * blockIdx.y + 2], kk = entries[4 * blockIdx.y + 3];
  const long long n = cout * cin * (kk > 1 ? kk : 1);
  for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < n; j += (long long)gridDim.x * 256) {
    long long src = j;
    if (kk > 1) {
      const long long co = j / (cin * kk), r = j - co * cin * kk;
      const long long ci = r / kk, t = r - ci * kk;
      src = (co * kk + t) * cin + ci;
    }
    const float v = sink[off + src];
    if (v != 0.f) {
      grad[off + j] += v;
      sink[off + src] = 0.f;
    }
  }
}

extern "C" int32_t dy_grad_sink_flush(const int64_t* entries, int32_t n_entries, float* grad, float* sink, dy_stream_t stream) {
  DY_REQUIRE(entries && grad && sink && n_entries > 0, DY_ERR_INVALID_ARG, "dy_grad_sink_flush: bad arguments");
  hipLaunchKernelGGL(grad_sink_flush_kernel, dim3(32, (unsigned)n_entries), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const long long*>(entries), grad, sink);
  return check_launch("dy_grad_sink_flush");
}

extern "C" int32_t dy_ema_update(float* ema, const float* p, int64_t n, float decay, dy_stream_t stream) {
  DY_REQUIRE(ema && p && n > 0, DY_ERR_INVALID_ARG, "dy_ema_update: bad arguments");
  hipLaunchKernelGGL(ema_kernel, dim3(grid1(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ema, p, (long long)n, decay);
  return check_launch("dy_ema_update");
}
