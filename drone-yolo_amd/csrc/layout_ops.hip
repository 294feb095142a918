// Bandwidth-bound layout kernels: input layout conversion, 2x nearest upsample, strided
// NHWC copy and the SPPF triple max-pool.  All move 16-byte chunks per lane (guide G13).
#include "common_hip.h"

namespace dy {

// ---- fp32 NCHW -> NHWC(T), zero-padded channels ---------------------------------------------
// One thread per pixel: plane reads are coalesced across lanes, the write is one or
// more 16-byte chunks per lane (c_pad * sizeof(T) bytes, contiguous across lanes when
// ld_dst == c_pad).  Reference: engine/predictor.py:118-136 (dtype cast of a tensor source).
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int n,
                                                           int c, int hw, int c_pad, int ld) {
  constexpr int EPC = Elem<T>::EPC;
  const long long total = (long long)n * hw;
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (long long)gridDim.x * 256) {
    const int img = (int)(pix / hw);
    const int p = (int)(pix - (long long)img * hw);
    const float* s = src + (size_t)img * c * hw + p;
    T* d = dst + (size_t)pix * ld;
    for (int c0 = 0; c0 < c_pad; c0 += EPC) {
      float f[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) f[e] = (c0 + e < c) ? s[(size_t)(c0 + e) * hw] : 0.f;
      *reinterpret_cast<u32x4*>(d + c0) = Chunk<T>::pack(f);
    }
  }
}

// ---- DY_F16X2 forms of the two layout casts: 8 channels = a (hi, lo) chunk pair (include/dyolo.h) -----------------------------------
__global__ __launch_bounds__(256) void nchw_to_nhwc_split_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, int n, int c, int hw, int c_pad, int ld) {
  const long long total = (long long)n * hw;
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (long long)gridDim.x * 256) {
    const int img = (int)(pix / hw);
    const int p = (int)(pix - (long long)img * hw);
    const float* s = src + (size_t)img * c * hw + p;
    unsigned char* d = dst + (size_t)pix * ld * 4;
    for (int c0 = 0; c0 < c_pad; c0 += 8) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = (c0 + e < c) ? s[(size_t)(c0 + e) * hw] : 0.f;
      u32x4 hi, lo;
      split8(f, hi, lo);
      *reinterpret_cast<u32x4*>(d + c0 * 4) = hi;
      *reinterpret_cast<u32x4*>(d + c0 * 4 + 16) = lo;
    }
  }
}

__global__ __launch_bounds__(256) void nhwc_split_to_nchw_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int n, int c, int hw, int ld) {
  const long long total = (long long)n * c * hw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % hw);
    const long long t = i / hw;
    const int ch = (int)(t % c);
    const int img = (int)(t / c);
    const f16_t* g = reinterpret_cast<const f16_t*>(src + (((size_t)img * hw + p) * ld + (ch & ~7)) * 4);
    dst[i] = (float)g[ch & 7] + (float)g[8 + (ch & 7)] * kSplitInv;
  }
}

// ---- uint8 NCHW -> NHWC(T) with x * scale, zero-padded channels: DetectionTrainer.preprocess_batch's
// `img.float() / 255` (models/yolo/detect/train.py:57-60) fused with the layout step of the training forward ----------
template <typename T>
__global__ __launch_bounds__(256) void nchw_u8_to_nhwc_kernel(const uint8_t* __restrict__ src, T* __restrict__ dst, int n, int c, int hw, int c_pad, int ld,
                                                              float scale) {
  constexpr int EPC = Elem<T>::EPC;
  const long long total = (long long)n * hw;
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (long long)gridDim.x * 256) {
    const int img = (int)(pix / hw);
    const int p = (int)(pix - (long long)img * hw);
    const uint8_t* s = src + (size_t)img * c * hw + p;
    T* d = dst + (size_t)pix * ld;
    for (int c0 = 0; c0 < c_pad; c0 += EPC) {
      float f[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) f[e] = (c0 + e < c) ? (float)s[(size_t)(c0 + e) * hw] / scale : 0.f;
      *reinterpret_cast<u32x4*>(d + c0) = Chunk<T>::pack(f);
    }
  }
}

// ---- NHWC(T) -> fp32 NCHW ---------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int n,
                                                           int c, int hw, int ld) {
  const long long total = (long long)n * c * hw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % hw);
    const long long t = i / hw;
    const int ch = (int)(t % c);
    const int img = (int)(t / c);
    dst[i] = Elem<T>::to_f32(src[((size_t)img * hw + p) * ld + ch]);
  }
}

// ---- 2x nearest upsample / strided copy (chunk granular) -------------------------------------
template <bool UP>
__global__ __launch_bounds__(256) void copy_chunks_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int n,
                                                          int ho, int wo, int cchunks, int lds_chunks,
                                                          int ldd_chunks) {
  // src dims are (ho/2, wo/2) when UP else (ho, wo); pitches are in 16-byte chunks.
  const long long total = (long long)n * ho * wo * cchunks;
  const int hs = UP ? ho / 2 : ho, ws = UP ? wo / 2 : wo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % cchunks);
    long long t = i / cchunks;
    const int x = (int)(t % wo);
    t /= wo;
    const int y = (int)(t % ho);
    const int img = (int)(t / ho);
    const int ys = UP ? (y >> 1) : y, xs = UP ? (x >> 1) : x;
    const size_t so = ((size_t)(img * hs + ys) * ws + xs) * (size_t)lds_chunks + cc;
    const size_t d_o = ((size_t)(img * ho + y) * wo + x) * (size_t)ldd_chunks + cc;
    dst[d_o] = src[so];
  }
}

// ---- SPPF: three chained k x k stride-1 max pools (block.py:185-191) ----------------------------
// One workgroup per (image, 16-byte channel chunk): the whole h x w map of that chunk
// sits in LDS; each pass is a separable row-max then column-max over a (2*(k/2)+1) window
// clipped at the border (MaxPool2d pads with -inf, i.e. ignores out-of-range taps).
template <typename T>
__device__ __forceinline__ u32x4 chunk_max(u32x4 a, u32x4 b) {
  constexpr int EPC = Elem<T>::EPC;
  float fa[EPC], fb[EPC];
  Chunk<T>::unpack(a, fa);
  Chunk<T>::unpack(b, fb);
#pragma unroll
  for (int e = 0; e < EPC; ++e) fa[e] = fmaxf(fa[e], fb[e]);
  return Chunk<T>::pack(fa);
}

// One workgroup = one image x G consecutive 16-byte channel chunks (G*16 contiguous bytes per pixel: whole 128-byte lines at G = 8;
// the round-2 kernel took ONE chunk per workgroup, so every line of x was fetched by eight workgroups: 0.42 GB read for a 52 MB
// input by the counters).  Two LDS images [pixel][G chunks]: row maxima into tmp, column maxima back into cur (nobody reads cur in
// that phase) and out to y_pass.
template <typename T>
__global__ __launch_bounds__(512) void sppf_maxpool3_kernel(const T* __restrict__ x, T* __restrict__ y1,
                                                            T* __restrict__ y2, T* __restrict__ y3, int h, int w,
                                                            int cgroups, int G, int ld, int r) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  constexpr int EPC = Elem<T>::EPC;
  const int hw = h * w, tot = hw * G;
  u32x4* cur = reinterpret_cast<u32x4*>(dyn_smem);
  u32x4* tmp = cur + tot;
  const int img = blockIdx.x / cgroups;
  const int cg = blockIdx.x - img * cgroups;
  const size_t base = (size_t)img * hw * ld + (size_t)cg * G * EPC;
  for (int i = threadIdx.x; i < tot; i += 512) {
    const int p = i / G, g = i - p * G;
    cur[i] = *reinterpret_cast<const u32x4*>(x + base + (size_t)p * ld + g * EPC);
  }
  __syncthreads();
  T* outs[3] = {y1, y2, y3};
#pragma unroll 1
  for (int pass = 0; pass < 3; ++pass) {
    for (int i = threadIdx.x; i < tot; i += 512) {
      const int p = i / G, g = i - p * G;
      const int yy = p / w, xx = p - yy * w;
      const int x0 = xx - r < 0 ? 0 : xx - r, x1 = xx + r >= w ? w - 1 : xx + r;
      u32x4 m = cur[(yy * w + x0) * G + g];
      for (int q = x0 + 1; q <= x1; ++q) m = chunk_max<T>(m, cur[(yy * w + q) * G + g]);
      tmp[i] = m;
    }
    __syncthreads();
    T* o = outs[pass];
    for (int i = threadIdx.x; i < tot; i += 512) {
      const int p = i / G, g = i - p * G;
      const int yy = p / w, xx = p - yy * w;
      const int y0 = yy - r < 0 ? 0 : yy - r, y1e = yy + r >= h ? h - 1 : yy + r;
      u32x4 m = tmp[(y0 * w + xx) * G + g];
      for (int q = y0 + 1; q <= y1e; ++q) m = chunk_max<T>(m, tmp[(q * w + xx) * G + g]);
      cur[i] = m;
      *reinterpret_cast<u32x4*>(o + base + (size_t)p * ld + g * EPC) = m;
    }
    __syncthreads();
  }
}

// DY_F16X2: the same pools on (hi, lo) chunk PAIRS (G counts pairs, 32 bytes each).  hi = rn_f16(x) is monotonic in x and lo orders the
// values that share a hi, so the maximum is a lexicographic choice between whole pairs — taken here by comparing the joined fp32
// values (exact: 22 significant bits) and selecting the pair, never re-splitting: the pooled element IS one of the inputs, bit for bit.
struct ChunkPair {
  u32x4 hi, lo;
};
__device__ __forceinline__ ChunkPair pair_max(const ChunkPair a, const ChunkPair b) {
  float fa[8], fb[8];
  join8(a.hi, a.lo, fa);
  join8(b.hi, b.lo, fb);
  const f16x8 ah = __builtin_bit_cast(f16x8, a.hi), al = __builtin_bit_cast(f16x8, a.lo), bh = __builtin_bit_cast(f16x8, b.hi), bl = __builtin_bit_cast(f16x8, b.lo);
  f16x8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool tb = fb[e] > fa[e];
    h[e] = tb ? bh[e] : ah[e];
    l[e] = tb ? bl[e] : al[e];
  }
  return ChunkPair{__builtin_bit_cast(u32x4, h), __builtin_bit_cast(u32x4, l)};
}

__global__ __launch_bounds__(512) void sppf_maxpool3_split_kernel(const unsigned char* __restrict__ x, unsigned char* __restrict__ y1, unsigned char* __restrict__ y2,
                                                                  unsigned char* __restrict__ y3, int h, int w, int cgroups, int G, int ld, int r) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  const int hw = h * w, tot = hw * G;
  ChunkPair* cur = reinterpret_cast<ChunkPair*>(dyn_smem);
  ChunkPair* tmp = cur + tot;
  const int img = blockIdx.x / cgroups;
  const int cg = blockIdx.x - img * cgroups;
  const size_t base = ((size_t)img * hw * ld + (size_t)cg * G * 8) * 4;
  for (int i = threadIdx.x; i < tot; i += 512) {
    const int p = i / G, g = i - p * G;
    const unsigned char* s = x + base + ((size_t)p * ld + g * 8) * 4;
    cur[i] = ChunkPair{*reinterpret_cast<const u32x4*>(s), *reinterpret_cast<const u32x4*>(s + 16)};
  }
  __syncthreads();
  unsigned char* outs[3] = {y1, y2, y3};
#pragma unroll 1
  for (int pass = 0; pass < 3; ++pass) {
    for (int i = threadIdx.x; i < tot; i += 512) {
      const int p = i / G, g = i - p * G;
      const int yy = p / w, xx = p - yy * w;
      const int x0 = xx - r < 0 ? 0 : xx - r, x1 = xx + r >= w ? w - 1 : xx + r;
      ChunkPair m = cur[(yy * w + x0) * G + g];
      for (int q = x0 + 1; q <= x1; ++q) m = pair_max(m, cur[(yy * w + q) * G + g]);
      tmp[i] = m;
    }
    __syncthreads();
    unsigned char* o = outs[pass];
    for (int i = threadIdx.x; i < tot; i += 512) {
      const int p = i / G, g = i - p * G;
      const int yy = p / w, xx = p - yy * w;
      const int y0 = yy - r < 0 ? 0 : yy - r, y1e = yy + r >= h ? h - 1 : yy + r;
      ChunkPair m = tmp[(y0 * w + xx) * G + g];
      for (int q = y0 + 1; q <= y1e; ++q) m = pair_max(m, tmp[(q * w + xx) * G + g]);
      cur[i] = m;
      unsigned char* d = o + base + ((size_t)p * ld + g * 8) * 4;
      *reinterpret_cast<u32x4*>(d) = m.hi;
      *reinterpret_cast<u32x4*>(d + 16) = m.lo;
    }
    __syncthreads();
  }
}

// ---- 16-bit / fp32 NHWC -> fp8 (e4m3fn) NHWC, q = sat(x / act_scale): hand-over from the fp16 image stem to the fp8 layers ----
template <typename T>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const T* __restrict__ src, fp8_t* __restrict__ dst, long long rows, int c16, int lds, int ldd, float inv) {
  constexpr int EPC = Elem<T>::EPC;  // source elements per 16-byte chunk; one destination chunk = 16 elements
  const long long total = rows * c16;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % c16);
    const long long r = i / c16;
    float f[16];
#pragma unroll
    for (int k = 0; k < 16 / EPC; ++k) {
      float t[EPC];
      Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(src + (size_t)r * lds + cc * 16 + k * EPC), t);
#pragma unroll
      for (int e = 0; e < EPC; ++e) f[k * EPC + e] = t[e] * inv;
    }
    *reinterpret_cast<u32x4*>(dst + (size_t)r * ldd + cc * 16) = Chunk<fp8_t>::pack(f);
  }
}


// ---- weight packing on the device (training: the fp32 master weights are re-packed every step) ------------------------------
// One thread per DESTINATION element of the packed buffer: it works out which logical weight W[co][ci][r][q] the element holds in
// the requested layout (include/dyolo.h: DY_WLAYOUT_ROWS / HALO3X3 / FRAG1X1; zero in the padding) and fetches it from the master
// tensor through its strides.  `tf` = transpose + flip: W[co][ci][r][q] = w[ci][co][k-1-r][k-1-q], the weights of the convolution
// that computes the input gradient.  Replaces the pad / permute / flip / cast chain of torch kernels (4-6 launches per
// convolution and step) by one launch.
struct PackArgs {
  const float* w;
  long long s_co, s_ci, s_r, s_q;  // element strides of the SOURCE tensor's (cout, cin, r, q) axes
  int lco, lci_valid, lci, k, tf;  // logical (cout, cin) of the packed convolution: lci_valid channels exist, the layout is lci wide
  int layout, e, a, b, nf;         // ROWS: a = k_pad, b = cout_pad; HALO3X3 / FRAG1X1: a = chunks per (co-tile), nf = BN / 16
  long long total;
};

// the logical weight destination element i of the packed buffer holds (0 in the padding)
__device__ __forceinline__ float pack_element(const PackArgs& p, long long i) {
  const int lco = p.lco, lci = p.lci;
  int co, ci, r, q;
  if (p.layout == DY_WLAYOUT_ROWS) {
    co = (int)(i / p.a);
    const int kk = (int)(i - (long long)co * p.a);
    const int tap = kk / lci;
    ci = kk - tap * lci;
    r = tap / p.k, q = tap - r * p.k;
    if (tap >= p.k * p.k) co = 1 << 30;  // row padding
  } else {  // fragment orders: (((nt * nch + c) * taps + tap) * NF + j) * 64 + lq * 16 + lr) * E + e
    const int taps = p.layout == DY_WLAYOUT_HALO3X3 ? 9 : 1;
    long long t = i;
    const int e = (int)(t % p.e);
    t /= p.e;
    const int lr = (int)(t % 16);
    t /= 16;
    const int lq = (int)(t % 4);
    t /= 4;
    const int j = (int)(t % p.nf);
    t /= p.nf;
    const int tap = (int)(t % taps);
    t /= taps;
    const int c = (int)(t % p.a);
    const int nt = (int)(t / p.a);
    co = (nt * p.nf + j) * 16 + lr;
    ci = c * 4 * p.e + lq * p.e + e;
    r = tap / 3, q = tap - r * 3;
    if (taps == 1) r = q = 0;
  }
  float v = 0.f;
  if (co < lco && ci < p.lci_valid) {
    const int sr = p.tf ? p.k - 1 - r : r, sq = p.tf ? p.k - 1 - q : q;
    const int sco = p.tf ? ci : co, sci = p.tf ? co : ci;
    v = p.w[sco * p.s_co + sci * p.s_ci + sr * p.s_r + sq * p.s_q];
  }
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weights_kernel(const PackArgs p, T* __restrict__ dst) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) dst[i] = Elem<T>::from_f32(pack_element(p, i));
}

// Every convolution of a training step in ONE launch (dy_pack_conv_weights_batched): a device table of jobs; a workgroup finds its
// job by its first-block number (the table is sorted by it) and walks the job's elements with the job's own block count.
struct PackEntry {
  PackArgs a;
  void* dst;
  int block0, nblocks;
};
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const PackEntry* __restrict__ tab, int n) {
  int lo = 0, hi = n - 1;  // last entry with block0 <= blockIdx.x (uniform: scalar loads)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].block0 <= (int)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const PackArgs p = tab[lo].a;
  T* __restrict__ dst = reinterpret_cast<T*>(tab[lo].dst);
  const long long stride = (long long)tab[lo].nblocks * 256;
  for (long long i = (long long)((int)blockIdx.x - tab[lo].block0) * 256 + threadIdx.x; i < p.total; i += stride) dst[i] = Elem<T>::from_f32(pack_element(p, i));
}

static inline int grid_for(long long items) {
  long long b = (items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 * 4 ? 2048 * 4 : b));
}

}  // namespace dy

using namespace dy;

extern "C" int32_t dy_nchw_f32_to_nhwc(const float* src, void* dst, int32_t n, int32_t c, int32_t h, int32_t w,
                                       int32_t c_pad, int32_t ld_dst, int32_t dtype, dy_stream_t stream) {
  const int es = dtype == DY_F16X2 ? 4 : dtype_size_no_fp8(dtype);
  DY_REQUIRE(src && dst && es, DY_ERR_INVALID_ARG, "dy_nchw_f32_to_nhwc: null pointer or bad dtype");
  DY_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, DY_ERR_INVALID_ARG, "dy_nchw_f32_to_nhwc: bad dims");
  const int epc = dtype == DY_F16X2 ? 8 : 16 / es;  // (split pairs: whole groups of 8 channels)
  if (dtype == DY_F16X2) {
    DY_REQUIRE(c_pad >= c && c_pad % 8 == 0 && ld_dst >= c_pad && ld_dst % 4 == 0 && aligned16(dst), DY_ERR_INVALID_ARG,
               "dy_nchw_f32_to_nhwc: DY_F16X2 needs c_pad in groups of 8 channels, dst 16B aligned");
    hipLaunchKernelGGL(nchw_to_nhwc_split_kernel, dim3(grid_for((long long)n * h * w)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, (unsigned char*)dst, n, c,
                       h * w, c_pad, ld_dst);
    return check_launch("nchw_to_nhwc_split_kernel");
  }
  DY_REQUIRE(c_pad >= c && c_pad % epc == 0 && ld_dst >= c_pad && (ld_dst * es) % 16 == 0 && aligned16(dst),
             DY_ERR_INVALID_ARG, "dy_nchw_f32_to_nhwc: c_pad/ld_dst must be multiples of %d elements, dst 16B aligned", epc);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = grid_for((long long)n * h * w);
  if (dtype == DY_BF16)
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, src, (bf16_t*)dst, n, c, h * w, c_pad, ld_dst);
  else if (dtype == DY_F16)
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<f16_t>), dim3(grid), dim3(256), 0, st, src, (f16_t*)dst, n, c, h * w, c_pad, ld_dst);
  else
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<float>), dim3(grid), dim3(256), 0, st, src, (float*)dst, n, c, h * w, c_pad, ld_dst);
  return check_launch("nchw_to_nhwc_kernel");
}

extern "C" int32_t dy_nchw_u8_to_nhwc(const uint8_t* src, void* dst, int32_t n, int32_t c, int32_t h, int32_t w, int32_t c_pad, int32_t ld_dst,
                                      float divisor, int32_t dtype, dy_stream_t stream) {
  const int es = dtype_size_no_fp8(dtype);
  DY_REQUIRE(src && dst && es && divisor != 0.f, DY_ERR_INVALID_ARG, "dy_nchw_u8_to_nhwc: null pointer, bad dtype or zero divisor");
  DY_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, DY_ERR_INVALID_ARG, "dy_nchw_u8_to_nhwc: bad dims");
  const int epc = 16 / es;
  DY_REQUIRE(c_pad >= c && c_pad % epc == 0 && ld_dst >= c_pad && (ld_dst * es) % 16 == 0 && aligned16(dst), DY_ERR_INVALID_ARG,
             "dy_nchw_u8_to_nhwc: c_pad/ld_dst must be multiples of %d elements, dst 16B aligned", epc);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = grid_for((long long)n * h * w);
  if (dtype == DY_BF16)
    hipLaunchKernelGGL((nchw_u8_to_nhwc_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, src, (bf16_t*)dst, n, c, h * w, c_pad, ld_dst, divisor);
  else if (dtype == DY_F16)
    hipLaunchKernelGGL((nchw_u8_to_nhwc_kernel<f16_t>), dim3(grid), dim3(256), 0, st, src, (f16_t*)dst, n, c, h * w, c_pad, ld_dst, divisor);
  else
    hipLaunchKernelGGL((nchw_u8_to_nhwc_kernel<float>), dim3(grid), dim3(256), 0, st, src, (float*)dst, n, c, h * w, c_pad, ld_dst, divisor);
  return check_launch("nchw_u8_to_nhwc_kernel");
}

extern "C" int32_t dy_nhwc_to_nchw_f32(const void* src, float* dst, int32_t n, int32_t c, int32_t h, int32_t w,
                                       int32_t ld_src, int32_t src_dtype, dy_stream_t stream) {
  const int es = src_dtype == DY_F16X2 ? 4 : dtype_size_no_fp8(src_dtype);
  DY_REQUIRE(src && dst && es, DY_ERR_INVALID_ARG, "dy_nhwc_to_nchw_f32: null pointer or bad dtype");
  DY_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && ld_src >= c, DY_ERR_INVALID_ARG, "dy_nhwc_to_nchw_f32: bad dims");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = grid_for((long long)n * c * h * w);
  if (src_dtype == DY_F16X2) {  // (a view that starts at a multiple of 8 channels of its buffer; c itself may be anything)
    DY_REQUIRE(aligned16(src) && ld_src % 4 == 0, DY_ERR_INVALID_ARG, "dy_nhwc_to_nchw_f32: DY_F16X2 view must be 16-byte aligned");
    hipLaunchKernelGGL(nhwc_split_to_nchw_kernel, dim3(grid), dim3(256), 0, st, (const unsigned char*)src, dst, n, c, h * w, ld_src);
    return check_launch("nhwc_split_to_nchw_kernel");
  }
  if (src_dtype == DY_BF16)
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, dst, n, c, h * w, ld_src);
  else if (src_dtype == DY_F16)
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<f16_t>), dim3(grid), dim3(256), 0, st, (const f16_t*)src, dst, n, c, h * w, ld_src);
  else
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)src, dst, n, c, h * w, ld_src);
  return check_launch("nhwc_to_nchw_kernel");
}

static int32_t copy_common(bool up, const void* src, void* dst, int32_t n, int32_t h, int32_t w, int32_t c,
                           int32_t ld_src, int32_t ld_dst, int32_t dtype, dy_stream_t stream, const char* who) {
  const int es = dy_dtype_size(dtype);
  DY_REQUIRE(src && dst && es, DY_ERR_INVALID_ARG, "%s: null pointer or bad dtype", who);
  const int epc = 16 / es;
  DY_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % epc == 0, DY_ERR_INVALID_ARG, "%s: c must be a multiple of %d", who, epc);
  DY_REQUIRE(ld_src >= c && ld_dst >= c && ld_src % epc == 0 && ld_dst % epc == 0 && aligned16(src) && aligned16(dst),
             DY_ERR_INVALID_ARG, "%s: views must be 16-byte aligned", who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int ho = up ? 2 * h : h, wo = up ? 2 * w : w;
  const int grid = grid_for((long long)n * ho * wo * (c / epc));
  if (up)
    hipLaunchKernelGGL((copy_chunks_kernel<true>), dim3(grid), dim3(256), 0, st, (const u32x4*)src, (u32x4*)dst, n, ho, wo,
                       c / epc, ld_src / epc, ld_dst / epc);
  else
    hipLaunchKernelGGL((copy_chunks_kernel<false>), dim3(grid), dim3(256), 0, st, (const u32x4*)src, (u32x4*)dst, n, ho, wo,
                       c / epc, ld_src / epc, ld_dst / epc);
  return check_launch(who);
}

extern "C" int32_t dy_upsample2x_nhwc(const void* src, void* dst, int32_t n, int32_t h, int32_t w, int32_t c,
                                      int32_t ld_src, int32_t ld_dst, int32_t dtype, dy_stream_t stream) {
  return copy_common(true, src, dst, n, h, w, c, ld_src, ld_dst, dtype, stream, "dy_upsample2x_nhwc");
}

extern "C" int32_t dy_copy_nhwc(const void* src, void* dst, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ld_src,
                                int32_t ld_dst, int32_t dtype, dy_stream_t stream) {
  return copy_common(false, src, dst, n, h, w, c, ld_src, ld_dst, dtype, stream, "dy_copy_nhwc");
}

extern "C" int32_t dy_sppf_maxpool3(const void* x, void* y1, void* y2, void* y3, int32_t n, int32_t h, int32_t w,
                                    int32_t c, int32_t ld, int32_t k, int32_t dtype, dy_stream_t stream) {
  const int es = dy_dtype_size(dtype);
  DY_REQUIRE(x && y1 && y2 && y3 && es, DY_ERR_INVALID_ARG, "dy_sppf_maxpool3: null pointer or bad dtype");
  const int epc = 16 / es;
  DY_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % epc == 0 && ld >= c && ld % epc == 0, DY_ERR_INVALID_ARG,
             "dy_sppf_maxpool3: c/ld must be multiples of %d", epc);
  DY_REQUIRE(k >= 1 && (k & 1), DY_ERR_INVALID_ARG, "dy_sppf_maxpool3: k must be odd");
  DY_REQUIRE(aligned16(x) && aligned16(y1) && aligned16(y2) && aligned16(y3), DY_ERR_INVALID_ARG,
             "dy_sppf_maxpool3: views must be 16-byte aligned");
  DY_REQUIRE((long long)h * w <= 4608, DY_ERR_UNSUPPORTED, "dy_sppf_maxpool3: h*w=%d exceeds the LDS-resident limit 4608", h * w);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int cchunks = c / epc;
  // chunks per workgroup: the largest divisor of the chunk count (at most 8 = a whole 128-byte line per pixel) whose two LDS images fit
  int G = 1;
  for (int g = 8; g >= 1; --g)
    if (cchunks % g == 0 && (size_t)h * w * 32 * g <= 147456) {
      G = g;
      break;
    }
  const size_t smem = (size_t)h * w * 32 * G;
  const int cgroups = cchunks / G;
  if (dtype == DY_F16X2) {
    DY_REQUIRE(c % 8 == 0, DY_ERR_INVALID_ARG, "dy_sppf_maxpool3: DY_F16X2 needs c in groups of 8 channels");
    const int pairs = c / 8;
    int Gp = 1;
    for (int g = 4; g >= 1; --g)
      if (pairs % g == 0 && (size_t)h * w * 64 * g <= 147456) {
        Gp = g;
        break;
      }
    const size_t sm = (size_t)h * w * 64 * Gp;
    DY_REQUIRE(sm <= 147456, DY_ERR_UNSUPPORTED, "dy_sppf_maxpool3: DY_F16X2 map of %d pixels exceeds the LDS-resident limit", h * w);
    if (sm > 48 * 1024) (void)hipFuncSetAttribute((const void*)sppf_maxpool3_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL(sppf_maxpool3_split_kernel, dim3((unsigned)(n * (pairs / Gp))), dim3(512), sm, st, (const unsigned char*)x, (unsigned char*)y1, (unsigned char*)y2,
                       (unsigned char*)y3, h, w, pairs / Gp, Gp, ld, k / 2);
    return check_launch("sppf_maxpool3_split_kernel");
  }
  const dim3 grid((unsigned)(n * cgroups));
#define DY_SPPF_LAUNCH(T)                                                                                          \
  do {                                                                                                              \
    if (smem > 48 * 1024)                                                                                           \
      (void)hipFuncSetAttribute((const void*)sppf_maxpool3_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                (int)smem);                                                                         \
    hipLaunchKernelGGL((sppf_maxpool3_kernel<T>), grid, dim3(512), smem, st, (const T*)x, (T*)y1, (T*)y2, (T*)y3, h, \
                       w, cgroups, G, ld, k / 2);                                                                   \
  } while (0)
  if (dtype == DY_BF16)
    DY_SPPF_LAUNCH(bf16_t);
  else if (dtype == DY_F16)
    DY_SPPF_LAUNCH(f16_t);
  else if (dtype == DY_FP8)  // max commutes with the (monotonic) quantisation: exact in the quantised domain
    DY_SPPF_LAUNCH(fp8_t);
  else
    DY_SPPF_LAUNCH(float);
#undef DY_SPPF_LAUNCH
  return check_launch("sppf_maxpool3_kernel");
}

extern "C" int32_t dy_quantize_fp8_nhwc(const void* src, void* dst, int64_t rows, int32_t c, int32_t ld_src, int32_t ld_dst, int32_t src_dtype, float act_scale,
                                        dy_stream_t stream) {
  const int es = dtype_size_no_fp8(src_dtype);
  DY_REQUIRE(src && dst && es && rows > 0 && c > 0 && act_scale > 0.f, DY_ERR_INVALID_ARG, "dy_quantize_fp8_nhwc: bad arguments");
  DY_REQUIRE(c % 16 == 0 && ld_src >= c && ld_dst >= c && (ld_src * es) % 16 == 0 && ld_dst % 16 == 0 && aligned16(src) && aligned16(dst), DY_ERR_INVALID_ARG,
             "dy_quantize_fp8_nhwc: c must be a multiple of 16 and both views whole 16-byte chunks");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = grid_for((long long)rows * (c / 16));
  const float inv = 1.f / act_scale;
  if (src_dtype == DY_BF16)
    hipLaunchKernelGGL((quantize_fp8_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, (fp8_t*)dst, (long long)rows, c / 16, ld_src, ld_dst, inv);
  else if (src_dtype == DY_F16)
    hipLaunchKernelGGL((quantize_fp8_kernel<f16_t>), dim3(grid), dim3(256), 0, st, (const f16_t*)src, (fp8_t*)dst, (long long)rows, c / 16, ld_src, ld_dst, inv);
  else
    hipLaunchKernelGGL((quantize_fp8_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)src, (fp8_t*)dst, (long long)rows, c / 16, ld_src, ld_dst, inv);
  return check_launch("quantize_fp8_kernel");
}

static int pack_args(const float* w, int64_t s_co, int64_t s_ci, int64_t s_r, int64_t s_q, int32_t cout, int32_t cin, int32_t ksize, int32_t transpose_flip,
                     int32_t cin_logical, const void* dst, int64_t dst_elems, int32_t dtype, int32_t w_layout, PackArgs& p) {
  const int es = dtype_size_no_fp8(dtype);
  DY_REQUIRE(w && dst && es && cout > 0 && cin > 0 && ksize >= 1 && dst_elems > 0 && aligned16(dst), DY_ERR_INVALID_ARG, "dy_pack_conv_weights: bad arguments");
  const int e = 16 / es;
  const int lco = transpose_flip ? cin : cout;
  int lci = transpose_flip ? cout : cin;
  if (cin_logical > 0) {  // the input view carries zero-padded channels (the 3-channel image padded to one chunk)
    DY_REQUIRE(!transpose_flip && cin_logical >= cin, DY_ERR_INVALID_ARG, "dy_pack_conv_weights: cin_logical");
    lci = cin_logical;
  }
  p = PackArgs{};
  p.w = w, p.s_co = s_co, p.s_ci = s_ci, p.s_r = s_r, p.s_q = s_q, p.k = ksize, p.tf = transpose_flip ? 1 : 0, p.layout = w_layout, p.e = e;
  p.lco = lco, p.lci_valid = transpose_flip ? cout : cin, p.lci = lci;
  long long need = 0;
  if (w_layout == DY_WLAYOUT_ROWS) {
    p.a = dy_conv_k_pad(lci, ksize, dtype), p.b = dy_conv_cout_pad(lco);
    need = (long long)p.b * p.a;
  } else if (w_layout == DY_WLAYOUT_HALO3X3) {
    DY_REQUIRE(ksize == 3, DY_ERR_INVALID_ARG, "dy_pack_conv_weights: DY_WLAYOUT_HALO3X3 is for 3x3 kernels");
    const int bn = lco > 32 ? 64 : 32, kc = 4 * e;
    p.nf = bn / 16, p.a = (lci + kc - 1) / kc;
    need = (long long)((lco + bn - 1) / bn) * p.a * 9 * p.nf * 64 * e;
  } else if (w_layout == DY_WLAYOUT_FRAG1X1) {
    DY_REQUIRE(ksize == 1, DY_ERR_INVALID_ARG, "dy_pack_conv_weights: DY_WLAYOUT_FRAG1X1 is for 1x1 kernels");
    const int bn = lco > 64 ? 128 : (lco > 16 ? 64 : 16), kc = 4 * e;
    p.nf = bn / 16, p.a = (lci + kc - 1) / kc;
    need = (long long)((lco + bn - 1) / bn) * p.a * p.nf * 64 * e;
  } else {
    DY_REQUIRE(false, DY_ERR_INVALID_ARG, "dy_pack_conv_weights: unknown w_layout %d", w_layout);
  }
  DY_REQUIRE(dst_elems == need, DY_ERR_INVALID_ARG, "dy_pack_conv_weights: dst holds %lld elements, the layout needs %lld", (long long)dst_elems, need);
  p.total = need;
  return 0;
}

extern "C" int32_t dy_pack_conv_weights(const float* w, int64_t s_co, int64_t s_ci, int64_t s_r, int64_t s_q, int32_t cout, int32_t cin, int32_t ksize,
                                        int32_t transpose_flip, int32_t cin_logical, void* dst, int64_t dst_elems, int32_t dtype, int32_t w_layout, dy_stream_t stream) {
  PackArgs p;
  if (const int rc = pack_args(w, s_co, s_ci, s_r, s_q, cout, cin, ksize, transpose_flip, cin_logical, dst, dst_elems, dtype, w_layout, p)) return rc;
  const long long need = p.total;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = grid_for(need);
  if (dtype == DY_BF16) hipLaunchKernelGGL((pack_weights_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, p, (bf16_t*)dst);
  else if (dtype == DY_F16) hipLaunchKernelGGL((pack_weights_kernel<f16_t>), dim3(grid), dim3(256), 0, st, p, (f16_t*)dst);
  else hipLaunchKernelGGL((pack_weights_kernel<float>), dim3(grid), dim3(256), 0, st, p, (float*)dst);
  return check_launch("pack_weights_kernel");
}

extern "C" int64_t dy_pack_conv_weights_table_bytes(int32_t n_jobs) { return n_jobs > 0 ? (int64_t)n_jobs * (int64_t)sizeof(PackEntry) : -1; }

extern "C" int32_t dy_pack_conv_weights_table(const dy_pack_job* jobs, int32_t n_jobs, int32_t dtype, void* table_host, int64_t table_bytes, int32_t* total_blocks) {
  DY_REQUIRE(jobs && n_jobs > 0 && table_host && total_blocks && table_bytes >= dy_pack_conv_weights_table_bytes(n_jobs), DY_ERR_INVALID_ARG,
             "dy_pack_conv_weights_table: null pointer or table too small");
  PackEntry* tab = reinterpret_cast<PackEntry*>(table_host);
  int blocks = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const dy_pack_job& j = jobs[i];
    PackEntry e{};
    if (const int rc = pack_args(j.w, j.s_co, j.s_ci, j.s_r, j.s_q, j.cout, j.cin, j.ksize, j.transpose_flip, j.cin_logical, j.dst, j.dst_elems, dtype, j.w_layout, e.a)) return rc;
    e.dst = j.dst;
    long long nb = (e.a.total + 2047) / 2048;  // eight elements per thread, at most 64 workgroups per job
    e.nblocks = (int)(nb < 1 ? 1 : (nb > 64 ? 64 : nb));
    e.block0 = blocks;
    blocks += e.nblocks;
    tab[i] = e;
  }
  *total_blocks = blocks;
  return 0;
}

extern "C" int32_t dy_pack_conv_weights_batched(const void* table_dev, int32_t n_jobs, int32_t total_blocks, int32_t dtype, dy_stream_t stream) {
  DY_REQUIRE(table_dev && n_jobs > 0 && total_blocks > 0 && dtype_size_no_fp8(dtype), DY_ERR_INVALID_ARG, "dy_pack_conv_weights_batched: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const PackEntry* tab = reinterpret_cast<const PackEntry*>(table_dev);
  if (dtype == DY_BF16) hipLaunchKernelGGL((pack_weights_batched_kernel<bf16_t>), dim3((unsigned)total_blocks), dim3(256), 0, st, tab, n_jobs);
  else if (dtype == DY_F16) hipLaunchKernelGGL((pack_weights_batched_kernel<f16_t>), dim3((unsigned)total_blocks), dim3(256), 0, st, tab, n_jobs);
  else hipLaunchKernelGGL((pack_weights_batched_kernel<float>), dim3((unsigned)total_blocks), dim3(256), 0, st, tab, n_jobs);
  return check_launch("pack_weights_batched_kernel");
}
