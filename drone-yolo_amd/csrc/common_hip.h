// Shared device/host helpers for libdyolo (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include "dyolo.h"

// The convolution sources are compiled TWICE (Makefile): once as they stand (namespace dy: SiLU in the reference's units, DY_ACT_SILU)
// and once with -DDYOLO_L2E_BUILD into namespace dy_l2e, where silu_f32 is the log2(e)-domain formula (DY_ACT_SILU_L2E, include/dyolo.h):
// the extern "C" entry points (first compilation only) pick the namespace from the activation code, so no kernel carries a run-time
// branch on it.  (A run-time wave-uniform branch in the epilogues was tried first: the duplicated epilogue code cost 2.4 % of the
// whole pass, more than the saved instruction gains.)
#ifdef DYOLO_L2E_BUILD
#define DY_NS dy_l2e
#else
#define DY_NS dy
#endif

namespace dy {
// ---- host-side error plumbing and the zeroing launch: ONE copy, in namespace dy (defined in api.cpp) ----
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// Zero `bytes` (a multiple of 4) at `p` on `stream` with a KERNEL.  Never hipMemsetAsync in this library: captured into a hipGraph the
// memset becomes a memset NODE, and on ROCm 7.0 such a node was seen to lose its order against the kernels around it — replays of a
// captured training step returned a doubled BCE sum (the accumulators were cleared at the wrong time).
void zero_async(void* p, size_t bytes, hipStream_t stream);
void note_stats(int written);  // dy_conv_stats_written(): set by a convolution launch that fills dy_conv_desc.bn_stats
}  // namespace dy
constexpr int kStatSlots = 1024;  // partial-sum slots of a BatchNorm workspace (bn_train.hip: kBnMaxSlabs; dy_bn_workspace_bytes)

namespace DY_NS {
#ifdef DYOLO_L2E_BUILD
using ::dy::check_launch;
using ::dy::note_stats;
using ::dy::set_error;
using ::dy::zero_async;
#endif

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;  // one 16-byte chunk
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
struct fp8_t {  // OCP e4m3fn, one byte (gfx950's fp8; MI300's fnuz encoding is a different format)
  unsigned char v;
};

constexpr int kWave = 64;

// Ablation switches (kernel variants / timing probes picked by DYOLO_* environment variables) exist only in builds made with
// -DDYOLO_ABLATE (tools/ab_conv.sh, `make ABLATE=1`).  The product library never reads the environment: a stray variable
// cannot select another kernel or skip work.
#ifdef DYOLO_ABLATE
static inline int dy_ablate(const char* name) {
  const char* v = getenv(name);
  return v ? atoi(v) : 0;
}
#else
static constexpr int dy_ablate(const char*) { return 0; }
#endif

// ---- element traits -----------------------------------------------------------
// EPC = elements per 16-byte chunk.  A "k-group" is 4 chunks (the K extent one
// lane-quarter layout of a 16x16 MFMA covers): 32 k for bf16/f16, 16 k for f32.
template <typename T> struct Elem;

template <> struct Elem<bf16_t> {
  static constexpr int EPC = 8;
  static __device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
  }
  static __device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
  static __device__ __forceinline__ bf16_t from_f32(float v) { return (bf16_t)v; }
};

template <> struct Elem<f16_t> {
  static constexpr int EPC = 8;
  static __device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                  0);
  }
  static __device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
  static __device__ __forceinline__ f16_t from_f32(float v) { return (f16_t)v; }
};

// fp8 (e4m3fn): 16 elements per 16-byte chunk, so a k-group (4 lane quarters) is 64 channels and takes TWO
// v_mfma_f32_16x16x32_fp8_fp8 (8 bytes per lane each): instruction j consumes bytes 8j..8j+7 of every quarter's chunk.
// A and B agree on that (quarter, byte) -> k map, which is all a dot product needs.  Same cycles per MFMA as bf16.
template <> struct Elem<fp8_t> {
  static constexpr int EPC = 16;
  static __device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    typedef __attribute__((ext_vector_type(2))) long l2;
    const l2 al = __builtin_bit_cast(l2, a), bl = __builtin_bit_cast(l2, b);
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(al[0], bl[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(al[1], bl[1], c, 0, 0, 0);
    return c;
  }
  static __device__ __forceinline__ float to_f32(fp8_t v) { return __builtin_amdgcn_cvt_f32_fp8((int)v.v, 0); }
  static __device__ __forceinline__ fp8_t from_f32(float v) {  // saturating (|v| > 448 -> +-448), round to nearest even
    v = __builtin_fminf(__builtin_fmaxf(v, -448.f), 448.f);
    return fp8_t{(unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false) & 0xff)};
  }
};

template <> struct Elem<float> {
  static constexpr int EPC = 4;
  // Lane-quarter q holds k = 4q..4q+3 of the 16-k group in its chunk; instruction j
  // consumes element j of every quarter, so A and B agree on the (q, j) -> k map.
  static __device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], c, 0, 0, 0);
    return c;
  }
  static __device__ __forceinline__ float to_f32(float v) { return v; }
  static __device__ __forceinline__ float from_f32(float v) { return v; }
};

// DY_F16X2 (include/dyolo.h): split float16 storage, x ~= hi + lo * 2^-11.  A 4-byte element like fp32 for every address computation
// (EPC = 4: a 16-byte chunk is the hi OR the lo halves of 8 channels; two adjacent chunks = 8 channels).  No mma(): the kernels that
// take the type issue their three 16-bit MFMAs themselves (csrc/conv_gemm_fk.hip).
struct f16x2_t {
  unsigned v;
};
template <> struct Elem<f16x2_t> {
  static constexpr int EPC = 4;
};
constexpr float kSplitScale = 2048.f, kSplitInv = 1.f / 2048.f;  // 2^11: the lo half is stored scaled into float16's normal range
// 8 fp32 values -> their hi chunk and lo chunk.  hi = rn_f16(x), flushed to 0 below float16's smallest normal (so that a matrix pipe
// that flushes denormal inputs and one that does not compute the same thing; the whole value then sits in lo: |x| 2^11 < 0.125) and
// clamped to the finite range; lo = rn_f16((x - hi) 2^11), |lo| <= |x| / 2.
__device__ __forceinline__ void split8(const float (&f)[8], u32x4& hi, u32x4& lo) {
  f16x8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = __builtin_fminf(__builtin_fmaxf(f[e], -65504.f), 65504.f);
    const f16_t hh = __builtin_fabsf(x) < 6.103515625e-5f ? (f16_t)0.f : (f16_t)x;
    h[e] = hh;
    l[e] = (f16_t)((x - (float)hh) * kSplitScale);
  }
  hi = __builtin_bit_cast(u32x4, h);
  lo = __builtin_bit_cast(u32x4, l);
}
__device__ __forceinline__ void join8(u32x4 hi, u32x4 lo, float (&f)[8]) {
  const f16x8 h = __builtin_bit_cast(f16x8, hi), l = __builtin_bit_cast(f16x8, lo);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)h[e] + (float)l[e] * kSplitInv;  // exact in fp32: 11 + 11 significant bits
}

// Unpack / pack one 16-byte chunk to fp32 lanes.
template <typename T> struct Chunk {
  static constexpr int EPC = Elem<T>::EPC;
  static __device__ __forceinline__ void unpack(u32x4 v, float (&f)[EPC]) {
    typedef __attribute__((ext_vector_type(EPC))) T vec_t;
    vec_t t = __builtin_bit_cast(vec_t, v);
#pragma unroll
    for (int i = 0; i < EPC; ++i) f[i] = Elem<T>::to_f32(t[i]);
  }
  static __device__ __forceinline__ u32x4 pack(const float (&f)[EPC]) {
    typedef __attribute__((ext_vector_type(EPC))) T vec_t;
    vec_t t;
#pragma unroll
    for (int i = 0; i < EPC; ++i) t[i] = Elem<T>::from_f32(f[i]);
    return __builtin_bit_cast(u32x4, t);
  }
};

template <> struct Chunk<fp8_t> {
  static constexpr int EPC = 16;
  static __device__ __forceinline__ void unpack(u32x4 v, float (&f)[16]) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      f[4 * w + 0] = __builtin_amdgcn_cvt_f32_fp8((int)v[w], 0);
      f[4 * w + 1] = __builtin_amdgcn_cvt_f32_fp8((int)v[w], 1);
      f[4 * w + 2] = __builtin_amdgcn_cvt_f32_fp8((int)v[w], 2);
      f[4 * w + 3] = __builtin_amdgcn_cvt_f32_fp8((int)v[w], 3);
    }
  }
  static __device__ __forceinline__ u32x4 pack(const float (&f)[16]) {
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float c[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) c[e] = __builtin_fminf(__builtin_fmaxf(f[4 * w + e], -448.f), 448.f);
      int r = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
      r = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], r, true);
      o[w] = (unsigned)r;
    }
    return o;
  }
};

// SiLU x*sigmoid(x) with the two hardware transcendentals (v_exp_f32, v_rcp_f32; ~1 ulp each) instead of
// an IEEE division: 5 VALU instructions per element.  The epilogue runs on every output element of the
// network, and on 64-channel 3x3 layers its VALU time is comparable to the MFMA time.
__device__ __forceinline__ float silu_f32(float x) {
#ifdef DYOLO_L2E_BUILD
  // DY_ACT_SILU_L2E: x is t = log2(e) * z (biases were packed times log2 e), sigmoid(z) = 1 / (1 + 2^-t): no multiply in front of
  // v_exp_f32, and t * sigmoid(z) = log2(e) * silu(z) is the next layer's scaled input — 4 VALU instructions per element instead of 5
  const float e = __builtin_amdgcn_exp2f(-x);
#else
  const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
#endif
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// Fence between the last MFMA of a tile and the epilogue's VALU code, fp32 (v_mfma_f32_16x16x4_f32) only.
// hipcc (ROCm 7.2) was seen to interleave epilogue v_adds that overwrite a register quad right behind an
// in-flight fp32 MFMA still reading that quad as its C operand, with only `s_nop 8` in between: elements 2,3
// of the accumulator came out wrong (WAR wait-state miscount for this 8-pass shape).  Pin the order and wait
// out the longest MFMA before any epilogue VALU write.  bf16/f16 shapes are unaffected and skip this.
template <typename T>
__device__ __forceinline__ void mfma_epilogue_fence() {
  if constexpr (sizeof(T) == 4) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ u32x4 zero_chunk() { return u32x4{0u, 0u, 0u, 0u}; }

// XCD-aware block remap (guide T1, bijective form): blocks b and b+8 share an XCD's
// L2, so hand every XCD a contiguous range of logical tile ids.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, i = bid >> 3;
  const unsigned base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
  return base + i;
}

// ---- exact unsigned division by a runtime-invariant divisor (Granlund-Montgomery), for n < 2^31 -------------
struct FastDiv {
  unsigned m;  // magic multiplier
  int l;       // ceil(log2(d)); 0 means d == 1
};
static inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.l = 0;
  while ((1ull << f.l) < d) ++f.l;
  f.m = (unsigned)(((1ull << 32) * ((1ull << f.l) - d)) / d + 1);
  return f;
}
__device__ __forceinline__ unsigned fastdiv(unsigned n, FastDiv f) {
  if (f.l == 0) return n;
  const unsigned t = __umulhi(f.m, n);
  return (t + ((n - t) >> 1)) >> (f.l - 1);
}

// element size for entry points that are built for the plain types only — not DY_FP8, not DY_F16X2 (they then report "bad dtype")
static inline int dtype_size_no_fp8(int dtype) { return (dtype == DY_FP8 || dtype == DY_F16X2) ? 0 : dy_dtype_size(dtype); }

// ---- host-side error plumbing: set_error / check_launch / zero_async are declared at the top (namespace dy, one copy in api.cpp) ----

#define DY_REQUIRE(cond, code, ...)   \
  do {                                \
    if (!(cond)) {                    \
      ::dy::set_error(__VA_ARGS__);   \
      return (code);                  \
    }                                 \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace DY_NS
