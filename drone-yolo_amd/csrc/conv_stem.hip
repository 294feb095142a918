// Fused stem: fp32 NCHW image -> 3x3 stride-2 pad-1 convolution (+bias, SiLU) -> NHWC activations.
//
// Replaces, in one pass over the image: the predictor's dtype/layout step (engine/predictor.py:118-136) and
// the model's first layer Conv(3, c2, 3, 2) (yolov8-p2-repvgg.yaml layer 0; nn/modules/conv.py:37-55).
// Without fusion the image is converted to NHWC (written, then read again by the conv): 2 x 6.5 MB per
// 640x640 image of extra HBM traffic on a layer that is purely bandwidth bound (K = 27).
//
// A 256-thread workgroup produces an 8 x 32 patch of output pixels x all output channels:
//   - the (17 x 65) x Cin fp32 input patch is loaded plane by plane with coalesced dword loads into LDS;
//   - each lane assembles its MFMA operand on the fly: lane (quarter lq, pixel lr) gathers the 8 taps
//     k = 8*lq .. 8*lq+7 of its pixel (k = c*9 + r*3 + q, zero beyond K) from LDS at 8 per-lane constant
//     offsets, converts to the storage dtype and packs one 16-byte chunk (im2col never touches memory);
//   - weights (cout x 32, OIHW order padded) sit in registers as MFMA A operands, so a lane ends up with
//     4 consecutive output channels of one pixel; bias + SiLU on registers, per-wave LDS transpose,
//     16-byte stores of whole pixel rows.
// fp32 storage uses two 16-wide k-groups of the fp32 MFMA with the same gather.
#include "common_hip.h"
#include <type_traits>

namespace DY_NS {

struct StemArgs {
  const float* x;
  const unsigned char* x8;  // U8 kernels: uint8 NCHW image, value = x8 / divisor (the training input: preprocess_batch's float() / 255)
  float divisor;
  const void* w;      // [cout_pad16][32] T, k = c*9 + r*3 + q, zero padded
  const float* bias;  // [cout_pad16]
  void* y;
  int N, Cin, H, W, Ho, Wo, Cout, ldy, act;
  int tilesX, tilesY;
  int vec4;  // image rows can be fetched as aligned float4 (W % 4 == 0, 16-byte aligned base)
};

constexpr int kStemTH = 8, kStemTW = 32;
constexpr int kStemPH = 2 * kStemTH + 1, kStemPW = 2 * kStemTW + 1;  // 17 x 65 input patch
constexpr int kStemPitch = 68;  // floats per patch row: 17 aligned float4 = image columns gx0-3 .. gx0+64 (column px at index px+3)
constexpr int kStemShift = 3;

template <typename T, int NF, bool U8 = false>
__global__ __launch_bounds__(256) void conv_stem_kernel(const StemArgs p) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int NKG = 32 / (4 * EPC);  // k-groups covering K padded to 32: 1 (16-bit) or 2 (fp32)
  constexpr int MF = 4;                // per wave: 2 rows x 32 cols = 4 fragments of 16 pixels
  constexpr int BN = NF * 16;
  constexpr int EP_PITCH = BN * (int)sizeof(T) + 16;
  constexpr int EP_BYTES = MF * 16 * EP_PITCH;
  constexpr int PLANE = kStemPH * kStemPitch;

  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  float* patch = reinterpret_cast<float*>(dyn_smem);                       // [Cin][17][67]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lr = lane & 15;
  unsigned char* escr = dyn_smem + ((p.Cin * PLANE * 4 + 15) / 16) * 16 + wave * EP_BYTES;

  int t = (int)xcd_remap(blockIdx.x, gridDim.x);  // neighbouring tiles behind one L2: the halo lines they share are fetched once (r05)
  const int tx = t % p.tilesX;
  t /= p.tilesX;
  const int ty = t % p.tilesY;
  const int n = t / p.tilesY;
  const int y0 = ty * kStemTH, x0 = tx * kStemTW;

  // ---- input patch -> LDS (zero outside the image = the conv's zero padding) ----------------------------
  // Row py of plane c holds image columns gx0-3 .. gx0+64 with gx0 = 2*x0 - 1; x0 is a multiple of 32, so gx0-3 is a
  // multiple of 4: with W % 4 == 0 and a 16-byte aligned image every row is 17 ALIGNED float4 loads, each entirely
  // inside or entirely outside the image (867 wide loads per workgroup instead of 3315 dword loads).
  const int gy0 = 2 * y0 - 1, gx0 = 2 * x0 - 1;
  if (p.vec4) {
    constexpr int V = kStemPitch / 4;  // float4 per row
    for (int i = tid; i < p.Cin * kStemPH * V; i += 256) {
      const int row = i / V, j = i - row * V;
      const int c = row / kStemPH, py = row - c * kStemPH;
      const int gy = gy0 + py, gx = gx0 - kStemShift + 4 * j;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) {
        if constexpr (U8) {
          const unsigned u = *reinterpret_cast<const unsigned*>(p.x8 + ((size_t)(n * p.Cin + c) * p.H + gy) * p.W + gx);
          v = f32x4{(float)(u & 255u) / p.divisor, (float)((u >> 8) & 255u) / p.divisor, (float)((u >> 16) & 255u) / p.divisor, (float)(u >> 24) / p.divisor};
        } else {
          v = *reinterpret_cast<const f32x4*>(p.x + ((size_t)(n * p.Cin + c) * p.H + gy) * p.W + gx);
        }
      }
      *reinterpret_cast<f32x4*>(patch + c * PLANE + py * kStemPitch + 4 * j) = v;
    }
  } else {
    for (int i = tid; i < p.Cin * kStemPH * kStemPW; i += 256) {
      const int c = i / (kStemPH * kStemPW);
      const int r2 = i - c * (kStemPH * kStemPW);
      const int py = r2 / kStemPW, px = r2 - py * kStemPW;
      const int gy = gy0 + py, gx = gx0 + px;
      float v = 0.f;
      if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) {
        const size_t at = ((size_t)(n * p.Cin + c) * p.H + gy) * p.W + gx;
        v = U8 ? (float)p.x8[at] / p.divisor : p.x[at];
      }
      patch[c * PLANE + py * kStemPitch + px + kStemShift] = v;
    }
  }

  // ---- weights -> registers (A operand fragments), per-lane tap offsets -----------------------------------
  u32x4 wfrag[NKG][NF];
#pragma unroll
  for (int kg = 0; kg < NKG; ++kg)
#pragma unroll
    for (int j = 0; j < NF; ++j)
      wfrag[kg][j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.w) + (size_t)(j * 16 + lr) * 32 + kg * 4 * EPC + lq * EPC);
  int koff[NKG][EPC];  // LDS float offset of tap k relative to the pixel's patch origin, or -1 beyond K
#pragma unroll
  for (int kg = 0; kg < NKG; ++kg)
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int k = kg * 4 * EPC + lq * EPC + e;
      const int c = k / 9, r = (k - c * 9) / 3, q = k - c * 9 - r * 3;
      koff[kg][e] = (k < p.Cin * 9) ? c * PLANE + r * kStemPitch + q : -1;
    }
  __syncthreads();

  // ---- MFMA: D[cout][pixel] ------------------------------------------------------------------------------------
  f32x4 acc[MF][NF];
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + j * 16 + lq * 4);
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[i][j] = bb;
  }
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    const int row = wave * 2 + (i >> 1), col = (i & 1) * 16 + lr;  // output pixel inside the 8 x 32 patch
    const float* org = patch + (2 * row) * kStemPitch + 2 * col + kStemShift;
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg) {
      float f[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) f[e] = koff[kg][e] >= 0 ? org[koff[kg][e]] : 0.f;
      const u32x4 a = Chunk<T>::pack(f);
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] = Elem<T>::mma(wfrag[kg][j], a, acc[i][j]);
    }
  }
  mfma_epilogue_fence<T>();

  // ---- epilogue: SiLU, per-wave transpose, 16-byte row stores --------------------------------------------
#pragma unroll
  for (int j = 0; j < NF; ++j)
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (p.act == DY_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f32(v[e]);
      }
      unsigned char* sp = escr + (i * 16 + lr) * EP_PITCH + (j * 16 + lq * 4) * (int)sizeof(T);
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
  constexpr int CPR = BN * (int)sizeof(T) / 16;
  constexpr int VE = 16 / (int)sizeof(T);
  T* __restrict__ yg = reinterpret_cast<T*>(p.y);
#pragma unroll
  for (int k = 0; k < MF * 16 * CPR / 64; ++k) {
    const int idx = k * 64 + lane;
    const int pixl = idx / CPR, cc = idx - pixl * CPR;
    const int i = pixl >> 4;
    const int yy = y0 + wave * 2 + (i >> 1), xx = x0 + (i & 1) * 16 + (pixl & 15);
    const int co = cc * VE;
    const u32x4 val = *reinterpret_cast<const u32x4*>(escr + pixl * EP_PITCH + cc * 16);
    if (yy < p.Ho && xx < p.Wo && co < p.Cout) {
      T* yp = yg + ((size_t)(n * p.Ho + yy) * p.Wo + xx) * (size_t)p.ldy + co;
      if (co + VE <= p.Cout) {
        *reinterpret_cast<u32x4*>(yp) = val;
      } else {
        typedef __attribute__((ext_vector_type(VE))) T vt;
        const vt sv = __builtin_bit_cast(vt, val);
#pragma unroll
        for (int e = 0; e < VE; ++e)
          if (e < p.Cout - co) yp[e] = sv[e];
      }
    }
  }
}

// DY_F16X2 (split float16, include/dyolo.h) form of the same kernel: the gathered taps are split into (hi, lo) on the fly — the image
// itself is an fp32 operand of the reference's first convolution, so it gets the same 22 bits as every later activation — and each
// (pixel fragment, cout fragment) takes the three 16-bit MFMAs w_hi x_hi + w_lo x_hi + (w_hi 2^-11) x_lo.  p.w = [cout_pad16][32] hi
// halves, then as many lo halves, then fp32[cout_pad16] inverse row scales (the rows were scaled into [2^13, 2^14) when packed); the
// output leaves as [hi x 8 | lo x 8] groups.  Replaces the layout cast + flat-K launch the type's first version ran: 3.0 -> ~1 ms at B = 256.
#ifndef DYOLO_L2E_BUILD
template <int NF>
__global__ __launch_bounds__(256) void conv_stem_split_kernel(const StemArgs p) {
  constexpr int MF = 4;
  constexpr int BN = NF * 16;
  constexpr int EP_PITCH = BN * 4 + 16;
  constexpr int EP_BYTES = MF * 16 * EP_PITCH;
  constexpr int PLANE = kStemPH * kStemPitch;

  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  float* patch = reinterpret_cast<float*>(dyn_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lr = lane & 15;
  unsigned char* escr = dyn_smem + ((p.Cin * PLANE * 4 + 15) / 16) * 16 + wave * EP_BYTES;

  int t = (int)xcd_remap(blockIdx.x, gridDim.x);  // neighbouring tiles behind one L2: the halo lines they share are fetched once (r05)
  const int tx = t % p.tilesX;
  t /= p.tilesX;
  const int ty = t % p.tilesY;
  const int n = t / p.tilesY;
  const int y0 = ty * kStemTH, x0 = tx * kStemTW;
  const int gy0 = 2 * y0 - 1, gx0 = 2 * x0 - 1;
  if (p.vec4) {
    constexpr int V = kStemPitch / 4;
    for (int i = tid; i < p.Cin * kStemPH * V; i += 256) {
      const int row = i / V, j = i - row * V;
      const int c = row / kStemPH, py = row - c * kStemPH;
      const int gy = gy0 + py, gx = gx0 - kStemShift + 4 * j;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) v = *reinterpret_cast<const f32x4*>(p.x + ((size_t)(n * p.Cin + c) * p.H + gy) * p.W + gx);
      *reinterpret_cast<f32x4*>(patch + c * PLANE + py * kStemPitch + 4 * j) = v;
    }
  } else {
    for (int i = tid; i < p.Cin * kStemPH * kStemPW; i += 256) {
      const int c = i / (kStemPH * kStemPW);
      const int r2 = i - c * (kStemPH * kStemPW);
      const int py = r2 / kStemPW, px = r2 - py * kStemPW;
      const int gy = gy0 + py, gx = gx0 + px;
      float v = 0.f;
      if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) v = p.x[((size_t)(n * p.Cin + c) * p.H + gy) * p.W + gx];
      patch[c * PLANE + py * kStemPitch + px + kStemShift] = v;
    }
  }
  const int cp16 = (p.Cout + 15) / 16 * 16;
  const f16_t* wh = reinterpret_cast<const f16_t*>(p.w);
  const f16_t* wl = wh + (size_t)cp16 * 32;
  const float* wsc = reinterpret_cast<const float*>(wl + (size_t)cp16 * 32);
  u32x4 fh[NF], fl[NF], fs[NF];
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    fh[j] = *reinterpret_cast<const u32x4*>(wh + (size_t)(j * 16 + lr) * 32 + lq * 8);
    fl[j] = *reinterpret_cast<const u32x4*>(wl + (size_t)(j * 16 + lr) * 32 + lq * 8);
    const f16x8 sv = __builtin_bit_cast(f16x8, fh[j]) * (f16_t)kSplitInv;
    fs[j] = __builtin_bit_cast(u32x4, sv);
  }
  int koff[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = lq * 8 + e;
    const int c = k / 9, r = (k - c * 9) / 3, q = k - c * 9 - r * 3;
    koff[e] = (k < p.Cin * 9) ? c * PLANE + r * kStemPitch + q : -1;
  }
  __syncthreads();

  f32x4 acc[MF][NF];
#pragma unroll
  for (int j = 0; j < NF; ++j)
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    const int row = wave * 2 + (i >> 1), col = (i & 1) * 16 + lr;
    const float* org = patch + (2 * row) * kStemPitch + 2 * col + kStemShift;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = koff[e] >= 0 ? org[koff[e]] : 0.f;
    u32x4 ah, al;
    split8(f, ah, al);
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      acc[i][j] = Elem<f16_t>::mma(fh[j], ah, acc[i][j]);
      acc[i][j] = Elem<f16_t>::mma(fl[j], ah, acc[i][j]);
      acc[i][j] = Elem<f16_t>::mma(fs[j], al, acc[i][j]);
    }
  }
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + j * 16 + lq * 4), sc = *reinterpret_cast<const f32x4*>(wsc + j * 16 + lq * 4);
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * sc[e] + bb[e];
      if (p.act == DY_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f32(v[e]);
      }
      *reinterpret_cast<f32x4*>(escr + (i * 16 + lr) * EP_PITCH + (j * 16 + lq * 4) * 4) = f32x4{v[0], v[1], v[2], v[3]};
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  constexpr int GPR = BN / 8;  // groups of 8 channels per pixel
  unsigned char* __restrict__ yb = reinterpret_cast<unsigned char*>(p.y);
#pragma unroll
  for (int k = 0; k < (MF * 16 * GPR + 63) / 64; ++k) {
    const int idx = k * 64 + lane;
    const int pixl = idx / GPR, g = idx - pixl * GPR;
    const int i = pixl >> 4;
    const int yy = y0 + wave * 2 + (i >> 1), xx = x0 + (i & 1) * 16 + (pixl & 15);
    if (pixl < MF * 16 && yy < p.Ho && xx < p.Wo && g * 8 < p.Cout) {
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(escr + pixl * EP_PITCH + g * 32), t1 = *reinterpret_cast<const f32x4*>(escr + pixl * EP_PITCH + g * 32 + 16);
      const float f[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
      u32x4 hi, lo;
      split8(f, hi, lo);
      unsigned char* dp = yb + (((size_t)(n * p.Ho + yy) * p.Wo + xx) * (size_t)p.ldy + (size_t)g * 8) * 4;
      *reinterpret_cast<u32x4*>(dp) = hi;
      *reinterpret_cast<u32x4*>(dp + 16) = lo;
    }
  }
}

template <int NF>
static int launch_stem_split(const StemArgs& a, hipStream_t st) {
  StemArgs p = a;
  p.tilesX = (p.Wo + kStemTW - 1) / kStemTW;
  p.tilesY = (p.Ho + kStemTH - 1) / kStemTH;
  p.vec4 = (p.W % 4 == 0 && (reinterpret_cast<uintptr_t>(p.x) & 15) == 0);
  const int smem = ((p.Cin * kStemPH * kStemPitch * 4 + 15) / 16) * 16 + 4 * 4 * 16 * (NF * 16 * 4 + 16);
  auto kern = conv_stem_split_kernel<NF>;
  static const hipError_t once = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)once;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.N * p.tilesY * p.tilesX)), dim3(256), smem, st, p);
  return check_launch("conv_stem_split_kernel");
}
#endif

template <typename T, int NF, bool U8 = false>
static int launch_stem(const StemArgs& a, hipStream_t st) {
  StemArgs p = a;
  p.tilesX = (p.Wo + kStemTW - 1) / kStemTW;
  p.tilesY = (p.Ho + kStemTH - 1) / kStemTH;
  p.vec4 = U8 ? (p.W % 4 == 0 && (reinterpret_cast<uintptr_t>(p.x8) & 3) == 0 && ((size_t)p.H * p.W) % 4 == 0) : (p.W % 4 == 0 && (reinterpret_cast<uintptr_t>(p.x) & 15) == 0);
  const int smem = ((p.Cin * kStemPH * kStemPitch * 4 + 15) / 16) * 16 + 4 * 4 * 16 * (NF * 16 * (int)sizeof(T) + 16);
  auto kern = conv_stem_kernel<T, NF, U8>;
  static const hipError_t once = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)once;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.N * p.tilesY * p.tilesX)), dim3(256), smem, st, p);
  return check_launch("conv_stem_kernel");
}

template <typename T>
static int launch_stem_dtype(const StemArgs& a, hipStream_t st) {
  switch ((a.Cout + 15) / 16) {
    case 1: return launch_stem<T, 1>(a, st);
    case 2: return launch_stem<T, 2>(a, st);
    case 3: return launch_stem<T, 3>(a, st);
    case 4: return launch_stem<T, 4>(a, st);
    case 5: return launch_stem<T, 5>(a, st);
    default:
      set_error("dy_stem_conv3x3s2_nchw: cout %d > 80 not built", a.Cout);
      return DY_ERR_UNSUPPORTED;
  }
}

}  // namespace DY_NS

using namespace DY_NS;

#ifndef DYOLO_L2E_BUILD
namespace dy_l2e {
int32_t stem_entry(const float* x, const void* w, const float* bias, void* y, int32_t n, int32_t cin, int32_t h, int32_t w_in, int32_t cout, int32_t ld_y, int32_t act,
                   int32_t dtype, dy_stream_t stream);
}
namespace dy {
int32_t stem_entry(const float* x, const void* w, const float* bias, void* y, int32_t n, int32_t cin, int32_t h, int32_t w_in, int32_t cout, int32_t ld_y, int32_t act,
                   int32_t dtype, dy_stream_t stream);
}
extern "C" int32_t dy_stem_conv3x3s2_nchw(const float* x, const void* w, const float* bias, void* y, int32_t n, int32_t cin,
                                          int32_t h, int32_t w_in, int32_t cout, int32_t ld_y, int32_t act, int32_t dtype,
                                          dy_stream_t stream) {
  if (act == DY_ACT_SILU_L2E) return dy_l2e::stem_entry(x, w, bias, y, n, cin, h, w_in, cout, ld_y, DY_ACT_SILU, dtype, stream);
  return dy::stem_entry(x, w, bias, y, n, cin, h, w_in, cout, ld_y, act, dtype, stream);
}

// uint8 source (training: DetectionTrainer.preprocess_batch's `img.float() / 255`, models/yolo/detect/train.py:57-60, folded into the stem):
// 32 output channels, 16-bit storage -- the Drone-YOLO / YOLOv8 stem of the s scale.
extern "C" int32_t dy_stem_conv3x3s2_nchw_u8(const uint8_t* x, float divisor, const void* w, const float* bias, void* y, int32_t n, int32_t cin, int32_t h,
                                             int32_t w_in, int32_t cout, int32_t ld_y, int32_t act, int32_t dtype, dy_stream_t stream) {
  DY_REQUIRE(x && w && bias && y && divisor > 0.f, DY_ERR_INVALID_ARG, "dy_stem_conv3x3s2_nchw_u8: null pointer or bad divisor");
  DY_REQUIRE((dtype == DY_BF16 || dtype == DY_F16) && n > 0 && h > 0 && w_in > 0 && cin >= 1 && cin * 9 <= 32 && cout > 0 && cout <= 80 && cout % 16 == 0 &&
                 act != DY_ACT_SILU_L2E, DY_ERR_UNSUPPORTED, "dy_stem_conv3x3s2_nchw_u8: 16-bit storage, cin <= 3, cout a multiple of 16 up to 80");
  DY_REQUIRE(ld_y >= cout && (ld_y * 2) % 16 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0, DY_ERR_INVALID_ARG,
             "dy_stem_conv3x3s2_nchw_u8: y / w must be 16-byte aligned, ld_y whole chunks");
  dy::StemArgs a{};
  a.x = nullptr, a.x8 = x, a.divisor = divisor, a.w = w, a.bias = bias, a.y = y;
  a.N = n, a.Cin = cin, a.H = h, a.W = w_in, a.Ho = (h - 1) / 2 + 1, a.Wo = (w_in - 1) / 2 + 1, a.Cout = cout, a.ldy = ld_y, a.act = act;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define DY_STEM_U8(T)                                               \
  switch (cout / 16) {                                               \
    case 1: return dy::launch_stem<T, 1, true>(a, st);               \
    case 2: return dy::launch_stem<T, 2, true>(a, st);               \
    case 3: return dy::launch_stem<T, 3, true>(a, st);               \
    case 4: return dy::launch_stem<T, 4, true>(a, st);               \
    default: return dy::launch_stem<T, 5, true>(a, st);              \
  }
  if (dtype == DY_BF16) { DY_STEM_U8(dy::bf16_t) }
  DY_STEM_U8(dy::f16_t)
#undef DY_STEM_U8
}
#endif

namespace DY_NS {
int32_t stem_entry(const float* x, const void* w, const float* bias, void* y, int32_t n, int32_t cin, int32_t h, int32_t w_in, int32_t cout, int32_t ld_y, int32_t act,
                   int32_t dtype, dy_stream_t stream) {
  const int es = dtype == DY_F16X2 ? 4 : dtype_size_no_fp8(dtype);
  DY_REQUIRE(x && w && bias && y && es, DY_ERR_INVALID_ARG, "dy_stem_conv3x3s2_nchw: null pointer or bad dtype");
  DY_REQUIRE(n > 0 && h > 0 && w_in > 0 && cout > 0 && cin >= 1 && cin * 9 <= 32, DY_ERR_INVALID_ARG,
             "dy_stem_conv3x3s2_nchw: needs 1 <= cin <= 3 (K = 9*cin <= 32)");
  DY_REQUIRE(ld_y >= cout && (ld_y * es) % 16 == 0 && aligned16(y) && aligned16(w) && aligned16(bias), DY_ERR_INVALID_ARG,
             "dy_stem_conv3x3s2_nchw: y/w/bias must be 16-byte aligned, output pitch a multiple of 16 bytes");
  StemArgs a{};
  a.x = x;
  a.w = w;
  a.bias = bias;
  a.y = y;
  a.N = n;
  a.Cin = cin;
  a.H = h;
  a.W = w_in;
  a.Ho = (h + 2 - 3) / 2 + 1;
  a.Wo = (w_in + 2 - 3) / 2 + 1;
  a.Cout = cout;
  a.ldy = ld_y;
  a.act = act;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DY_F16X2) {
#ifndef DYOLO_L2E_BUILD
    DY_REQUIRE(cout % 8 == 0 && cout <= 64, DY_ERR_UNSUPPORTED, "dy_stem_conv3x3s2_nchw: DY_F16X2 is built for cout a multiple of 8 up to 64");
    switch ((cout + 15) / 16) {
      case 1: return launch_stem_split<1>(a, st);
      case 2: return launch_stem_split<2>(a, st);
      case 3: return launch_stem_split<3>(a, st);
      default: return launch_stem_split<4>(a, st);
    }
#else
    set_error("dy_stem_conv3x3s2_nchw: DY_F16X2 runs in the reference's activation units");
    return DY_ERR_UNSUPPORTED;
#endif
  }
  switch (dtype) {
    case DY_BF16: return launch_stem_dtype<bf16_t>(a, st);
    case DY_F16: return launch_stem_dtype<f16_t>(a, st);
    default: return launch_stem_dtype<float>(a, st);
  }
}
}  // namespace DY_NS
