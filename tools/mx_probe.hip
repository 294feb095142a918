// gfx950 block-scaled fp8 MFMA (v_mfma_[scale_]f32_16x16x128_f8f6f4 with e4m3 operands): operand lane map and rate.
// build: hipcc -O3 --offload-arch=gfx950 tools/mx_probe.hip -o tools/bin/mx_probe ; run on the GPU box.
//
// (1) LAYOUT.  Hypothesis H1 (the 16x16x32 family's map stretched to K = 128): lane l holds row (A) / column (B) l & 15 and
//     k = 32 * (l >> 4) + j in byte j of its 32-byte fragment (register j / 4, byte j % 4).  Checked with exact small-integer data
//     (every e4m3 value used is an integer <= 8, every product and sum exact in fp32) and an ASYMMETRIC B, for the unscaled opcode
//     (scale operands 0: hipcc emits v_mfma_f32_16x16x128_f8f6f4, implicit scale 1) and the scaled one with E8M0 127 = 2^0.
// (2) RATE.  Waves that do nothing but that MFMA on registers, against v_mfma_f32_16x16x32_fp8_fp8 and _bf16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static unsigned char e4m3_of_int(int v) {  // exact encodings of 0..8 (bias 7: 1 -> 0x38, 2 -> 0x40, 3 -> 0x44, 4 -> 0x48, 5 -> 0x4a, 6 -> 0x4c, 7 -> 0x4e, 8 -> 0x50)
  static const unsigned char t[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50};
  return t[v];
}

template <int SCALED>
__global__ void layout_kernel(const unsigned char* a, const unsigned char* b, float* c) {
  // a: [16][128] row-major e4m3, b: [128][16] (k-major) e4m3 -> c [16][16]
  const int l = threadIdx.x, r = l & 15, q = l >> 4;
  i32x8 av, bv;
  unsigned char ab[32], bb[32];
  for (int j = 0; j < 32; ++j) {
    ab[j] = a[r * 128 + 32 * q + j];
    bb[j] = b[(32 * q + j) * 16 + r];
  }
  memcpy(&av, ab, 32);
  memcpy(&bv, bb, 32);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (SCALED) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, 0, 0, 0);
  // C/D map of the 16x16 shapes: col = lane & 15, row = (lane >> 4) * 4 + reg
  for (int i = 0; i < 4; ++i) c[(q * 4 + i) * 16 + r] = acc[i];
}

template <int KIND, int NACC>
__global__ __launch_bounds__(1024) void rate_kernel(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = 0x38404448 + (int)threadIdx.x * 0x01010101 % 7;
    b[i] = 0x3c3a3834 ^ (i * 0x00010001);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0, 0, 0);
      else if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      else if (KIND == 2) {
        const long al = ((long)a[1] << 32) | (unsigned)a[0], bl = ((long)b[1] << 32) | (unsigned)b[0];
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(al, bl, acc[i], 0, 0, 0);
      } else {
        typedef __attribute__((ext_vector_type(4))) int i32x4;
        const i32x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x8, b4), acc[i], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  // ---- (1) layout ----
  std::vector<unsigned char> ha(16 * 128), hb(128 * 16);
  std::vector<int> ia(16 * 128), ib(128 * 16);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int)((s >> 20) % 9); };
  for (int i = 0; i < 16 * 128; ++i) { ia[i] = rnd(); ha[i] = e4m3_of_int(ia[i]); }
  for (int i = 0; i < 128 * 16; ++i) { ib[i] = rnd(); hb[i] = e4m3_of_int(ib[i]); }
  unsigned char *da, *db;
  float* dc;
  hipMalloc(&da, ha.size());
  hipMalloc(&db, hb.size());
  hipMalloc(&dc, 256 * sizeof(float));
  hipMemcpy(da, ha.data(), ha.size(), hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), hb.size(), hipMemcpyHostToDevice);
  int bad_total = 0;
  for (int scaled = 0; scaled < 2; ++scaled) {
    if (scaled) hipLaunchKernelGGL(layout_kernel<1>, dim3(1), dim3(64), 0, 0, da, db, dc);
    else hipLaunchKernelGGL(layout_kernel<0>, dim3(1), dim3(64), 0, 0, da, db, dc);
    std::vector<float> hc(256);
    hipMemcpy(hc.data(), dc, 256 * sizeof(float), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        long ref = 0;
        for (int k = 0; k < 128; ++k) ref += (long)ia[i * 128 + k] * ib[k * 16 + j];
        if ((long)hc[i * 16 + j] != ref) {
          if (bad < 4) printf("  mismatch (%d,%d): got %.1f want %ld\n", i, j, hc[i * 16 + j], ref);
          ++bad;
        }
      }
    printf("layout H1 (row = lane & 15, k = 32 * (lane >> 4) + byte), %s opcode: %s (%d of 256 wrong)\n", scaled ? "scaled (E8M0 127)" : "unscaled", bad ? "WRONG" : "EXACT", bad);
    bad_total += bad;
  }
  // ---- (2) rate ----
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  float* out;
  hipMalloc(&out, 1024 * 1024 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[4] = {"mx fp8 16x16x128 unscaled", "mx fp8 16x16x128 scaled", "fp8 16x16x32", "bf16 16x16x32"};
  const double kk[4] = {128, 128, 32, 32};
  for (int kind = 0; kind < 4; ++kind)
    for (int wpc : {4, 8, 16}) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL((rate_kernel<0, 8>), dim3(256), dim3(wpc * 64), 0, 0, out, iters);
        else if (kind == 1) hipLaunchKernelGGL((rate_kernel<1, 8>), dim3(256), dim3(wpc * 64), 0, 0, out, iters);
        else if (kind == 2) hipLaunchKernelGGL((rate_kernel<2, 8>), dim3(256), dim3(wpc * 64), 0, 0, out, iters);
        else hipLaunchKernelGGL((rate_kernel<3, 8>), dim3(256), dim3(wpc * 64), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double flops = 256.0 * wpc * (double)iters * 8 * (2.0 * 16 * 16 * kk[kind]);
      printf("%-28s waves/CU=%2d: %.3f ms  %.0f TFLOP/s\n", names[kind], wpc, best, flops / best / 1e9);
    }
  return bad_total ? 1 : 0;
}
