// Practical MFMA ceiling of the device: waves that do nothing but v_mfma_f32_16x16x32_bf16 on registers.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o tools/bin/mfma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  f32x4 acc[NACC];
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float* out;
  hipMalloc(&out, 1024 * 1024 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wpc : {4, 8}) {           // waves per CU (1 or 2 per SIMD)
    for (int rep = 0; rep < 3; ++rep) {
      const int threads = wpc * 64;
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe<16>, dim3(256), dim3(threads), 0, 0, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = 256.0 * wpc * (double)iters * 16 * (2.0 * 16 * 16 * 32);
      printf("waves/CU=%d iters=%d: %.3f ms  %.1f TFLOP/s  (%.2f GHz-equivalent at 1024 flop/clk/SIMD... %.0f cycles@2.4GHz per MFMA)\n", wpc, iters, ms,
             flops / ms / 1e9, flops / ms / 1e9 / (256 * 4 * 1024.0) * 1e3 / 1e3, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * (wpc / 4)));
    }
  }
  return 0;
}
