// Cost of the SiLU epilogue arithmetic: cycles per wave64 instruction sequence, 2 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o tools/bin/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float seed) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) v[i] = __builtin_amdgcn_exp2f(v[i]);
      if (MODE == 1) v[i] = __builtin_amdgcn_rcpf(v[i]);
      if (MODE == 2) { const float e = __builtin_amdgcn_exp2f(v[i] * -1.4426950408889634f); v[i] = v[i] * __builtin_amdgcn_rcpf(1.0f + e); }
      if (MODE == 3) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
      if (MODE == 4) {  // rational sigmoid stand-in: 8 FMAs
        float t = v[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) t = __builtin_fmaf(t, v[i], 0.25f);
        v[i] = t;
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, float* out, int iters, int ops) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, out, iters, 0.37f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 2 waves x iters x 8 sequences
    const double seq = 2.0 * iters * 8;
    if (rep) printf("%-28s %.3f ms  -> %.1f cycles @2.4GHz per wave64 sequence (%d VALU ops)\n", name, ms, ms * 1e-3 * 2.4e9 / seq, ops);
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float* out;
  (void)hipMalloc(&out, 1024 * 1024 * sizeof(float));
  run<3>("v_fma_f32", out, iters, 1);
  run<0>("v_exp_f32", out, iters, 1);
  run<1>("v_rcp_f32", out, iters, 1);
  run<2>("silu (mul exp add rcp mul)", out, iters, 5);
  run<4>("8 dependent fma", out, iters, 8);
  return 0;
}
