// Can v_mfma_f64_16x16x4_f64 and ordinary fp64 VALU work of the SAME wave (or of two waves on one SIMD) overlap on gfx950?
// Three kernels, identical launch shape: (a) N MFMAs per iteration on 4 independent accumulators, (b) M independent v_fma_f64
// per iteration, (c) both interleaved in program order.  If t(c) ~ max(t(a), t(b)) the pipes overlap; if t(c) ~ t(a) + t(b)
// they serialise.   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = d4{seed, seed, seed, seed};
  double a = seed + threadIdx.x * 1e-9, b = 1.0000001;
  double v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed * (i + 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE != 1) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
      if (MODE != 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], b, a);     // 16 independent fp64 FMAs = 64 cycles of VALU issue
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static float run(double* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double* out;
  hipMalloc(&out, sizeof(double) * 256 * 4096);
  const int iters = 20000;
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int blocks = 256 * waves_per_simd;      // 256 CUs x 4 SIMDs: one 256-thread block = one wave per SIMD
    const float a = run<0>(out, blocks, iters), b = run<1>(out, blocks, iters), c = run<2>(out, blocks, iters);
    printf("%d wave(s)/SIMD: mfma only %.3f ms, valu only %.3f ms, interleaved %.3f ms  (sum %.3f, max %.3f)\n", waves_per_simd, a, b, c, a + b,
           a > b ? a : b);
  }
  return 0;
}
