// How fast can a per-lane-sample fp64 GEMV (y = W x, W wave-uniform) run on gfx950 when the weights are fed
// (a) from LDS with wave-uniform (broadcast) ds_read, (b) from scalar loads (constant address space), against (c) the
// same FMA count with the weights held in registers?  ROWS x 32 weights, x[32] per lane, 256 threads per block.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/valu_gemv_feed.hip -o tools/probe/valu_gemv_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef const double __attribute__((address_space(4))) * cptr;

template <int MODE, int ROWS>
__global__ __launch_bounds__(256) void k(const double* __restrict__ Wg, double* out, int iters) {
  extern __shared__ double Wl[];
  if (MODE == 0)
    for (int i = threadIdx.x; i < ROWS * 32; i += 256) Wl[i] = Wg[i];
  __syncthreads();
  cptr Wc = (cptr)Wg;
  double x[32];
  for (int i = 0; i < 32; ++i) x[i] = 1.0 + 1e-3 * ((threadIdx.x + i) & 7);
  double sink = 0;
  for (int it = 0; it < iters; ++it) {
    for (int j0 = 0; j0 < ROWS; j0 += 8) {
      double y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = 0;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          double w;
          if (MODE == 0) w = Wl[(j0 + j) * 32 + i];
          else if (MODE == 1) w = Wc[(j0 + j) * 32 + i];
          else w = 1.0000001;
          y[j] = __builtin_fma(w, x[i], y[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) sink += y[j];
    }
    x[it & 31] += sink * 1e-30;
  }
  out[blockIdx.x * 256 + threadIdx.x] = sink;
}

template <int MODE, int ROWS>
static float run(const double* W, double* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t lds = MODE == 0 ? sizeof(double) * ROWS * 32 : 0;
  hipLaunchKernelGGL((k<MODE, ROWS>), dim3(blocks), dim3(256), lds, 0, W, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, ROWS>), dim3(blocks), dim3(256), lds, 0, W, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int ROWS>
static void sweep(const double* W, double* out) {
  const int iters = 400 * 264 / ROWS;
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * wps;
    const float a = run<0, ROWS>(W, out, blocks, iters), b = run<1, ROWS>(W, out, blocks, iters), c = run<2, ROWS>(W, out, blocks, iters);
    const double fma_per_wave = double(iters) * ROWS * 32;
    printf("rows %3d, %d wave(s)/SIMD: lds %.3f ms, s_load %.3f ms, registers %.3f ms   (ideal at 4 cycles/FMA, 2.4 GHz: %.3f ms)\n", ROWS, wps, a,
           b, c, wps * fma_per_wave * 4 / 2.4e6);
  }
}

int main() {
  double *W, *out;
  std::vector<double> h(264 * 32, 1.0000001);
  hipMalloc(&W, sizeof(double) * h.size());
  hipMemcpy(W, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
  hipMalloc(&out, sizeof(double) * 256 * 1024);
  sweep<32>(W, out);
  sweep<264>(W, out);
  return 0;
}
