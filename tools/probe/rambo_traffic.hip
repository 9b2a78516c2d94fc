// Traffic-only stand-in for k_rambo<4, 0>: per sample 64 B read (8 doubles of a row-major (N, 8) matrix, four 16-byte loads per lane...
// the 2 -> 4 map takes 3n - 4 = 8 random numbers), 192 B written (24 doubles of a row-major (N, 24) momenta matrix, each wave streams its own
// 64 rows = 12 KiB as twelve 1 KiB wave instructions of 16 bytes per lane) and 8 B written (weight) -- 264 B per sample, no arithmetic.
// What this reaches is the ceiling of the map's HBM pattern on this machine; k_rambo is priced against it in DESIGN.md.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/rambo_traffic.hip -o /tmp/rambo_traffic && /tmp/rambo_traffic [N] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

template <int WRITE_M, int READ>
__global__ __launch_bounds__(128) void k_traffic(const double* __restrict__ r, long N, double* __restrict__ mom, double* __restrict__ w) {
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const long ntiles = N / 128;
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long i = t * 128 + tid;
    d2 a = {1.0, 2.0}, b = a, c = a, d = a;
    if (READ) {
      const d2* row = reinterpret_cast<const d2*>(r + i * 8);
      a = row[0]; b = row[1]; c = row[2]; d = row[3];
    }
    const double s = (a.x + b.x) + (c.x + d.x) + (a.y + b.y) + (c.y + d.y);
    if (WRITE_M) {
      double* dst = mom + (t * 128 + wv * 64) * 24;
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        d2 v = {s + u, s - u};
        *reinterpret_cast<d2*>(dst + 2 * ln + 128 * u) = v;
      }
    }
    w[i] = s;
  }
}

// float4-style copy of the same byte count (the 6.3 TB/s reference point)
__global__ __launch_bounds__(256) void k_copy(const d2* __restrict__ src, d2* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

int main(int argc, char** argv) {
  const long N = argc > 1 ? atol(argv[1]) : 1000000;
  const int reps = argc > 2 ? atoi(argv[2]) : 40;
  const int SETS = 4;                                  // rotating buffer sets: 4 x 264 MB > the 256 MB Infinity Cache
  std::vector<double*> r(SETS), m(SETS), w(SETS);
  for (int s = 0; s < SETS; ++s) {
    hipMalloc(&r[s], N * 8 * 8);
    hipMalloc(&m[s], N * 24 * 8);
    hipMalloc(&w[s], N * 8);
    hipMemset(r[s], 0, N * 8 * 8);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  auto run = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 5; ++i) launch(i % SETS);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch(i % SETS);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-44s %8.2f us per launch   %7.1f GB/s\n", name, us, bytes / us * 1e-3);
  };
  for (int mult : {8, 12, 16, 24}) {
    const int grid = cus * mult;
    char nm[96];
    snprintf(nm, sizeof nm, "read 64 + write 192 + 8 B/sample, grid %dxCU", mult);
    run(nm, [&](int s) { hipLaunchKernelGGL((k_traffic<1, 1>), dim3(grid), dim3(128), 0, 0, r[s], N, m[s], w[s]); }, 264.0 * N);
  }
  const int grid = cus * 12;
  run("write 192 + 8 B/sample only", [&](int s) { hipLaunchKernelGGL((k_traffic<1, 0>), dim3(grid), dim3(128), 0, 0, r[s], N, m[s], w[s]); }, 200.0 * N);
  run("read 64 + write 8 B/sample only", [&](int s) { hipLaunchKernelGGL((k_traffic<0, 1>), dim3(grid), dim3(128), 0, 0, r[s], N, m[s], w[s]); }, 72.0 * N);
  run("copy, 132 B read + 132 B written per sample", [&](int s) {
    hipLaunchKernelGGL(k_copy, dim3(cus * 8), dim3(256), 0, 0, (const d2*)m[s], (d2*)m[(s + 1) % SETS], N * 132 / 16); }, 264.0 * N);
  return 0;
}
