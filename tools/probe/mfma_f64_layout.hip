// Probe: lane -> element maps of v_mfma_f64_16x16x4_f64 on gfx950 (A = arbitrary, B asymmetric).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* C) {   // A 16x4, B 4x16 row-major, C 16x16
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];       // A[i=l&15][k=l>>4]
  const double b = B[(l >> 4) * 16 + (l & 15)];      // B[k=l>>4][j=l&15]
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[(l * 4 + r)] = c[r];
}
int main() {
  double hA[64], hB[64], hC[256], ref[256];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1 + i * 0.5 + k * 7;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = 3 + k * 11 + j * j * 0.25;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  double *dA, *dB, *dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dC);
  hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  int okA = 1, okB = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const double v = hC[l * 4 + r];
    const int col = l & 15;
    if (v != ref[((l >> 4) + 4 * r) * 16 + col]) okA = 0;       // row = (lane>>4) + 4*reg
    if (v != ref[((l >> 4) * 4 + r) * 16 + col]) okB = 0;       // row = 4*(lane>>4) + reg
  }
  printf("row=(lane>>4)+4*reg: %s ; row=4*(lane>>4)+reg: %s\n", okA ? "MATCH" : "no", okB ? "MATCH" : "no");
  return 0;
}
