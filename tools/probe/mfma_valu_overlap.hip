// Which instruction classes share issue / datapath time with v_mfma_f64_16x16x4_f64 on gfx950?
//
// One workgroup of 512 threads per CU (256 workgroups): SIMD s holds wave s ("A" role) and wave s + 4 ("B" role).  Each role runs one
// instruction stream (or exits at once), so every pairing is measured for two DIFFERENT waves of a SIMD, and -- stream MIX_* -- for one
// wave that interleaves the MFMAs with the other class in program order.  If t(A+B) ~ max(t(A), t(B)) the two classes overlap; if
// t(A+B) ~ t(A) + t(B) they serialise.  Work per stream and iteration: 4 MFMAs (4 x 64 cycles of the f64 matrix pipe) resp. 64 vector
// instructions (fp64: 64 x 4 cycles; fp32 / int32: 64 x 1..4 cycles).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu_overlap.hip -o tools/probe/mfma_valu_overlap && tools/probe/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

enum Stream { NONE = 0, MFMA = 1, F64 = 2, F32 = 3, I32 = 4, LDSR = 5, MIX_F64 = 6, MIX_F32 = 7, MIX_I32 = 8, SEL64 = 9, MIX_LDS = 10 };
static const char* kName[] = {"-", "mfma_f64", "valu_f64_fma", "valu_f32_fma", "valu_i32", "lds_read_b64", "mfma+f64 (one wave)",
                              "mfma+f32 (one wave)", "mfma+i32 (one wave)", "valu_cndmask_b64", "mfma+lds (one wave)"};

template <int S>
__device__ __forceinline__ double stream(int iters, double seed, double* lds) {
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = d4{seed, seed, seed, seed};
  const double a = seed + threadIdx.x * 1e-9, b = 1.0000001;
  double v[16];
  float f[16];
  unsigned u[16];
  for (int i = 0; i < 16; ++i) v[i] = seed * (i + 1), f[i] = (float)seed * (i + 1), u[i] = (unsigned)(threadIdx.x * 31 + i);
  const float fb = 1.0000001f, fa = (float)a;
  const unsigned ua = threadIdx.x | 1u;
  double l = 0.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (S == MFMA || S == MIX_F64 || S == MIX_F32 || S == MIX_I32 || S == MIX_LDS) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
      if (S == F64 || S == MIX_F64) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], b, a);
      }
      if (S == F32 || S == MIX_F32) {
#pragma unroll
        for (int i = 0; i < 16; ++i) f[i] = __builtin_fmaf(f[i], fb, fa);
      }
      if (S == I32 || S == MIX_I32) {
#pragma unroll
        for (int i = 0; i < 16; ++i) u[i] = (u[i] ^ ua) + (u[i] >> 3);     // v_xor + v_lshrrev + v_add (or v_lshl_add): 2-3 int ops
      }
      if (S == SEL64) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (u[i] + it) & 1 ? v[(i + 1) & 15] : v[i];   // v_cndmask pairs + an int op
      }
      if (S == LDSR || S == MIX_LDS) {          // 16 ds_read_b64 with NO vector instruction behind them (results discarded by inline asm)
        const unsigned ad = (unsigned)(size_t)lds + (threadIdx.x & 63) * 8;
        double t0, t1, t2, t3;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n"
                       : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3)
                       : "v"(ad));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (it == iters + 5) l += t0 + t1 + t2 + t3;
      }
    }
  }
  double s = l;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i] + (double)f[i] + (double)u[i];
  return s;
}

template <int SA, int SB>
__global__ __launch_bounds__(512) void k(double* out, int iters, double seed) {
  __shared__ double lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = seed * i;
  __syncthreads();
  const bool roleA = (threadIdx.x >> 6) < 4;
  double s;
  if (roleA) {
    if (SA == NONE) return;
    s = stream<SA>(iters, seed, lds);
  } else {
    if (SB == NONE) return;
    s = stream<SB>(iters, seed, lds);
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int SA, int SB>
static float run(double* out, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SA, SB>), dim3(256), dim3(512), 0, 0, out, iters, 1.0);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SA, SB>), dim3(256), dim3(512), 0, 0, out, iters, 1.0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best;
}

template <int SA, int SB>
static void pair(double* out, int iters) {
  const float a = run<SA, NONE>(out, iters), b = run<NONE, SB>(out, iters), c = run<SA, SB>(out, iters);
  const float mx = a > b ? a : b;
  printf("A = %-20s B = %-20s | A alone %6.3f ms | B alone %6.3f ms | both %6.3f ms | sum %6.3f | max %6.3f | overlap %4.0f %%\n", kName[SA], kName[SB], a, b, c,
         a + b, mx, 100.0 * (a + b - c) / (a + b - mx));
}

int main() {
  double* out;
  (void)hipMalloc(&out, sizeof(double) * 512 * 256);
  const int iters = 20000;
  printf("gfx950, 256 workgroups x 512 threads (one per CU): wave s = role A, wave s + 4 = role B on SIMD s; %d iterations, best of 3\n", iters);
  printf("ideal: 4 MFMA f64 16x16x4 = 256 cycles / iteration -> %.3f ms at 2.4 GHz; 64 fp64 FMA = 256 cycles -> the same\n", iters * 256 / 2.4e6);
  printf("-- two different waves on one SIMD\n");
  pair<MFMA, MFMA>(out, iters);
  pair<F64, F64>(out, iters);
  pair<MFMA, F64>(out, iters);
  pair<MFMA, F32>(out, iters);
  pair<MFMA, I32>(out, iters);
  pair<MFMA, SEL64>(out, iters);
  pair<MFMA, LDSR>(out, iters);
  pair<F64, F32>(out, iters);
  pair<F64, I32>(out, iters);
  pair<F64, LDSR>(out, iters);
  printf("-- one wave interleaving both classes in program order (A only), and two such waves per SIMD\n");
  pair<MIX_F64, NONE>(out, iters);
  pair<MIX_F32, NONE>(out, iters);
  pair<MIX_I32, NONE>(out, iters);
  pair<MIX_LDS, NONE>(out, iters);
  pair<MIX_F64, MIX_F64>(out, iters);
  pair<MIX_F32, MIX_F32>(out, iters);
  pair<MIX_I32, MIX_I32>(out, iters);
  return 0;
}
