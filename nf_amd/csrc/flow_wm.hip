// One translation unit per conditioner register width: compiled with -DNIS_WM={4,8,16,32,64}
// so the five widths build in parallel.  Instantiates the statistics kernel and every
// (cell kind, bin-array size) variant of the apply kernel for that width.
#include "flow_kernels.h"

#ifndef NIS_WM
#error "compile with -DNIS_WM=<4|8|16|32|64>"
#endif

template <>
hipError_t launch_stats<NIS_WM>(const CellK& c, const double* planes, long ld, long N, int layer, int grid, hipStream_t s) {
  constexpr int TPB = nis_stats_tpb(NIS_WM);
  hipLaunchKernelGGL((k_bn_stats<NIS_WM, TPB>), dim3(grid), dim3(TPB), 0, s, c, planes, ld, N, layer);
  return hipGetLastError();
}

#define NIS_APPLY(KIND, KMV) \
  hipLaunchKernelGGL((k_cell_apply<KIND, NIS_WM, KMV>), dim3(grid), dim3(NIS_TPB), 0, s, c, planes, ld, N, bn_train, bins, fz)

template <>
hipError_t launch_apply<NIS_WM>(const CellK& c, double* planes, long ld, long N, int bn_train, int km, int* bins, const FuseF& fz,
                                int grid, hipStream_t s) {
  if (c.kind == NIS_PWLIN) {
    NIS_APPLY(NIS_PWLIN, 4);
  } else {
    switch (km) {
      case 4: NIS_APPLY(NIS_PWQUAD, 4); break;
      case 8: NIS_APPLY(NIS_PWQUAD, 8); break;
      case 16: NIS_APPLY(NIS_PWQUAD, 16); break;
      case 32: NIS_APPLY(NIS_PWQUAD, 32); break;
      default: return hipErrorInvalidValue;
    }
  }
  return hipGetLastError();
}
