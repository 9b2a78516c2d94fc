// One translation unit per conditioner register width: compiled with -DNIS_WM={4,8,16,32,64}
// so the five widths build in parallel.  Instantiates the statistics kernel and every
// (cell kind, bin-array size) variant of the apply kernel for that width.
#include "flow_kernels.h"

#ifndef NIS_WM
#error "compile with -DNIS_WM=<4|8|16|32|64>"
#endif

template <>
hipError_t launch_stats<NIS_WM>(const CellK& c, const double* planes, long ld, long N, int layer, int grid, hipStream_t s) {
  hipLaunchKernelGGL(k_bn_stats<NIS_WM>, dim3(grid), dim3(NIS_TPB), 0, s, c, planes, ld, N, layer);
  return hipGetLastError();
}

template <>
hipError_t launch_apply<NIS_WM>(const CellK& c, double* planes, long ld, long N, int bn_train, int km, int* bins, int grid,
                                hipStream_t s) {
  if (c.kind == NIS_PWLIN) {
    hipLaunchKernelGGL((k_cell_apply<NIS_PWLIN, NIS_WM, 4>), dim3(grid), dim3(NIS_TPB), 0, s, c, planes, ld, N, bn_train, bins);
  } else {
    switch (km) {
      case 4:
        hipLaunchKernelGGL((k_cell_apply<NIS_PWQUAD, NIS_WM, 4>), dim3(grid), dim3(NIS_TPB), 0, s, c, planes, ld, N, bn_train, bins);
        break;
      case 8:
        hipLaunchKernelGGL((k_cell_apply<NIS_PWQUAD, NIS_WM, 8>), dim3(grid), dim3(NIS_TPB), 0, s, c, planes, ld, N, bn_train, bins);
        break;
      case 16:
        hipLaunchKernelGGL((k_cell_apply<NIS_PWQUAD, NIS_WM, 16>), dim3(grid), dim3(NIS_TPB), 0, s, c, planes, ld, N, bn_train, bins);
        break;
      case 32:
        hipLaunchKernelGGL((k_cell_apply<NIS_PWQUAD, NIS_WM, 32>), dim3(grid), dim3(NIS_TPB), 0, s, c, planes, ld, N, bn_train, bins);
        break;
      default:
        return hipErrorInvalidValue;
    }
  }
  return hipGetLastError();
}
