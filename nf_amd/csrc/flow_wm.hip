// One translation unit per conditioner register width: compiled with -DNIS_WM={4,8,16,32,64}
// so the five widths build in parallel.  Instantiates the statistics kernel and every
// (cell kind, bin-array size) variant of the apply kernel for that width.
#include <cstdio>
#include <cstdlib>

#if !defined(NIS_EXP_BITS) && defined(NIS_WM) && NIS_WM <= 8
#define NIS_EXP_BITS 10         // narrow conditioners: VALU-issue bound, LDS to spare -> 1024-entry exp table + cubic (nis_common.h)
#elif !defined(NIS_EXP_BITS)
#define NIS_EXP_BITS 8          // matrix-core units: 256 entries + quartic fit beside the LDS image of the apply kernel
#endif
#include "flow_kernels.h"

static bool nis_env_int_wm(const char* name) {
  const char* v = getenv(name);
  return v && *v && atoi(v) != 0;
}

#ifndef NIS_WM
#error "compile with -DNIS_WM=<4|8|16|32|64>"
#endif

template <>
hipError_t launch_stats<NIS_WM>(const CellK& c, const double* planes, long ld, long N, int layer, int grid, hipStream_t s) {
#if NIS_WM >= 16
  if (layer == 0 && c.dcov) {                      // BN_0 records with the pass-through co-moments (BN_1 is derived from them)
    hipLaunchKernelGGL((k_bn_stats0_cov<NIS_WM, 256>), dim3(grid, c.Gy), dim3(256), 0, s, c, planes, ld, N);
    return hipGetLastError();
  }
#endif
  constexpr int TPB = nis_stats_tpb(NIS_WM);
#if NIS_WM == 4
  if (c.wx == 3) {                       // every layer at most 3 wide (README / C1 / C2 nets): 3x3 blocks instead of 4x4
    constexpr bool hoisted = true;
    const bool rows = c.wrows && c.x_wcol >= 0;        // pass-through column still lives in the float32 rows of w
    if (rows && !(c.P == 1 && c.fold && hoisted && layer >= 2 && layer <= 5)) return hipErrorInvalidValue;
    if (rows && c.wplanar) {
      switch (layer) {
        case 2: hipLaunchKernelGGL((k_bn_stats_fold<NIS_WM, TPB, 3, 2, true, true>), dim3(grid, c.Gy), dim3(TPB), 0, s, c, planes, ld, N); break;
        case 3: hipLaunchKernelGGL((k_bn_stats_fold<NIS_WM, TPB, 3, 3, true, true>), dim3(grid, c.Gy), dim3(TPB), 0, s, c, planes, ld, N); break;
        default: return hipErrorInvalidValue;       // planar draws exist for the three-hidden-layer SPEC chain
      }
    } else if (rows) {
      switch (layer) {
        case 2: hipLaunchKernelGGL((k_bn_stats_fold<NIS_WM, TPB, 3, 2, true>), dim3(grid, c.Gy), dim3(TPB), 0, s, c, planes, ld, N); break;
        case 3: hipLaunchKernelGGL((k_bn_stats_fold<NIS_WM, TPB, 3, 3, true>), dim3(grid, c.Gy), dim3(TPB), 0, s, c, planes, ld, N); break;
        case 4: hipLaunchKernelGGL((k_bn_stats_fold<NIS_WM, TPB, 3, 4, true>), dim3(grid, c.Gy), dim3(TPB), 0, s, c, planes, ld, N); break;
        default: hipLaunchKernelGGL((k_bn_stats_fold<NIS_WM, TPB, 3, 5, true>), dim3(grid, c.Gy), dim3(TPB), 0, s, c, planes, ld, N); break;
      }
    } else if (c.P == 1 && c.fold && hoisted && layer >= 2 && layer <= 5) {     // folded chain, coefficients in registers
      switch (layer) {
        case 2: hipLaunchKernelGGL((k_bn_stats_fold<NIS_WM, TPB, 3, 2>), dim3(grid, c.Gy), dim3(TPB), 0, s, c, planes, ld, N); break;
        case 3: hipLaunchKernelGGL((k_bn_stats_fold<NIS_WM, TPB, 3, 3>), dim3(grid, c.Gy), dim3(TPB), 0, s, c, planes, ld, N); break;
        case 4: hipLaunchKernelGGL((k_bn_stats_fold<NIS_WM, TPB, 3, 4>), dim3(grid, c.Gy), dim3(TPB), 0, s, c, planes, ld, N); break;
        default: hipLaunchKernelGGL((k_bn_stats_fold<NIS_WM, TPB, 3, 5>), dim3(grid, c.Gy), dim3(TPB), 0, s, c, planes, ld, N); break;
      }
    } else if (c.P == 1)
      hipLaunchKernelGGL((k_bn_stats<NIS_WM, TPB, 3, true>), dim3(grid, c.Gy), dim3(TPB), 0, s, c, planes, ld, N, layer);
    else
      hipLaunchKernelGGL((k_bn_stats<NIS_WM, TPB, 3>), dim3(grid, c.Gy), dim3(TPB), 0, s, c, planes, ld, N, layer);
    return hipGetLastError();
  }
#endif
  if (c.wrows && c.x_wcol >= 0) return hipErrorInvalidValue;          // row input exists for the folded 3-wide chain only
  hipLaunchKernelGGL((k_bn_stats<NIS_WM, TPB>), dim3(grid, c.Gy), dim3(TPB), 0, s, c, planes, ld, N, layer);
  return hipGetLastError();
}

template <typename K>
static int stats_occ_of(K kern) {
  constexpr int TPB = nis_stats_tpb(NIS_WM);
  hipFuncAttributes at;
  int n = 1;
  if (hipFuncGetAttributes(&at, reinterpret_cast<const void*>(kern)) == hipSuccess && at.numRegs > 0) {
    const int alloc = (at.numRegs + 7) / 8 * 8;
    int waves = 512 / alloc;
    waves = waves < 1 ? 1 : waves > 8 ? 8 : waves;
    n = (waves * 4 * 64 + TPB - 1) / TPB;     // waves/SIMD * 4 SIMDs * 64 lanes / workgroup size, rounded up (measured: 2 beats 1)
    const int by_lds = at.sharedSizeBytes > 0 ? (int)(160 * 1024 / at.sharedSizeBytes) : 8;
    n = n < by_lds ? n : by_lds;
    if (n < 1) n = 1;
    if (getenv("NIS_DEBUG_OCC")) fprintf(stderr, "[nis] k_bn_stats<%d>: numRegs %d lds %zu -> %d workgroups/CU\n", NIS_WM, at.numRegs, (size_t)at.sharedSizeBytes, n);
  }
  return n;
}

template <>
int stats_blocks_per_cu<NIS_WM>(int wx, bool folded) {
  static int v[2] = {-1, -1};
  constexpr int TPB = nis_stats_tpb(NIS_WM);
#if NIS_WM == 4
  // k_bn_stats_fold keeps NIS_STATS_U loads in flight per wave and pays its prologue once per workgroup: one 512-thread
  // workgroup per CU measured best (C2 bench: 20.5e9 samples/s at 1, 20.2 at 2, 19.5 at 4)
  if (wx == 3 && folded) return 1;
  if (wx == 3) {
    if (v[1] < 0) v[1] = stats_occ_of(k_bn_stats<NIS_WM, TPB, 3>);
    return v[1];
  }
#endif
  if (v[0] < 0) v[0] = stats_occ_of(k_bn_stats<NIS_WM, TPB>);
  return v[0];
}

// resident 256-thread workgroups per CU from the kernel's register allocation (the occupancy API
// under-reports these kernels: measured optimum 5/CU where it answers 3): 512 VGPRs per SIMD lane, granule 8.
template <typename K>
static int occ_of(K kern) {
  hipFuncAttributes at;
  if (hipFuncGetAttributes(&at, reinterpret_cast<const void*>(kern)) != hipSuccess || at.numRegs < 1) return 4;
  const int alloc = (at.numRegs + 7) / 8 * 8;
  int waves = 512 / alloc;                       // waves per SIMD; a 256-thread workgroup puts one wave on each SIMD
  waves = waves < 1 ? 1 : waves > 8 ? 8 : waves;
  int by_lds = at.sharedSizeBytes > 0 ? (int)(160 * 1024 / at.sharedSizeBytes) : 8;
  int n = waves < by_lds ? waves : (by_lds < 1 ? 1 : by_lds);
  // These kernels are fp64-issue bound: a CU works off its workgroups at the same rate with 2 or with 6 resident waves
  // per SIMD, so MORE and smaller workgroups than fit at once only improve the balance between CUs (round 1: 5 per CU 212 us vs
  // 277 us at the 3 that are resident; round-4 sweep of 5 / 6 / 9 / 12 / 15, profiles/r04_c2_attempts.txt: 6 -- two full rounds
  // of the three resident workgroups -- is 1-2 % ahead of 5, 12 and 15 fall behind).
#ifndef NIS_APPLY_MIN_BPC
#define NIS_APPLY_MIN_BPC 6
#endif
  if (n < NIS_APPLY_MIN_BPC) n = NIS_APPLY_MIN_BPC;
  if (getenv("NIS_DEBUG_OCC")) fprintf(stderr, "[nis] apply kernel (WM %d): numRegs %d lds %zu -> %d workgroups/CU\n", NIS_WM, at.numRegs, (size_t)at.sharedSizeBytes, n);
  return n;
}

template <>
int apply_blocks_per_cu<NIS_WM>(int kind, int km, int wx) {
  static int cache[10] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
  const bool exact = km < 0;
  const int kk = exact ? -km : km;
  if (kind == NIS_PWQUAD && kk > 32) {                  // scratch-array variant (33 .. NIS_MAX_BINS bins)
    static int big = -1;
    if (big < 0) big = occ_of(k_cell_apply<NIS_PWQUAD, NIS_WM, NIS_KM_BIG, false>);
    return big;
  }
#if NIS_WM == 4
  if (wx == 3 && kk <= 16 && kind != NIS_AFFINE) {           // the 3-wide instantiations have their own register budget
    static int c3[7] = {-1, -1, -1, -1, -1, -1, -1};
    const int s3 = kind == NIS_PWLIN ? 0 : (kk == 4 ? 1 : kk == 8 ? 2 : 3) + (exact ? 3 : 0);
    if (c3[s3] < 0) {
      switch (s3) {
        case 0: c3[s3] = occ_of(k_cell_apply<NIS_PWLIN, NIS_WM, 4, false, 3>); break;
        case 1: c3[s3] = occ_of(k_cell_apply<NIS_PWQUAD, NIS_WM, 4, false, 3>); break;
        case 2: c3[s3] = occ_of(k_cell_apply<NIS_PWQUAD, NIS_WM, 8, false, 3>); break;
        case 3: c3[s3] = occ_of(k_cell_apply<NIS_PWQUAD, NIS_WM, 16, false, 3>); break;
        case 4: c3[s3] = occ_of(k_cell_apply<NIS_PWQUAD, NIS_WM, 4, true, 3>); break;
        case 5: c3[s3] = occ_of(k_cell_apply<NIS_PWQUAD, NIS_WM, 8, true, 3>); break;
        default: c3[s3] = occ_of(k_cell_apply<NIS_PWQUAD, NIS_WM, 16, true, 3>); break;
      }
    }
    return c3[s3];
  }
#endif
  const int slot = kind == NIS_PWLIN ? 0 : kind == NIS_AFFINE ? 9 : (kk == 4 ? 1 : kk == 8 ? 2 : kk == 16 ? 3 : 4) + (exact ? 4 : 0);
  if (cache[slot] < 0) {
    switch (slot) {
      case 0: cache[slot] = occ_of(k_cell_apply<NIS_PWLIN, NIS_WM, 4, false>); break;
      case 1: cache[slot] = occ_of(k_cell_apply<NIS_PWQUAD, NIS_WM, 4, false>); break;
      case 2: cache[slot] = occ_of(k_cell_apply<NIS_PWQUAD, NIS_WM, 8, false>); break;
      case 3: cache[slot] = occ_of(k_cell_apply<NIS_PWQUAD, NIS_WM, 16, false>); break;
      case 4: cache[slot] = occ_of(k_cell_apply<NIS_PWQUAD, NIS_WM, 32, false>); break;
      case 5: cache[slot] = occ_of(k_cell_apply<NIS_PWQUAD, NIS_WM, 4, true>); break;
      case 6: cache[slot] = occ_of(k_cell_apply<NIS_PWQUAD, NIS_WM, 8, true>); break;
      case 7: cache[slot] = occ_of(k_cell_apply<NIS_PWQUAD, NIS_WM, 16, true>); break;
      case 9: cache[slot] = occ_of(k_cell_apply<NIS_AFFINE, NIS_WM, 4, false>); break;
      default: cache[slot] = occ_of(k_cell_apply<NIS_PWQUAD, NIS_WM, 32, true>); break;
    }
  }
  return cache[slot];
}

#define NIS_APPLY(KIND, KMV, EX) \
  hipLaunchKernelGGL((k_cell_apply<KIND, NIS_WM, KMV, EX>), dim3(grid, c.Gy), dim3(NIS_TPB), 0, s, c, planes, ld, N, bn_train, bins, fz)

// km < 0 selects the K == KM specialisation (no per-bin guards)
template <>
hipError_t launch_apply<NIS_WM>(const CellK& c, double* planes, long ld, long N, int bn_train, int km, int* bins, const FuseF& fz,
                                int grid, hipStream_t s) {
#if NIS_WM == 4
#define NIS_APPLY3(KIND, KMV, EX) \
  hipLaunchKernelGGL((k_cell_apply<KIND, NIS_WM, KMV, EX, 3>), dim3(grid, c.Gy), dim3(NIS_TPB), 0, s, c, planes, ld, N, bn_train, bins, fz)
#define NIS_APPLY3S(KMV, SP)                                                                                                       \
  do {                                                                                                                             \
    if (rows && c.wplanar && !bins && KMV == 4 && c.first)   /* in-kernel draws: float32 column planes */                           \
      hipLaunchKernelGGL((k_cell_apply<NIS_PWQUAD, NIS_WM, 4, true, 3, SP, true, true, false, true>), dim3(grid, c.Gy), dim3(NIS_TPB), 0, s, \
                         c, planes, ld, N, bn_train, bins, fz);                                                                    \
    else if (rows && c.wplanar && !bins && KMV == 4)                                                                               \
      hipLaunchKernelGGL((k_cell_apply<NIS_PWQUAD, NIS_WM, 4, true, 3, SP, true, false, false, true>), dim3(grid, c.Gy), dim3(NIS_TPB), 0, s, \
                         c, planes, ld, N, bn_train, bins, fz);                                                                    \
    else if (rows && c.wplanar)                                                                                                    \
      return hipErrorInvalidValue;                                                                                                 \
    else if (bins && rows && KMV == 4 && c.first)     /* test instantiation: SPEC kernels that also write the bin indices */        \
      hipLaunchKernelGGL((k_cell_apply<NIS_PWQUAD, NIS_WM, 4, true, 3, SP, true, true, true>), dim3(grid, c.Gy), dim3(NIS_TPB), 0, s, c, \
                         planes, ld, N, bn_train, bins, fz);                                                                       \
    else if (bins && rows && KMV == 4)                                                                                             \
      hipLaunchKernelGGL((k_cell_apply<NIS_PWQUAD, NIS_WM, 4, true, 3, SP, true, false, true>), dim3(grid, c.Gy), dim3(NIS_TPB), 0, s, c, \
                         planes, ld, N, bn_train, bins, fz);                                                                       \
    else if (bins)                                                                                                                 \
      return hipErrorInvalidValue;                                                                                                 \
    else if (rows && c.first)                                                                                                      \
      hipLaunchKernelGGL((k_cell_apply<NIS_PWQUAD, NIS_WM, KMV, true, 3, SP, true, true>), dim3(grid, c.Gy), dim3(NIS_TPB), 0, s, c,   \
                         planes, ld, N, bn_train, bins, fz);                                                                       \
    else if (rows)                                                                                                                 \
      hipLaunchKernelGGL((k_cell_apply<NIS_PWQUAD, NIS_WM, KMV, true, 3, SP, true, false>), dim3(grid, c.Gy), dim3(NIS_TPB), 0, s, c,  \
                         planes, ld, N, bn_train, bins, fz);                                                                       \
    else                                                                                                                           \
      hipLaunchKernelGGL((k_cell_apply<NIS_PWQUAD, NIS_WM, KMV, true, 3, SP>), dim3(grid, c.Gy), dim3(NIS_TPB), 0, s, c, planes, ld, \
                         N, bn_train, bins, fz);                                                                                   \
  } while (0)
  const bool rows = c.wrows && c.xb_wcol >= 0;           // transformed input (and maybe the pass-through one) from the rows of w
  if (c.wrows && c.x_wcol >= 0 && !rows) return hipErrorInvalidValue;
  if (rows && (c.first != 0) != (c.x_wcol >= 0)) return hipErrorInvalidValue;   // row kernels: first cell <=> both inputs in w
  if (c.wx == 3 && c.kind == NIS_PWQUAD && km < 0 && c.P == 1 && c.T == 1 && c.L == 3 && c.fold && !rows && !bins && !fz.fres && fz.kind < 0 &&
      fz.write_planes && !nis_env_int_wm("NIS_NO_SPEC_HIST")) {
    // the same two cells in the training forward (history: outputs to the next plane set): 3 = next-cell rider, 4 = last cell; 4 also
    // in place (out_off == 0): the last cell of an integrate() set that keeps X and J for a caller-side integrand (NIS_F_NONE)
    const int spec = (c.next_P == 1 && c.next_src[0] == 1) ? (c.out_off != 0 ? 3 : 0) : c.next_P == 0 ? 4 : 0;
#define NIS_APPLY3H(KMV, SP) \
  hipLaunchKernelGGL((k_cell_apply<NIS_PWQUAD, NIS_WM, KMV, true, 3, SP>), dim3(grid, c.Gy), dim3(NIS_TPB), 0, s, c, planes, ld, N, bn_train, bins, fz)
    if (spec == 3) {
      switch (km) {
        case -4: NIS_APPLY3H(4, 3); return hipGetLastError();
        case -8: NIS_APPLY3H(8, 3); return hipGetLastError();
        case -16: NIS_APPLY3H(16, 3); return hipGetLastError();
        default: break;
      }
    } else if (spec == 4) {
      switch (km) {
        case -4: NIS_APPLY3H(4, 4); return hipGetLastError();
        case -8: NIS_APPLY3H(8, 4); return hipGetLastError();
        case -16: NIS_APPLY3H(16, 4); return hipGetLastError();
        default: break;
      }
    }
#undef NIS_APPLY3H
  }
  if (c.wx == 3 && c.kind == NIS_PWQUAD && km < 0 && c.P == 1 && c.T == 1 && c.L == 3 && c.fold && c.out_off == 0 && (!bins || (rows && km == -4)) && !fz.fres) {
    // the two cells of a d = 2 chain: every wave-uniform switch resolved at compile time (k_cell_apply, SPEC)
    // 1: the next cell passes THIS cell's output through (next_src[0] == 1: the alternating roll of a d = 2 chain);
    // 2: the camel's two arguments are this cell's two columns, in either order (the integrand is symmetric)
    const int spec = (c.next_P == 1 && c.next_src[0] == 1 && fz.kind < 0 && fz.write_planes) ? 1
                     : (c.next_P == 0 && fz.kind == NIS_F_CAMEL && !fz.write_planes && (fz.loc0 == 0 || fz.loc0 == 1) &&
                        fz.loc1 == 1 - fz.loc0) ? 2 : 0;
    if (spec == 1) {
      switch (km) {
        case -4: NIS_APPLY3S(4, 1); return hipGetLastError();
        case -8: NIS_APPLY3S(8, 1); return hipGetLastError();
        case -16: NIS_APPLY3S(16, 1); return hipGetLastError();
        default: break;
      }
    } else if (spec == 2) {
      switch (km) {
        case -4: NIS_APPLY3S(4, 2); return hipGetLastError();
        case -8: NIS_APPLY3S(8, 2); return hipGetLastError();
        case -16: NIS_APPLY3S(16, 2); return hipGetLastError();
        default: break;
      }
    }
  }
  if (rows) return hipErrorInvalidValue;                 // row input exists on the SPEC kernels only (api.hip: direct_rows_ok)
  if (c.wx == 3) {
    if (c.kind == NIS_PWLIN) {
      NIS_APPLY3(NIS_PWLIN, 4, false);
      return hipGetLastError();
    }
    if (c.kind == NIS_PWQUAD) {
      switch (km) {
        case 4: NIS_APPLY3(NIS_PWQUAD, 4, false); return hipGetLastError();
        case 8: NIS_APPLY3(NIS_PWQUAD, 8, false); return hipGetLastError();
        case 16: NIS_APPLY3(NIS_PWQUAD, 16, false); return hipGetLastError();
        case -4: NIS_APPLY3(NIS_PWQUAD, 4, true); return hipGetLastError();
        case -8: NIS_APPLY3(NIS_PWQUAD, 8, true); return hipGetLastError();
        case -16: NIS_APPLY3(NIS_PWQUAD, 16, true); return hipGetLastError();
        default: break;                   // other bin counts: generic 4-wide kernel
      }
    }
  }
#endif
  if (c.kind == NIS_PWLIN) {
    NIS_APPLY(NIS_PWLIN, 4, false);
  } else if (c.kind == NIS_AFFINE) {
    NIS_APPLY(NIS_AFFINE, 4, false);
  } else {
    switch (km) {
      case 4: NIS_APPLY(NIS_PWQUAD, 4, false); break;
      case 8: NIS_APPLY(NIS_PWQUAD, 8, false); break;
      case 16: NIS_APPLY(NIS_PWQUAD, 16, false); break;
      case 32: NIS_APPLY(NIS_PWQUAD, 32, false); break;
      case NIS_KM_BIG: NIS_APPLY(NIS_PWQUAD, NIS_KM_BIG, false); break;
      case -4: NIS_APPLY(NIS_PWQUAD, 4, true); break;
      case -8: NIS_APPLY(NIS_PWQUAD, 8, true); break;
      case -16: NIS_APPLY(NIS_PWQUAD, 16, true); break;
      case -32: NIS_APPLY(NIS_PWQUAD, 32, true); break;
      default: return hipErrorInvalidValue;
    }
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// matrix-core (MFMA f64) variants for wide conditioners
#if NIS_WM >= 16
#include "flow_mfma.h"

// the cell as the apply kernel sees it: with the relative image (NIS_PACK_REL) a column has 2K rows (2K - 1 computed, one zero row)
static CellK mf_view(const CellK& c_in) {
  CellK c = c_in;
  if (c.regmode == NIS_PACK_REL) {
    c.out = nis_rel_out(c.K);
    c.R = c.T * c.out;
  }
  return c;
}
// per wave: the whole row tiles of one column batch (flow_mfma.h, mf_cb_tiles); at least the prologue's reduction scratch
static size_t mf_ring_bytes_n(const CellK& c, int ncb) {
  const size_t ring = (size_t)(NIS_MF_TPB / 64) * mf_cb_tiles(c.T, c.out, c.R, ncb) * 256 * sizeof(double);
  return std::max<size_t>(ring, 3 * NIS_MF_TPB * sizeof(double));
}
static const size_t MF_STATIC_LDS = 7 * 1024;            // static LDS of k_cell_apply_mfma: 6.8 KB (2 KB of it the exp table)
// columns per bijector batch: four (every lane busy in the bijector phase) unless the image of four columns leaves ONE workgroup
// per CU while two or one column(s) would admit two -- one wave per SIMD hides nothing (C3 ran at 2x its MFMA + VALU time)
static int mf_pick_ncb(const CellK& c) {
  const int waves = (c.kind == NIS_PWLIN && NIS_MF_WAVES_LIN) ? NIS_MF_WAVES_LIN : NIS_MF_WAVES;
  for (int n = NIS_MF_NCB; n >= 1; n >>= 1)
    if (waves * (mf_ring_bytes_n(c, n) + MF_STATIC_LDS) <= 160 * 1024) return n;
  for (int n = NIS_MF_NCB; n >= 1; n >>= 1)                  // no batch size admits two workgroups: the largest that fits at all
    if (mf_ring_bytes_n(c, n) + 24 * 1024 <= 160 * 1024) return n;
  return 1;
}
static size_t mf_ring_bytes(const CellK& c) { return mf_ring_bytes_n(c, mf_pick_ncb(c)); }

template <>
bool mfma_supported<NIS_WM>(const CellK& c_in) {
  const CellK c = mf_view(c_in);
  return mf_ring_bytes(c) + 24 * 1024 <= 160 * 1024 && (c.kind == NIS_PWLIN || c.K <= 32);
}

template <>
int mfma_blocks_per_cu<NIS_WM>(const CellK& c_in) {
  const CellK c = mf_view(c_in);
  const size_t lds = mf_ring_bytes(c) + MF_STATIC_LDS;
  int n = (int)(160 * 1024 / lds);
  const int waves = (c.kind == NIS_PWLIN && NIS_MF_WAVES_LIN) ? NIS_MF_WAVES_LIN : NIS_MF_WAVES;
  return n < 1 ? 1 : n > waves ? waves : n;   // <= 256 VGPR+AGPR per lane: two waves per SIMD when the LDS images fit
}

template <>
hipError_t launch_stats_mfma<NIS_WM>(const CellK& c, const MfmaW& w, const double* planes, long ld, long N, int layer, int grid,
                                     hipStream_t s) {
  if (nis_mf_stats_tpb(NIS_WM, (long long)c.Gy * N) == NIS_MF_STATS_TPB && NIS_MF_STATS_TPB != NIS_MF_TPB)
    hipLaunchKernelGGL((k_bn_stats_mfma<NIS_WM, NIS_WM <= 32 ? NIS_MF_STATS_TPB : NIS_MF_TPB>), dim3(grid, c.Gy), dim3(NIS_MF_STATS_TPB), 0, s, c, planes, ld, N, layer, w.wm_all);
  else
    hipLaunchKernelGGL((k_bn_stats_mfma<NIS_WM, NIS_MF_TPB>), dim3(grid, c.Gy), dim3(NIS_MF_TPB), 0, s, c, planes, ld, N, layer, w.wm_all);
  return hipGetLastError();
}

#define NIS_APPLY_MF(KIND, KMV, EX)                                                                                               \
  hipLaunchKernelGGL((k_cell_apply_mfma<KIND, NIS_WM, KMV, EX>), dim3(grid, c.Gy), dim3(NIS_MF_TPB), mf_ring_bytes(c), s, c, planes, ld, \
                     N, bn_train, bins, w.wm_all, w.wm_last, w.bias_pad, fz)
#define NIS_APPLY_MFF(KIND, KMV)                                                                                                  \
  hipLaunchKernelGGL((k_cell_apply_mfma<KIND, NIS_WM, KMV, true, true>), dim3(grid, c.Gy), dim3(NIS_MF_TPB), mf_ring_bytes(c), s, c, planes, \
                     ld, N, bn_train, bins, w.wm_all, w.wm_last, w.bias_pad, fz)
// PWQuad with K == KM: the relative image (nis_common.h: NIS_MF_REL is a build-time choice, one instantiation per shape)
#define NIS_APPLY_MFQ(KMV, FU)                                                                                                    \
  hipLaunchKernelGGL((k_cell_apply_mfma<NIS_PWQUAD, NIS_WM, KMV, true, FU, NIS_MF_REL != 0>), dim3(grid, c.Gy), dim3(NIS_MF_TPB),    \
                     mf_ring_bytes(c), s, c, planes, ld, N, bn_train, bins, w.wm_all, w.wm_last, w.bias_pad, fz)

template <>
hipError_t launch_apply_mfma<NIS_WM>(const CellK& c_in, const MfmaW& w, double* planes, long ld, long N, int bn_train, int km, int* bins,
                                     const FuseF& fz, int grid, hipStream_t s) {
  CellK c = mf_view(c_in);
  const bool rel = c.regmode == NIS_PACK_REL;
  c.mf_ncb = mf_pick_ncb(c);
  if (rel != (NIS_MF_REL != 0 && c.kind == NIS_PWQUAD && km < 0)) return hipErrorInvalidValue;      // packer and launcher disagree
  if (fz.kind >= 0) {                      // last cell of an integrate step: integrand * J + moment records fused (FUSE instantiations)
    if (!nis_mfma_fusable(c.kind, km) || bins || fz.fres) return hipErrorInvalidValue;
    if (c.kind == NIS_PWLIN) {
      NIS_APPLY_MFF(NIS_PWLIN, 4);
    } else if (c.kind == NIS_AFFINE) {
      NIS_APPLY_MFF(NIS_AFFINE, 4);
    } else {
      switch (km) {
        case -4: NIS_APPLY_MFQ(4, true); break;
        case -8: NIS_APPLY_MFQ(8, true); break;
        case -16: NIS_APPLY_MFQ(16, true); break;
        case -32: NIS_APPLY_MFQ(32, true); break;
        default: return hipErrorInvalidValue;
      }
    }
    return hipGetLastError();
  }
  if (c.kind == NIS_PWLIN) {
    NIS_APPLY_MF(NIS_PWLIN, 4, false);
  } else if (c.kind == NIS_AFFINE) {
    NIS_APPLY_MF(NIS_AFFINE, 4, false);
  } else {
    switch (km) {
      case 4: NIS_APPLY_MF(NIS_PWQUAD, 4, false); break;
      case 8: NIS_APPLY_MF(NIS_PWQUAD, 8, false); break;
      case 16: NIS_APPLY_MF(NIS_PWQUAD, 16, false); break;
      case 32: NIS_APPLY_MF(NIS_PWQUAD, 32, false); break;
      case -4: NIS_APPLY_MFQ(4, false); break;
      case -8: NIS_APPLY_MFQ(8, false); break;
      case -16: NIS_APPLY_MFQ(16, false); break;
      case -32: NIS_APPLY_MFQ(32, false); break;
      default: return hipErrorInvalidValue;
    }
  }
  return hipGetLastError();
}
#else
template <>
bool mfma_supported<NIS_WM>(const CellK&) { return false; }
template <>
int mfma_blocks_per_cu<NIS_WM>(const CellK&) { return 1; }
template <>
hipError_t launch_stats_mfma<NIS_WM>(const CellK&, const MfmaW&, const double*, long, long, int, int, hipStream_t) {
  return hipErrorInvalidValue;
}
template <>
hipError_t launch_apply_mfma<NIS_WM>(const CellK&, const MfmaW&, double*, long, long, int, int, int*, const FuseF&, int, hipStream_t) {
  return hipErrorInvalidValue;
}
#endif
