// Persistent whole-step kernel of the d = 2 two-cell PWQuad chain (README / C1 / C2 model): interface between api.hip and
// flow_mega.hip.  See flow_mega.hip for the design.
#pragma once
#include "nis_common.h"

#define NIS_MEGA_TPB 256
// Geometry (compile time): resident workgroups per CU = teams.  Each team has ONE wave per SIMD, and one wave alone issues
// fp64 at half the pipe rate, so the number of teams sets how much of a hand-off's latency other teams can cover.
//   4 teams: 40 KB of LDS and <= 128 VGPRs per workgroup -- 3912 resident samples (N <= 1,001,472 at 256 members), the bin's
//            three values picked by selects, records swept in two halves through an 8 KB scratch;
//   3 teams: 52 KB / <= 168 VGPRs -- 4096 samples, LDS gather columns for the bin's values, one sweep.
#ifndef NIS_MEGA_TEAMS
#define NIS_MEGA_TEAMS 3                 // measured: 3 teams 52.6 us/step, 4 teams 55.9 us/step (C2, 1e6)
#endif
#if NIS_MEGA_TEAMS == 4
#define NIS_MEGA_CAP 3912            // samples of one batch a workgroup keeps in LDS (8 bytes each)
#define NIS_MEGA_GATHER 0
#define NIS_MEGA_SWEEP_REC 128       // records per sweep chunk (64 bytes each in the LDS scratch)
#else
#define NIS_MEGA_CAP 4096
#define NIS_MEGA_GATHER 1
#define NIS_MEGA_SWEEP_REC 256
#endif
#define NIS_MEGA_SLOTS 7             // hand-offs per batch: BN_0, BN_2, BN_3 of cell 0, BN_0, BN_2, BN_3 of cell 1 (+1 spare)
#define NIS_MEGA_MAX_NT 256          // team size limit (a member sweeps NT x 16 granules with 16 loads per thread)
#define NIS_MEGA_MAX_TEAMS NIS_MEGA_TEAMS
#define NIS_MEGA_REPL 32             // replicas of every arrival counter (each on a line of its own)
#define NIS_MEGA_MAX_G 1500          // epoch tags: (batch index within the team) * 8 + slot + 1 < 4096

struct MegaCell {
  const double* wp;                  // packed hidden weights L x 4 x 4 (k_pack image, WM = 4)
  const double* wlast;               // 9 x 4
  const double* bias;                // 9
  const double* gb;                  // (L+1) x 2 x 4 gamma | beta
  double* gstat;                     // (L+1) x G x 2 x 4: batch mean | unbiased variance for k_group_epilogue
};

struct MegaArgs {
  MegaCell cell[2];
  const float* w;                    // (G, N, 2) float32 rows; null: counter-based draws (nis_philox_rows' mapping)
  float* w_out;                      // optional (G, N, 2): the draws that were used (tests)
  long N;
  int G, NT, teams, n_loc;
  int col0;                          // column of w that cell 0 passes through
  unsigned long long seed, stream0;  // Philox key, stream id of batch 0
  const unsigned long long* rng_state;   // device {seed, next stream} overriding them (graph replays), or null
  double* spill_j;                   // [teams][NT][CAP] Jacobian of cell 0
  float* spill_x;                    // [teams][NT][CAP] cell 0's pass-through input (float32 lattice value)
  unsigned long long* rec;           // [teams][SLOTS][NT][16] hand-off granules {tag, 32-bit half of a double}
  unsigned* ctr;                     // [teams][8][REPL] arrival counters (stride 32 words), zeroed by the launch function
  int* status;                       // bit 1: a hand-off timed out (results invalid)
  unsigned launch_seq;               // tag salt: 20 bits
  double fp[4];                      // camel: a, b, s, 1/s^2
  double* fpart;                     // moment records (G, maxb, 4)
  int maxb;
};

// bytes of device scratch behind the flow workspace body (spill + granules + counters), for `teams` x `NT` workgroups
size_t nis_mega_scratch_bytes();
// resident geometry on the current device: team size (= CUs) and number of teams; false: the kernel cannot run here
bool nis_mega_geometry(int* NT, int* teams);
// carve the scratch, reset counters (and, when `fresh`, the granule tags) and launch
hipError_t nis_mega_launch(MegaArgs a, void* scratch, bool fresh, hipStream_t s);
// the draws of batch streams [stream0, stream0 + G) as (G, N, 2) float32 rows
hipError_t nis_philox_rows_launch(unsigned long long seed, unsigned long long stream0, int G, long N, float* out, hipStream_t s);
