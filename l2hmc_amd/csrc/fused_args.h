// Launch descriptor of the whole-trajectory kernels (fused_traj.hip: 16 rows per workgroup; fused_traj4.hip: the
// sub-tile form for batches that cannot fill 16-row tiles on every CU).
#pragma once
#include "stq_dense.h"

namespace l2hmc {

struct FusedArgs {
  int T, X, num_steps, step_begin, step_end;
  float eps, beta;
  const float* masks;                    // [num_steps][D]
  l2hmc_dense_net xnet, vnet;            // .packed must be set
  l2hmc_conv3d_front xfront, vfront;     // ConvNet3D only
  const float* x0; const float* v0;      // [rows][D]
  const int* dir;                        // [rows] or NULL
  int64_t x_mod;                         // > 0: row r starts from x0[r % x_mod] (both directions of one batch)
  int64_t dir_split;                     // dir == NULL and > 0: rows >= dir_split integrate backward
  int64_t rows;
  float* x_out; float* v_out;            // [rows][D]
  float* logdet;                         // [rows] or NULL; written (=) or accumulated (+=)
  int logdet_accumulate;
  float* p_accept;                       // [rows] or NULL
  unsigned long long* stamps;            // diagnostic builds only
  int stagger;                           // cycles of start delay per in-XCD workgroup index (0 = none)
  FusedTape tx, tv;                      // training tape per network (all-NULL = sampling)
  // Whole-MCMC-step mode (l2hmc_gauge_mcmc_step, l2hmc_gauge_transition_draw; step_B > 0): the kernel draws its own momenta / coin /
  // MH uniform (Philox streams (seed, 2 draw) and (seed, 2 draw + 1), bit-identical to l2hmc_fill_*), integrates,
  // mixes, accepts, measures and wraps -- ONE launch per MCMC step.  A workgroup then owns 8 chains x both
  // directions (rows 0-7 forward, 8-15 backward of the same chains) or, with step_both = 0, 16 chains in the
  // direction their coin selects; x0 = the step's input samples [B][D], v0 / dir / x_out / v_out are unused.
  float* step_x_next;                    // [B][D] wrapped output samples (may alias x0), or NULL
  float* step_xprop; float* step_vprop; float* step_xout;   // [B][D] apply_transition's outputs (unwrapped), or NULL
  int64_t step_B;                        // > 0 switches the mode on: chains of the WHOLE batch (Philox stream offsets, count)
  int64_t step_Bl, step_chain0;          // chains of THIS launch and the first one's index in the batch; the per-chain
                                         // pointers (x0, step_x_next, step_px, ...) are already moved to that chain
  int step_sums_acc;                     // 1: add this launch's sums to step_sums (second launch of a batch cut in two)
  unsigned long long step_seed, step_draw;
  int step_both;
  float* step_px; float* step_act; float* step_plq; float* step_chg; float* step_dq;   // [B] each, or NULL
  float* step_sums;                      // [4] = [sum p, sum |dQ|, B, ticket] or NULL; ticket 0 on entry, left 0
  float* step_part;                      // [2 * workgroups] scratch for the fixed-order sums
};

// the sub-tile form (fused_traj4.hip): GenericNet 8x8 plans, sampling only
size_t fused4_pack_floats(const l2hmc_dense_net* n);                 // floats of its weight image (appended to the 16-row image)
int launch_fused4_pack(const l2hmc_dense_net* n, float* image4, hipStream_t stream);
int fused4_rows_per_wg(int64_t rows, int cus);                       // 0: use the 16-row form; else 4, 8 or 12 (at most one workgroup per CU)
int launch_fused4(const FusedArgs& a, int rows_per_wg, hipStream_t stream);

// the 32-row form (fused_traj32.hip): same plans, same packed image as the 16-row form; for batches of more than one
// round of 16-row workgroups (each weight fragment then feeds two MFMAs)
int fused32_supported(const l2hmc_dense_net* n);
int launch_fused32(const FusedArgs& a, hipStream_t stream);

}  // namespace l2hmc
