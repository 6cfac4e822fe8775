// Launch descriptors shared by stq_dense.hip (kernels) and leapfrog.hip (orchestration).
#pragma once
#include "common.h"

namespace l2hmc {

// L1 / L2:  out = relu(A . Wt^T + bias [+ t-term])
struct GemmReluArgs {
  const float* A1; int lda1; int K1;   // columns [0, K1) of A
  const float* A2; int lda2;           // columns [K1, K) of A (NULL if K1 == K)
  const float* cmask_f;                // optional [K-K1] multiplier on A2, forward rows
  const float* cmask_b;                //                                   backward rows
  const int* dir;                      // [rows] 0/1 or NULL (all forward)
  const float* Wt; int K; int N;       // [N][K]
  const float* bias;                   // [N]
  const float* wt0; const float* wt1;  // t_layer kernel rows [N] each, or NULL
  float tc_f, ts_f, tc_b, ts_b;        // (cos, sin) of the step's time, per direction
  float* out; int ldo;
  int64_t rows;
  int mtiles, ntiles;
  unsigned long long* stamps;          // diagnostic builds only (-DL2HMC_STAMPS), else NULL
  int kind;                            // 0: relu layer (forward); 3: out = product where gate > 0; 4: plain product
  const float* gate; int ldg;          // kind 3: forward activations [rows][N]
  // Recurring first-layer products (leapfrog.hip: two position sub-updates share the momentum half of the product,
  // the momentum update at the start of a step repeats the whole product of the previous step's last one).  The
  // product is an fp32 fma chain in ascending k, so a chain cut at a tile boundary, kept in fp32 and continued later
  // gives the bits of the uninterrupted chain.
  const float* acc_in; int k_begin;    // start the accumulators from acc_in [rows][N] and the k-loop at k_begin
  float* acc_out; int k_dump;          // write the raw accumulators (no bias / relu) once k has reached k_dump
};                                     // (both multiples of the k-tile; k_dump == K: the full product)

// h1 = relu(pre + bias + t . Wt) from a saved first-layer product (the epilogue of gemm_relu_kernel<., 1> alone)
struct L1FinishArgs {
  const float* pre; float* out; int N; int64_t rows;
  const float* bias; const float* wt0; const float* wt1;
  const int* dir; float tc_f, ts_f, tc_b, ts_b;
};
int launch_l1_finish(const L1FinishArgs& a, hipStream_t stream);

// heads: (S,T,Q) = h2 . Whd^T + bhd, then materialise or fused v/x update
enum HeadsMode { kHeadsMaterialise = 0, kHeadsUpdateV = 1, kHeadsUpdateX = 2 };

struct HeadsArgs {
  const float* A; int lda; int K;        // h2 [rows][K]
  const float* Wt;                       // [3][D][K]
  const float* bhd;                      // [3][D]
  const float* cs; const float* cq;      // [D]
  int q_tanh; int D; int64_t rows;
  int mode;
  float* S; float* T; float* Q;          // mode 0 outputs [rows][D]
  float* x; float* v;                    // updated in place (v: mode 1, x: mode 2)
  const float* g;                        // force (mode 1)
  const int* dir;                        // [rows] or NULL
  const float* keep_f; const float* keep_b;  // [D] keep masks per direction (mode 2)
  float eps;
  float* ld_part; int ncb;               // [rows][ncb] log-det partials, += own slot
  int mtiles, ntiles;
  unsigned long long* stamps;            // diagnostic builds only
  // Position sub-updates only touch the columns their keep mask does not hold fixed (x' = keep x + (1 - keep) (...),
  // the log-det term carries the same factor): with these set the kernels form S / T / Q for those columns alone.
  // cols_f / cols_b: ascending column lists for forward / backward rows, *cnt_f / *cnt_b their lengths (device
  // memory, written by active_cols_kernel); rows >= dir_split are the backward ones (a multiple of the row tile).
  const int* cols_f; const int* cols_b; const int* cnt_f; const int* cnt_b; int64_t dir_split;
};

// In-kernel cycle stamps (cdna_hip_programming.md section 7): compiled in only with
// -DL2HMC_STAMPS, never in the shipped library.  Slot layout per workgroup: 8 words.
#ifdef L2HMC_STAMPS
extern unsigned long long* g_stamp_buf;
extern int g_stamp_cls;
#define L2HMC_STAMP(i)                                                                    \
  do {                                                                                    \
    if (p.stamps && threadIdx.x == 0) {                                                   \
      unsigned long long t_;                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                  \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
      __builtin_amdgcn_sched_barrier(0);                                                  \
      p.stamps[blockIdx.x * 8 + (i)] = t_;                                                \
    }                                                                                     \
  } while (0)
#define L2HMC_STAMP_REAL(i)                                                               \
  do {                                                                                    \
    if (p.stamps && threadIdx.x == 0) {                                                   \
      unsigned long long t_;                                                              \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
      p.stamps[blockIdx.x * 8 + (i)] = t_;                                                \
    }                                                                                     \
  } while (0)
#else
#define L2HMC_STAMP(i) do {} while (0)
#define L2HMC_STAMP_REAL(i) do {} while (0)
#endif

// conv front-end of ConvNet3D, both network inputs in one launch (index 0: first input, 1: second)
struct ConvFrontArgs {
  int T, X, F;
  const float* in[2];                    // [rows][2*T*X]
  const float* cmask_f; const float* cmask_b;   // optional [D] multiplier on the SECOND input, per direction
  const int* dir;                        // [rows] or NULL
  const float* w1[2]; const float* b1[2];   // Conv3D kernels [3][3][2][1][F], biases [F]
  const float* w2[2]; const float* b2[2];   // Conv3D kernels [2][2][2][F][2F], biases [2F]
  float* out[2]; int ldo;                // [rows][nflat] flattened features
  int64_t rows;
  int cpw;                               // chains per workgroup (set by the launcher)
  int ldi;                               // row stride of `in` (0 = 2*T*X)
  int only;                              // 0: both inputs; 1: the first input only; 2: the second input only
  unsigned long long* stamps;            // diagnostic builds only (class 7): phase boundaries per workgroup
};

// backward of the front-end (training path); input `which` lives at column offset which*D of `in` / `din`
// and which*nflat of `dfeat`
struct ConvBwdArgs {
  int T, X, F;
  const float* in; int ldi;              // taped raw inputs [rows][>= 2D]
  const float* dfeat; int ldf;           // d loss / d features [rows][>= 2*nflat]
  const float* w1[2]; const float* b1[2]; const float* w2[2]; const float* b2[2];
  float* din; int ldd;                   // out: d loss / d raw inputs [rows][>= 2D]
  float* part;                           // [workgroups][2][conv3d_bwd_part_floats(F)], +=
  int64_t rows;
  int cpw;
  unsigned long long* stamps;            // diagnostic builds only (class 6): phase boundaries of workgroup x
};
size_t conv3d_bwd_part_floats(int F);
int conv3d_cpw(int T, int X, int F);
int launch_conv3d_front_bwd(ConvBwdArgs& a, hipStream_t stream);
int conv3d_nflat(int T, int X, int F);
int launch_conv3d_front(ConvFrontArgs& a, hipStream_t stream);

int launch_gemm_relu(GemmReluArgs& a, hipStream_t stream);
int launch_heads(HeadsArgs& a, hipStream_t stream);
// lists[s][0] = columns with mask[s][c] != 1, lists[s][1] = columns with mask[s][c] != 0 (ascending), counts[s][0..1]
int launch_active_cols(const float* masks, int num_steps, int D, int* lists, int* counts, hipStream_t stream);
int dense_net_supported(const l2hmc_dense_net* n);
int dense_net_tileable(const l2hmc_dense_net* n);   // every width a multiple of 32 (fast staged loads)
int fused_plan_supported(const l2hmc_gauge_plan* p);
// Optional tape of the whole-trajectory kernel for the training path (train.hip): per network, every call's
// first-layer input [a | b*mask], hidden activations, (S, T, Q) planes and the state the sub-update consumed,
// laid out [call][rows][.] exactly as the layered taped forward writes them.  NULL pointers = no taping.
struct FusedTape {
  float* in;    // [calls][rows][2D]
  float* h1;    // [calls][rows][H]
  float* h2;    // [calls][rows][H]
  float* stq;   // [calls][3][rows][D]
  float* st;    // [calls][rows][D]
  float* feat;  // ConvNet3D plans: [calls][rows][Ka+Kb] front-end features (NULL otherwise)
  unsigned* gate;   // [calls][2 layers][workgroups][256 threads]: relu masks of h1 / h2 in the lane's own C-fragment
                    // order (bit t*4+e), so the fused reverse pass gates its deltas with one 4-byte load per layer
};
int launch_fused_trajectory(const l2hmc_gauge_plan* p, float beta, int step_begin, int step_end,
                            const float* x0, const float* v0, const int* dir, int64_t rows, float* x_out,
                            float* v_out, float* logdet, int logdet_accumulate, float* p_accept,
                            hipStream_t stream, int64_t x_mod = 0, int64_t dir_split = 0,
                            const FusedTape* tape_x = nullptr, const FusedTape* tape_v = nullptr);
int launch_fused_step(const l2hmc_gauge_plan* p, float beta, const float* x_in, float* x_next, int64_t B,
                      uint64_t seed, uint64_t draw, int both, float* px, float* actions, float* plaqs, float* charges,
                      float* dq, float* step_sums, float* part, hipStream_t stream, float* x_prop = nullptr,
                      float* v_prop = nullptr, float* x_out = nullptr);
// whole-trajectory reverse pass (fused_train.hip); deltas_*: {dout, d2, d1} tapes, coef_parts: {dcs_x, dcq_x, dcs_v, dcq_v}
size_t fused_bwd_pack_floats(const l2hmc_dense_net* n);
int fused_train_supported(const l2hmc_gauge_plan* p);
int fused_train_forward_supported(const l2hmc_gauge_plan* p);
int launch_fused_train_backward(const l2hmc_gauge_plan* p, float beta, const int* dir, int64_t rows, float* dx,
                                float* dv, const float* dld, const FusedTape& tx, const FusedTape& tv,
                                float* const deltas_x[3], float* const deltas_v[3], float* pack_x, float* pack_v,
                                float* const coef_parts[4], float* deps_part, hipStream_t stream);
// one network call of a layered reverse pass in one launch (fused_train.hip: gauge_trunk_bwd_kernel)
struct TrunkBwdArgs {
  int mode;                                      // 1: momentum update (VNet call), 2: position update (XNet call)
  float eps;
  const float* pk;                               // backward image of the net (pack_fused_bwd_kernel)
  const float* cs; const float* cq; int q_tanh;
  const float* keep_f; const float* keep_b;      // mode 2: keep masks of this sub-update per direction
  const int* dir; int64_t rows;
  const float* stq; int64_t plane;               // this call's tape: S, T, Q planes; consumed state; raw inputs; h1; h2
  const float* st; const float* in; const float* h1; const float* h2;
  const float* dld;
  float* dx; float* dv;                          // [rows][D] in/out
  float* dg;                                     // mode 1 out: d loss / d force [rows][D]
  float* dout; float* d2; float* d1;             // this call's delta tapes
  float* dfeat;                                  // out [rows][K1]
  float* dcs_part; float* dcq_part; float* deps_part;   // [workgroups][D] x 2, [workgroups]; +=
};
int trunk_bwd_supported(const l2hmc_dense_net* n);
int launch_trunk_bwd_pack(const l2hmc_dense_net* n, float* pack, hipStream_t stream);
int launch_trunk_bwd(TrunkBwdArgs& a, hipStream_t stream);
int launch_u1_action_force(const float* x, int64_t rows, int T, int X, float beta, float* action,
                           float* force, float* avg_plaq, float* top_charge, hipStream_t stream);

}  // namespace l2hmc
