/*
 * l2hmc_hip.h -- C ABI of libl2hmc_hip.so: the MI355X (gfx950) implementation
 * of the L2HMC augmented-leapfrog hot path of saforem2/l2hmc.
 *
 * Drop-in boundary.  The reference is pure Python on TensorFlow 1.x; its "FFI"
 * for this path is the set of TF graph ops its Dynamics classes emit.  Each
 * entry point below replaces the ops of the reference function cited next to
 * it (paths relative to the reference checkout, `l2hmc/...`).  The Python host
 * classes in l2hmc_amd/ (GaugeDynamics, Dynamics, propose, ...) bind these via
 * ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 (int32 where said);
 *     rows are chains (or chain x direction pairs), row-major [rows][D];
 *   - nothing allocates, frees or synchronises: work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream); scratch
 *     comes from the caller through (ws, ws_bytes), sized by the *_ws_bytes
 *     queries, so calls are hipGraph-capturable;
 *   - inputs are never written; outputs never alias inputs unless documented;
 *   - return value: L2HMC_OK or an error code; l2hmc_last_error() gives text.
 *     Shape/argument violations are rejected on the host before any launch;
 *   - randomness is always an input buffer (l2hmc_fill_* produce them), so a
 *     caller can replay the reference's draws;
 *   - direction codes: 0 = forward (_forward_lf), 1 = backward (_backward_lf).
 */
#ifndef L2HMC_HIP_H
#define L2HMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define L2HMC_OK 0
#define L2HMC_ERR_ARG 1       /* bad shape / null pointer / unsupported size */
#define L2HMC_ERR_HIP 2       /* a HIP runtime call or launch failed */
#define L2HMC_ERR_WORKSPACE 3 /* workspace too small */

#define L2HMC_ABI_VERSION 1

typedef void* l2hmc_stream_t;

int l2hmc_abi_version(void);
const char* l2hmc_last_error(void);

/* ------------------------------------------------------------------------
 * U(1) lattice: action, force, observables.
 *   lattice/lattice.py:337-362 total_action; :285-313 calc_plaq_observables;
 *   gauge_model.py:659-725 _calc_plaq_sums/_total_actions/_avg_plaqs/_top_charges;
 *   dynamics/gauge_dynamics.py:698-709 grad_potential (autodiff of the action;
 *   closed form in lattice/gauge_lattice.py:427-459).
 * x: [rows][T][X][2].  Any output pointer may be NULL (skipped).
 *   action[r]  = sum_ij (1 - cos P)          force[r][:] = beta * dS/dx
 *   avg_plaq[r]= sum_ij cos P / (T*X)        top_charge[r] = sum_ij project(P) / 2pi
 * ------------------------------------------------------------------------ */
int l2hmc_u1_action_force(const float* x, int64_t rows, int32_t T, int32_t X, float beta,
                          float* action, float* force, float* avg_plaq, float* top_charge,
                          l2hmc_stream_t stream);

/* gauge_model.py:659-681: plaq[r][i][j] = x0[i,j] - x1[i,j] - x0[i,j+1] + x1[i+1,j]. */
int l2hmc_u1_plaq_sums(const float* x, int64_t rows, int32_t T, int32_t X, float* plaq,
                       l2hmc_stream_t stream);

/* gauge_model.py:1180,1388: samples = np.mod(x_out, 2*pi) between MCMC steps, done on the device
 * (fp32, result in [0, 2*pi)).  out may alias x. */
int l2hmc_wrap_angle(const float* x, int64_t n, float* out, l2hmc_stream_t stream);

/* gauge_dynamics.py:683-689 / utils/dynamics.py:112-113: out[r] = 0.5 * sum_d v^2. */
int l2hmc_kinetic_energy(const float* v, int64_t rows, int32_t D, float* out, l2hmc_stream_t stream);

/* ------------------------------------------------------------------------
 * Dense S/T/Q network (GenericNet / dense trunk of ConvNet3D / `network` MLP).
 *   network/generic_net.py:129-146, network/conv_net.py:264-280,
 *   utils/network.py:89-114.
 * Weights are PACKED by the host (l2hmc_amd/network.py) from the reference's
 * [in,out] Dense kernels into k-contiguous ("[out][in]") device buffers:
 *   w1_t   [H][Ka+Kb]  rows of v_layer (first input, Ka cols) then x_layer (Kb)
 *   wt     [2][H]      t_layer kernel            b1 [H] = b_v + b_x + b_t
 *   wh_t   [H][H]      h_layer                   bh [H]
 *   whd_t  [3][D][H]   scale / translation / transformation layers
 *   bhd    [3][D]      their biases
 *   coeff_s, coeff_q [D]  coeff_scale / coeff_transformation (exp applied on device)
 *   q_tanh: 0 = GenericNet/ConvNet3D (no tanh on transformation, quirk Q1),
 *           1 = utils/network.py `network` (ScaleTanh on F as well).
 * Shapes (generic_net.py:20-93 accepts any x_dim / num_hidden):
 *   - the layer-by-layer kernels (l2hmc_stq_dense, every l2hmc_gauge_* entry point) take ANY positive D, H, Ka, Kb
 *     and any lattice T x X; widths that are multiples of 32 with 16-byte aligned rows take the staged 16-byte
 *     loads, everything else (6x6: D = 72, H = 288; 3x5: D = 30) a bounds-checked instantiation of the same
 *     kernels -- identical arithmetic, slower loads;
 *   - the whole-trajectory kernels exist for D = 128 (T*X = 64, X a power of two): GenericNet H = 512 and
 *     ConvNet3D F = 8 / H = 256; other shapes run layer by layer (l2hmc_dense_pack_bytes() == 0 says which);
 *   - the TRAINING entry points (l2hmc_gauge_train_*) need D, H, Ka, Kb multiples of 32;
 *   - ConvNet3D needs T and X multiples of 4 (two 2x2 poolings) and filter sizes (3,3,2), (2,2,2).
 * ------------------------------------------------------------------------ */
typedef struct l2hmc_dense_net {
  int32_t D;   /* output width (x_dim) */
  int32_t H;   /* hidden width */
  int32_t Ka;  /* width of first input  (v_layer / embed_1) */
  int32_t Kb;  /* width of second input (x_layer / embed_2) */
  const float* w1_t;
  const float* wt;
  const float* b1;
  const float* wh_t;
  const float* bh;
  const float* whd_t;
  const float* bhd;
  const float* coeff_s;
  const float* coeff_q;
  int32_t q_tanh;
  int32_t reserved;
  const float* packed; /* optional: l2hmc_dense_pack() image for the fused trajectory kernel, or NULL */
} l2hmc_dense_net;

/* Fragment-ordered copy of (w1_t, wh_t, whd_t) for the whole-trajectory kernel:
 * [wave][k-chunk][n-tile][lane][4] per layer, so every B-operand load of a wave is
 * one contiguous 1 KiB read.  GenericNet plans get a second image behind it for the
 * kernel's sub-tile form ([wave][k-chunk][64-column block][4][lane][4]; batches that cannot
 * put a 16-row tile on every CU run 4 / 8 / 12 rows per workgroup, bit-identical results).
 * pack_bytes() is 0 when the shape has no fused kernel (then leave .packed NULL: the
 * layer-by-layer kernels are used).  Re-pack after every weight update. */
size_t l2hmc_dense_pack_bytes(const l2hmc_dense_net* net);
int l2hmc_dense_pack(const l2hmc_dense_net* net, float* packed, l2hmc_stream_t stream);

/* Convolutional front-end of ConvNet3D, channels_last (network/conv_net.py:90-164, 247-262):
 * per network input Conv3D(F,(3,3,2),same,relu) -> MaxPool3D(2,2,'same') -> Conv3D(2F,(2,2,2),same,relu)
 * -> MaxPool3D -> flatten to nflat = (T/4)*(X/4)*2F features, which feed the dense trunk above with
 * Ka = Kb = nflat.  Kernels are passed in the Keras layout [k0][k1][k2][Cin][Cout] unchanged.
 *   *_a: first input  (conv_v1 / conv_v2)      *_b: second input (conv_x1 / conv_x2)              */
typedef struct l2hmc_conv3d_front {
  int32_t F;          /* num_filters (gauge_dynamics.py:130: space_size) */
  int32_t reserved;
  const float* w1_a; const float* b1_a; const float* w2_a; const float* b2_a;
  const float* w1_b; const float* b1_b; const float* w2_b; const float* b2_b;
} l2hmc_conv3d_front;

/* scratch for one net evaluation on `rows` rows: two [rows][H] activations */
size_t l2hmc_stq_ws_bytes(int64_t rows, int32_t H);

/* (S, T, Q) = net([a, b * bmask, t])  with t = [t_cos, t_sin] for every row.
 * a: [rows][Ka], b: [rows][Kb], bmask: [Kb] or NULL.  S, T, Q: [rows][D]. */
int l2hmc_stq_dense(const l2hmc_dense_net* net, const float* a, const float* b, const float* bmask,
                    float t_cos, float t_sin, int64_t rows, float* S, float* T, float* Q,
                    void* ws, size_t ws_bytes, l2hmc_stream_t stream);

/* ConvNet3D.call (network/conv_net.py:247-280) on lattice inputs a, b: [rows][2*T*X]. */
size_t l2hmc_stq_conv3d_ws_bytes(int64_t rows, int32_t H, int32_t T, int32_t X, int32_t F);
int l2hmc_stq_conv3d(const l2hmc_conv3d_front* front, const l2hmc_dense_net* net, int32_t T, int32_t X,
                     const float* a, const float* b, const float* bmask, float t_cos, float t_sin,
                     int64_t rows, float* S, float* Tr, float* Q, void* ws, size_t ws_bytes,
                     l2hmc_stream_t stream);

/* ------------------------------------------------------------------------
 * Leapfrog sub-updates on materialised S/T/Q (standalone forms).
 *   v: gauge_dynamics.py:486-508 (dir 0), :537-561 (dir 1)
 *   x: gauge_dynamics.py:511-534 (dir 0), :565-590 (dir 1);
 *      `keep` is the mask fed to the net (1 = element kept, 0 = updated).
 * logdet[r] receives sum_d s (v) or sum_d (1-keep) s (x).  out may alias in.
 * ------------------------------------------------------------------------ */
int l2hmc_lf_update_v(const float* v, const float* grad, const float* S, const float* T,
                      const float* Q, float eps, int32_t dir, int64_t rows, int32_t D,
                      float* v_out, float* logdet, l2hmc_stream_t stream);
int l2hmc_lf_update_x(const float* x, const float* v, const float* keep, const float* S,
                      const float* T, const float* Q, float eps, int32_t dir, int64_t rows,
                      int32_t D, float* x_out, float* logdet, l2hmc_stream_t stream);

/* gauge_dynamics.py:592-609 / utils/dynamics.py:312-319:
 * p = exp(min(h_old - h_new + sumlogdet, 0)), non-finite -> 0. */
int l2hmc_accept_prob(const float* h_old, const float* h_new, const float* sumlogdet, int64_t n,
                      float* p, l2hmc_stream_t stream);

/* gauge_dynamics.py:221-257 (strict=1: accept iff p > u, forward iff coin > 0.5)
 * utils/sampler.py:33-59    (strict=0: accept iff p - u >= 0, forward iff coin != 0).
 * xf,vf,pf / xb,vb,pb: forward / backward trajectories of the same B chains. */
int l2hmc_mix_accept(const float* x, const float* xf, const float* vf, const float* pf,
                     const float* xb, const float* vb, const float* pb, const float* coin,
                     const float* u, int32_t strict, int64_t B, int32_t D, float* x_prop,
                     float* v_prop, float* p, float* x_out, l2hmc_stream_t stream);

/* ------------------------------------------------------------------------
 * Lattice integrator: gauge_dynamics.py:412-483 (_forward_lf/_backward_lf),
 * :261-313 (transition_kernel), :195-259 (apply_transition).
 * ------------------------------------------------------------------------ */
#define L2HMC_PLAN_LAYERED 1   /* never use the fused whole-trajectory kernel */
#define L2HMC_PLAN_CONV3D 2    /* nets are ConvNet3D: xfront / vfront are set, nets have Ka = Kb = nflat */
#define L2HMC_PLAN_SELECTED_ONLY 4   /* l2hmc_gauge_mcmc_step integrates only the direction each chain's coin picks:
                                      * same draws, same outputs as the default whenever the other direction is
                                      * finite (the reference multiplies it by an exact 0), half the work */
#define L2HMC_PLAN_RECOMPUTE 8       /* layer-by-layer path: form every first-layer product anew, as the reference's graph
                                      * does, instead of keeping the ones a leapfrog step repeats (XNet's product with
                                      * the momentum across the two position sub-updates; VNet's whole product from the
                                      * end of one step to the start of the next).  Same bits either way; diagnostic */
#define L2HMC_PLAN_TILES16_ONLY 16   /* whole-trajectory kernel: every batch on the 16-row form (no sub-tile form, no 32-row
                                      * form, no launch split).  All forms give the same bits; A/B and the bit-identity test */
#define L2HMC_PLAN_ALL_COLUMNS 32    /* layer-by-layer path: a position sub-update forms S / T / Q for EVERY column, as the
                                      * reference's graph does, instead of only the columns its mask lets move (identical x, v
                                      * for finite trajectories; 0 x non-finite differs: DESIGN.md D1) */
typedef struct l2hmc_gauge_plan {
  int32_t T, X;            /* lattice extents; D = 2*T*X */
  int32_t num_steps;       /* N_LF */
  int32_t hmc;             /* 1: S=T=Q=0 (gauge_dynamics.py:102-108), nets ignored */
  float eps;
  int32_t flags;           /* L2HMC_PLAN_* bits */
  const float* masks;      /* [num_steps][D] 0/1, gauge_dynamics.py:651-661 */
  l2hmc_dense_net xnet;    /* position_fn */
  l2hmc_dense_net vnet;    /* momentum_fn */
  l2hmc_conv3d_front xfront;   /* conv layers of position_fn (L2HMC_PLAN_CONV3D only) */
  l2hmc_conv3d_front vfront;   /* conv layers of momentum_fn */
} l2hmc_gauge_plan;

size_t l2hmc_gauge_ws_bytes(const l2hmc_gauge_plan* plan, int64_t rows);

/* Which kernels a plan runs through (no reference counterpart; for callers that report or size by it):
 * 1 = a whole-trajectory kernel exists for this shape and L2HMC_PLAN_LAYERED is not set, 0 = layer by layer,
 * negative = invalid plan (l2hmc_last_error says why). */
int l2hmc_gauge_plan_fused(const l2hmc_gauge_plan* plan);

/* How one MCMC step of a whole-trajectory plan (GenericNet, 8x8) over `rows_all` chain-rows is cut into launches on a
 * device with `cus` compute units (host logic only, no device call, no reference counterpart): writes up to three
 * {rows, rows per workgroup} parts and returns their number.  Rows per workgroup 4 / 8 / 12 = sub-tile form, 16, 32. */
int l2hmc_gauge_step_plan(int64_t rows_all, int32_t cus, int64_t* rows_out, int32_t* rows_per_wg_out);

/* One augmented leapfrog step IN PLACE on x, v: [rows][D]; dir: [rows] int32 per
 * row (0 fwd, 1 bwd), or NULL = all forward.  `step` is the loop counter t of
 * transition_kernel: backward rows use index num_steps-1-step for time and
 * mask, as _backward_lf does internally.  logdet[rows] is ACCUMULATED (+=). */
int l2hmc_gauge_leapfrog(const l2hmc_gauge_plan* plan, float beta, int32_t step, float* x, float* v,
                         const int32_t* dir, int64_t rows, float* logdet, void* ws, size_t ws_bytes,
                         l2hmc_stream_t stream);

/* Full trajectory of `rows` chain-direction pairs: x0, v0 -> x_out, v_out,
 * sumlogdet, accept probability (any of the last two may be NULL). */
int l2hmc_gauge_trajectory(const l2hmc_gauge_plan* plan, float beta, const float* x0, const float* v0,
                           const int32_t* dir, int64_t rows, float* x_out, float* v_out,
                           float* sumlogdet, float* p_accept, void* ws, size_t ws_bytes,
                           l2hmc_stream_t stream);

/* apply_transition on B chains.  v0_f, v0_b: [B][D] momenta for the two
 * directions; coin, u: [B] uniforms.  both_directions=1 integrates forward AND
 * backward for every chain like the reference and mixes (2B rows); 0 integrates
 * only the direction each chain's coin selects (B rows; identical outputs when
 * the unselected trajectory is finite). Outputs [B][D], [B][D], [B], [B][D]. */
size_t l2hmc_gauge_transition_ws_bytes(const l2hmc_gauge_plan* plan, int64_t B, int32_t both_directions);
int l2hmc_gauge_transition(const l2hmc_gauge_plan* plan, float beta, const float* x, const float* v0_f,
                           const float* v0_b, const float* coin, const float* u, int64_t B,
                           int32_t both_directions, float* x_prop, float* v_prop, float* p_accept,
                           float* x_out, void* ws, size_t ws_bytes, l2hmc_stream_t stream);

/* apply_transition (gauge_dynamics.py:195-259) with the library's own draws -- Philox streams (seed, 2*draw) for the
 * stacked momenta [v0_f; v0_b] and (seed, 2*draw+1) for coin | u, exactly the layout of l2hmc_gauge_mcmc_step,
 * reproducible with l2hmc_fill_normal/_uniform -- instead of caller-provided ones: what `dynamics(x, beta)` is in
 * the reference.  ONE launch for plans with a whole-trajectory kernel.  ws: l2hmc_gauge_mcmc_step_ws_bytes(plan, B). */
int l2hmc_gauge_transition_draw(const l2hmc_gauge_plan* plan, float beta, const float* x, int64_t B, uint64_t seed,
                                uint64_t draw, float* x_prop, float* v_prop, float* p_accept, float* x_out,
                                void* ws, size_t ws_bytes, l2hmc_stream_t stream);

/* One MCMC step of the sampling loop on device-resident chains (gauge_model.py:1371-1388 around
 * apply_transition): draws (Philox streams (seed, 2*draw) for the stacked momenta [v0_f; v0_b] and
 * (seed, 2*draw+1) for coin | u -- reproducible with l2hmc_fill_normal/_uniform), both trajectories, mix,
 * accept/reject, then x <- mod(x_out, 2*pi) IN PLACE.  Per-chain outputs (any may be NULL): px = accept
 * probability; actions / plaqs / charges = observables of the step's INPUT samples (as :256-266);
 * charge_diff = |Q(x_in) - Q(x_out)| (:718-725).  ONE launch when the plan has a whole-trajectory kernel: the
 * kernel generates its own draws, a workgroup owns both directions of its chains and finishes the step in its
 * epilogue (no intermediate ever reaches HBM); other plans run the same step through the public ops. */
size_t l2hmc_gauge_mcmc_step_ws_bytes(const l2hmc_gauge_plan* plan, int64_t B);
int l2hmc_gauge_mcmc_step(const l2hmc_gauge_plan* plan, float beta, float* x, int64_t B, uint64_t seed,
                          uint64_t draw, float* px, float* actions, float* plaqs, float* charges,
                          float* charge_diff, void* ws, size_t ws_bytes, l2hmc_stream_t stream);
/* Same step, out of place (x_next may equal x_in) and with the per-step scalars the cross-shard reduce_mean of
 * gauge_model.py:795 needs: step_sums (FOUR floats, or NULL) = [sum p_accept, sum |dQ|, B, scratch] -- the first
 * three are what a rank all-reduces per accept/reject (fixed summation order).  The fourth word is a ticket the
 * one-launch step uses to find its last workgroup: it MUST BE 0 ON ENTRY and is left at 0, so a buffer zeroed once
 * can be reused for every step. */
int l2hmc_gauge_mcmc_step_ex(const l2hmc_gauge_plan* plan, float beta, const float* x_in, float* x_next, int64_t B,
                             uint64_t seed, uint64_t draw, float* px, float* actions, float* plaqs, float* charges,
                             float* charge_diff, float* step_sums, void* ws, size_t ws_bytes,
                             l2hmc_stream_t stream);

/* Forward value of the training loss, per chain (gauge_model.py:766-795): terms[b] = std_loss + charge_loss;
 * the scalar loss is their mean over ALL chains of all ranks.  x, x_prop, z: [B][2*T*X]; px, pz: [B].
 * metric: 0 'l1', 1 'l2', 2 'cos', 3 'cos2', 4 'cos_diff' (:632-657).  Both auxiliary terms compare z with
 * x_prop, exactly as the reference writes them. */
int l2hmc_gauge_loss_terms(const float* x, const float* x_prop, const float* px, const float* z,
                           const float* pz, int64_t B, int32_t T, int32_t X, int32_t metric,
                           float loss_scale, float aux_weight, float std_weight, float charge_weight,
                           float* terms, l2hmc_stream_t stream);

/* ------------------------------------------------------------------------
 * Training: gradients of the loss with respect to the network weights and the step size, and the
 * optimiser step -- tf.gradients(loss, dynamics.variables) + AdamOptimizer.apply_gradients of
 * gauge_model.py:799-830, :942-969.  GenericNet and ConvNet3D plans (hmc = 0).
 *
 * Chains are integrated in the direction their coin selects (rows = x chains then z chains for the
 * loss of gauge_model.py:728-797; the masked-out direction carries exactly zero gradient in the
 * reference's graph).  The sequence for one training step is
 *   l2hmc_gauge_train_forward   -> x_N, v_N, sumlogdet, p        (keeps every intermediate in `ws`)
 *   l2hmc_gauge_loss_backward   -> per-chain loss, d loss / d (x_N, v_N, sumlogdet)
 *   l2hmc_gauge_train_backward  -> gradients (same k-contiguous layout as struct l2hmc_dense_net)
 *   [all-reduce of the flat gradient buffers across ranks]
 *   l2hmc_grad_sumsq / l2hmc_adam_step
 * ------------------------------------------------------------------------ */
typedef struct l2hmc_dense_grads {
  float* w1_t;     /* [H][Ka+Kb] */
  float* wt;       /* [2][H]  */
  float* b1;       /* [H]: gradient of EACH of the three first-layer biases (they are summed in b1) */
  float* wh_t;     /* [H][H]  */
  float* bh;       /* [H]     */
  float* whd_t;    /* [3][D][H] */
  float* bhd;      /* [3][D]  */
  float* coeff_s;  /* [D]     */
  float* coeff_q;  /* [D]     */
} l2hmc_dense_grads;

/* Gradients of a ConvNet3D front-end, in the Keras layout of the weights (struct l2hmc_conv3d_front);
 * the dd = 1 slice of the second kernel only ever multiplies padding, so its gradient is exactly 0. */
typedef struct l2hmc_conv3d_grads {
  float* w1_a; float* b1_a; float* w2_a; float* b2_a;   /* first input  */
  float* w1_b; float* b1_b; float* w2_b; float* b2_b;   /* second input */
} l2hmc_conv3d_grads;

size_t l2hmc_gauge_train_ws_bytes(const l2hmc_gauge_plan* plan, int64_t rows);
/* Same outputs as l2hmc_gauge_trajectory (dir: per-row 0 forward / 1 backward, NULL = all forward);
 * `ws` must stay untouched until the matching l2hmc_gauge_train_backward has run. */
int l2hmc_gauge_train_forward(const l2hmc_gauge_plan* plan, float beta, const float* x0, const float* v0,
                              const int32_t* dir, int64_t rows, float* x_out, float* v_out, float* sumlogdet,
                              float* p_accept, void* ws, size_t ws_bytes, l2hmc_stream_t stream);
/* dx, dv: [rows][D] = d loss / d (x_N, v_N) on entry, overwritten (d loss / d (x_0, v_0) on exit);
 * dlogdet: [rows] = d loss / d sumlogdet.  gx, gv, deps (1 float) are overwritten with the gradients.
 * gxf, gvf: for L2HMC_PLAN_CONV3D plans, where the gradients of the Conv3D kernels / biases go; NULL for
 * GenericNet plans. */
int l2hmc_gauge_train_backward(const l2hmc_gauge_plan* plan, float beta, const int32_t* dir, int64_t rows,
                               float* dx, float* dv, const float* dlogdet, const l2hmc_dense_grads* gx,
                               const l2hmc_dense_grads* gv, const l2hmc_conv3d_grads* gxf,
                               const l2hmc_conv3d_grads* gvf, float* deps, void* ws, size_t ws_bytes,
                               l2hmc_stream_t stream);
/* The same pass with BUCKET NOTIFICATIONS for a data-parallel caller (gauge_model.py:942-943: Horovod's
 * DistributedOptimizer all-reduces gradients tensor by tensor while the rest of the backward graph still runs).
 * The weight gradients are finished in seven contiguous groups; as soon as everything that produces group b has
 * been ENQUEUED on `stream`, `on_bucket(user, b)` runs on the calling host thread -- the caller records an event
 * there and starts its collective for that group on another stream, so the exchange of group b overlaps the
 * computation of group b + 1.  Groups, in the order they complete (k = 0: gx, k = 1: gv):
 *   3k + 0: w1_t, wt, b1        3k + 1: wh_t, bh        3k + 2: whd_t, bhd, coeff_s, coeff_q
 *   L2HMC_GRAD_BUCKET_REST (6): Conv3D front-end gradients (gxf, gvf) and deps -- always last.
 * Results are identical to l2hmc_gauge_train_backward. */
#define L2HMC_GRAD_BUCKET_REST 6
typedef void (*l2hmc_bucket_fn)(void* user, int32_t bucket);
int l2hmc_gauge_train_backward_buckets(const l2hmc_gauge_plan* plan, float beta, const int32_t* dir, int64_t rows,
                                       float* dx, float* dv, const float* dlogdet, const l2hmc_dense_grads* gx,
                                       const l2hmc_dense_grads* gv, const l2hmc_conv3d_grads* gxf,
                                       const l2hmc_conv3d_grads* gvf, float* deps, void* ws, size_t ws_bytes,
                                       l2hmc_stream_t stream, l2hmc_bucket_fn on_bucket, void* user);
/* Loss (gauge_model.py:728-797) and its gradient with respect to the proposed states of the 2B stacked
 * chains (rows [0,B): started at x; rows [B,2B): started at z), through the accept probabilities
 * (gauge_dynamics.py:592-609).  x0, xN, vN: [2B][2*T*X]; p: [2B]; inv_count = 1 / (number of chains the
 * mean of :795 runs over, all ranks); terms: [B] or NULL; dxN, dvN: [2B][D]; dlogdet: [2B]. */
int l2hmc_gauge_loss_backward(int32_t T, int32_t X, float beta, const float* x0, const float* xN,
                              const float* vN, const float* p, int64_t B, int32_t metric, float loss_scale,
                              float aux_weight, float std_weight, float charge_weight, float inv_count,
                              float* terms, float* dxN, float* dvN, float* dlogdet, l2hmc_stream_t stream);
/* *out (device) = [*out +] sum g[i]^2, elements in [tri_lo, tri_hi) counted three times (the packed
 * first-layer bias stands for three reference variables); fixed summation order. */
int l2hmc_grad_sumsq(const float* g, int64_t n, int64_t tri_lo, int64_t tri_hi, float* out, int32_t accumulate,
                     l2hmc_stream_t stream);
/* Adam as tf.train.AdamOptimizer applies it: m, v moments; lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) is
 * computed by the caller; w -= lr_t * m / (sqrt(v) + eps).  gnorm_sq (device scalar) != NULL applies
 * tf.clip_by_global_norm(clip) first.  Elements in [tri_lo, tri_hi) move three times as far (see above). */
int l2hmc_adam_step(float* w, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1,
                    float beta2, float eps, const float* gnorm_sq, float clip, int64_t tri_lo, int64_t tri_hi,
                    l2hmc_stream_t stream);

/* ------------------------------------------------------------------------
 * Generic integrator on 2-D toy targets (MoG / SCG):
 *   utils/dynamics.py:120-225,255-319; utils/sampler.py:28-59;
 *   utils/distributions.py:32-39,63-68,151-158.
 * Target: mixture of K Gaussians (K=1, log_const ignored => plain Gaussian).
 *   energy(x) = -logsumexp_k( -0.5 (x-mu_k)^T P_k (x-mu_k) + log_const[k] ) / temperature
 * ------------------------------------------------------------------------ */
/* Limits of this path (utils/network.py:89-114 takes any x_dim / num_nodes; the reference's own configurations are
 * 2-D targets with num_nodes 10 (SCGExperiment.ipynb) and 50 (mog_model.py, network.py:89)): x_dim <= 8 with at most
 * 8 mixture components (the closed-form energy, its gradient and Hessian-vector product live in registers), and
 * num_nodes <= 64 (both networks' weights stay in LDS for the whole trajectory).  Larger values are refused with
 * L2HMC_ERR_ARG; wider toy networks belong on the MFMA path above. */
#define L2HMC_MAX_MIX 8
#define L2HMC_MAX_SMALL_DIM 8
typedef struct l2hmc_mog_target {
  int32_t dim;                     /* <= L2HMC_MAX_SMALL_DIM */
  int32_t K;                       /* <= L2HMC_MAX_MIX */
  int32_t is_gaussian;             /* 1: energy = 0.5 (x-mu)^T P (x-mu) (distributions.py:63-68) */
  float temperature;
  const float* mu;                 /* [K][dim] */
  const float* prec;               /* [K][dim][dim] inverse covariances */
  const float* log_const;          /* [K] log(pi_k / sqrt((2pi)^dim det Sigma_k)) */
} l2hmc_mog_target;

typedef struct l2hmc_small_plan {
  int32_t x_dim, num_nodes, trajectory_length, hmc;
  float eps;
  int32_t first_layer_form;        /* 0: chosen by batch size; 1 / 2: force the matrix-pipe / VALU form of the first layer;
                                    * 3: two waves per group of 16 rows (latency form, 17..64 hidden units).  The forms
                                    * group their sums differently and agree to rounding */
  const float* masks;              /* [trajectory_length][x_dim] */
  l2hmc_dense_net xnet, vnet;      /* q_tanh = 1, Ka = Kb = x_dim */
  l2hmc_mog_target target;
} l2hmc_small_plan;

int l2hmc_mog_energy_grad(const l2hmc_mog_target* tgt, const float* x, int64_t rows, float* energy,
                          float* grad, l2hmc_stream_t stream);

/* Dynamics.forward / .backward (dir per row, NULL = all forward): whole
 * trajectory per chain in one kernel. */
int l2hmc_small_trajectory(const l2hmc_small_plan* plan, const float* x0, const float* v0,
                           const int32_t* dir, int64_t rows, float* x_out, float* v_out,
                           float* sumlogdet, float* p_accept, l2hmc_stream_t stream);

/* utils/sampler.py:28-55 `propose` (and :57-59 `tf_accept` when x_out != NULL) in ONE launch for the L2HMC sampler
 * (plan->hmc == 0).  Chain c's direction bit, forward momentum, backward momentum and Metropolis-Hastings uniform are
 * the Philox streams (seed, draw0), (seed, draw0 + 1), (seed, draw0 + 2), (seed, draw0 + 3): exactly the elements
 * l2hmc_fill_uniform(B) / l2hmc_fill_normal(B * x_dim) write with those (seed, offset) pairs, so the call equals
 * fill x 4 + l2hmc_small_trajectory(2B rows) + l2hmc_mix_accept(strict = 0) bit for bit.  Both trajectories of a
 * chain run in the same wave.  Outputs (each may be NULL): Lx, Lv [B][x_dim] the proposal selected by the direction
 * bit (forward iff uniform >= 0.5), px [B] its accept probability, x_out [B][x_dim] = Lx where px - u >= 0 else x. */
int l2hmc_small_propose(const l2hmc_small_plan* plan, const float* x, int64_t B, uint64_t seed, uint64_t draw0,
                        float* Lx, float* Lv, float* px, float* x_out, l2hmc_stream_t stream);

/* One training evaluation on the toy targets (mog_model.py:324-363): `rows` = 2B stacked chains (B started at
 * x, B at z ~ N(0,1); sampler.py:28-55 picks a direction per chain, passed in `dir`), each integrated in its
 * direction; per chain v = |x0 - x_N|^2 * p + 1e-4, term = scale / v - v / scale, loss = inv_count * sum of all
 * 2B terms (inv_count = 1/B).  ONE launch runs forward, loss and the whole reverse pass on-chip.
 * Outputs: x_out, v_out [rows][x_dim] proposals; p_accept, terms [rows]; grads (device, overwritten):
 * [xnet gradient | vnet gradient | d loss / d eps], each network in the flat order
 * [w1_t | wt | b1 | wh_t | bh | whd_t | bhd | coeff_s | coeff_q] of struct l2hmc_dense_net
 * (d loss / d alpha = eps * d loss / d eps, utils/dynamics.py:51-60). */
size_t l2hmc_small_train_ws_bytes(const l2hmc_small_plan* plan, int64_t rows);
int l2hmc_small_train_step(const l2hmc_small_plan* plan, const float* x0, const float* v0, const int32_t* dir,
                           int64_t rows, float scale, float inv_count, float* x_out, float* v_out,
                           float* p_accept, float* terms, float* grads, void* ws, size_t ws_bytes,
                           l2hmc_stream_t stream);

/* ------------------------------------------------------------------------
 * Counter-based RNG (Philox4x32-10) standing in for tf.random_normal /
 * tf.random_uniform (gauge_dynamics.py:223,246,269).  Same (seed, offset, n)
 * => same stream on any launch geometry.
 * ------------------------------------------------------------------------ */
int l2hmc_fill_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, l2hmc_stream_t stream);
int l2hmc_fill_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, l2hmc_stream_t stream);

/* ------------------------------------------------------------------------
 * Measurement aid (no reference counterpart): time every launch of one kernel
 * class with HIP events recorded on the launch stream.  begin() arms it (0
 * disarms), end() synchronises the recorded events and returns the summed
 * kernel time and launch count.  One process-wide recorder (mutex-protected; launches of concurrent threads are
 * recorded one after the other); not for use while capturing a graph.
 *   1 = first dense layer   2 = hidden dense layer   3 = heads (+update)
 *   4 = u1_action_force     5 = fused whole-trajectory kernel
 *   6 = ConvNet3D front-end 7 = toy-target trajectory kernel (l2hmc_small_*)
 * ------------------------------------------------------------------------ */
int l2hmc_profile_begin(int32_t kernel_class);
int l2hmc_profile_end(double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* L2HMC_HIP_H */
