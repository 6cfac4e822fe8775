#!/usr/bin/env python
"""Throughput of the L2HMC hot path on MI355X: chain-leapfrog-steps per second.

    python bench.py --gpus N --steps K --warmup W [--config {1,2,3,4,5}] [--scaling {weak,strong}]

One "step" is one pass of the hot path over one batch of synthetic chains, everything resident in HBM:
  * lattice configs (3, 4, 5): one `apply_transition` (dynamics/gauge_dynamics.py:195-259) -- momenta, direction
    coin and MH uniform drawn on the device, BOTH directions integrated (as the reference does), accept/reject,
    wrap to [0, 2pi) and the per-step observables;
  * toy configs (1, 2): one `propose(x, dynamics, do_mh_step=True)` (utils/sampler.py:28-59).

`--config c` selects BASELINE.json configs[c-1]; the default (3) is the configuration the metric is quoted on
(2D U(1) 8x8, beta 2.0, batch 2048 per GPU, 10 leapfrog steps, GenericNet H=512, fp32) and its line also
carries a compact `configs` table with the other four workloads at their per-GPU size.

N > 1: a bare `python bench.py --gpus N` starts its own N ranks (child processes through
torch.distributed.run, before this process touches the GPU); under torch.distributed.run (WORLD_SIZE set) the
process is a rank.  One rank per GPU; chains are independent, so the ranks share nothing but one small RCCL
all-reduce of the per-step scalar sums (accept probability, |dQ|, count), issued on a side stream.
  * `--scaling weak`  : every rank integrates the config's per-GPU batch (128 / 4096 / 2048 / 1024 / 2048 chains)
  * `--scaling strong`: the config's GLOBAL batch (128 / 4096 / 2048 / 8192 / 16384) is cut into N contiguous
                        shards (l2hmc_amd.dist.shard_bounds): rank r integrates chains [lo_r, hi_r)
Default: weak for configs 1-3, strong for configs 4 and 5 (BASELINE.json names those as sharded workloads:
8192 and 16384 chains over 8 GPUs = 1024 and 2048 per GPU; reference: gauge_model.py:942-943, 1008, 1095).

Prints ONE JSON line (rank 0).  `value` counts USEFUL chain-leapfrog steps (chains x N_LF per step); the
executed count is twice that (both directions) and is what the roofline FLOPs use.  Every `frac` / `tflops` /
`achieved` in the line prices the FLOPs the kernels EXECUTE (recurring first-layer products are kept, position
sub-updates form only the columns their mask lets move: `executed_macs_per_lf`) and can therefore never exceed 1;
the reference graph's own count (every first layer and every head column formed anew) divided
by the same time is reported beside it as `reference_flops_equiv_tflops` / `reference_flops_equiv_frac`.
A secondary leg that fails is reported under an "error" key AND makes the process exit non-zero after the line is
printed.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, chip-level parameters

# BASELINE.json configs[i-1]; sizes and synthetic inputs as SURVEY.md 8 (sizes table) / 8d state them.
# `global` = the batch BASELINE.json names; `per_gpu` = what one GPU of the named partition integrates.
CONFIGS = {
    1: dict(kind="toy", target="scg", D=2, H=10, N=5, eps=0.1, per_gpu=128, glob=128, scaling="weak",
            steps=50, warmup=5, prewarm=20,
            name="2-D strongly-correlated Gaussian, batch 128, 5 LF steps, MLP H=10 (BASELINE.json configs[0])"),
    2: dict(kind="toy", target="mog", D=2, H=50, N=10, eps=0.1, per_gpu=4096, glob=4096, scaling="weak",
            steps=50, warmup=5, prewarm=20,
            name="2-D mixture of Gaussians, batch 4096, 10 LF steps, MLP H=50 (BASELINE.json configs[1])"),
    3: dict(kind="gauge", L=8, beta=2.0, eps=0.25, N=10, arch="generic", per_gpu=2048, glob=2048, scaling="weak",
            steps=50, warmup=5, prewarm=150,
            name="U(1) 8x8 lattice, beta=2.0, batch 2048 per GPU, 10 LF steps, GenericNet H=512 "
                 "(BASELINE.json configs[2])"),
    4: dict(kind="gauge", L=16, beta=3.0, eps=0.2, N=15, arch="conv3D", per_gpu=1024, glob=8192, scaling="strong",
            steps=10, warmup=2, prewarm=6,
            name="U(1) 16x16 lattice, beta=3.0, batch 8192 sharded over 8 GPUs (1024 per GPU), 15 LF steps, "
                 "ConvNet3D F=16 H=1024 (BASELINE.json configs[3])"),
    5: dict(kind="gauge", L=32, beta=4.0, eps=0.1, N=25, arch="generic", per_gpu=2048, glob=16384, scaling="strong",
            steps=3, warmup=1, prewarm=1,
            name="U(1) 32x32 lattice, beta=4.0, batch 16384 sharded over 8 GPUs (2048 per GPU), 25 LF steps, "
                 "GenericNet H=8192 (BASELINE.json configs[4])"),
}
HEADLINE_METRIC = "leapfrog-steps/sec (whole node), 8x8 U(1) batch 2048, 10 LF"


def net_macs(D, H, Ka=None, Kb=None):
    """dense trunk MACs per call per chain = (Ka + Kb) H + 2H + H^2 + 3HD (generic: Ka = Kb = D)."""
    Ka = D if Ka is None else Ka
    Kb = D if Kb is None else Kb
    return (Ka + Kb) * H + 2 * H + H * H + 3 * H * D


def conv_front_macs(L, F):
    """both inputs through conv1 (3,3,2) and conv2 (2,2,2): 2 (L L 2 F 18 + (L/2)^2 2F 8F)."""
    return 2 * (L * L * 2 * F * 18 + (L // 2) ** 2 * 2 * F * 8 * F)


def config_macs(cfg):
    """(MACs per net call per chain, dims dict): FLOPs per chain-LF step = 8 x MACs (4 net calls x 2)."""
    c = CONFIGS[cfg]
    if c["kind"] == "toy":
        D, H = c["D"], c["H"]
        return net_macs(D, H), dict(D=D, H=H)
    L = c["L"]
    D = 2 * L * L
    if c["arch"] == "generic":
        H = 4 * D
        return net_macs(D, H), dict(D=D, H=H, Ka=D, Kb=D)
    F, H = L, 2 * D
    nflat = (L // 4) ** 2 * 2 * F
    return conv_front_macs(L, F) + net_macs(D, H, nflat, nflat), dict(D=D, H=H, Ka=nflat, Kb=nflat, F=F)


def executed_macs_per_lf(cfg, fused, active_cols):
    """MACs per chain-LF step the kernels actually execute (the `frac` figures price the reference's count, 4 x
    config_macs: it forms every first layer anew and every head column whatever the keep mask).
    Kept first-layer products (both paths): of the 4 first layers / conv front-ends of a step 2.5 are formed (+ the
    first momentum update of a trajectory).  active_cols (layer-by-layer path, rows' directions known per tile):
    the two position sub-updates form S / T / Q for the half of the columns they move."""
    c = CONFIGS[cfg]
    m, d = config_macs(cfg)
    if c["kind"] == "toy":             # the toy kernel evaluates every network call in full
        return 4.0 * m
    D, H, N = d["D"], d["H"], c["N"]
    l1 = (d["Ka"] + d["Kb"]) * H + (conv_front_macs(c["L"], d["F"]) if c["arch"] == "conv3D" else 0)
    l1_calls = (2.5 * N + 1) / N
    heads_calls = 3.0 if active_cols else 4.0
    return l1_calls * l1 + 4 * (2 * H + H * H) + heads_calls * 3 * H * D


def heads_use_active_columns(rows, D, H, dir_split):
    """Mirror of csrc/stq_dense.hip:launch_heads for aligned shapes (every BASELINE config): does a position
    sub-update over `rows` stacked rows (forward rows [0, dir_split), backward rows behind them) form S / T / Q only for
    the columns its mask lets move?  The 128 x 64-tile form (grids of >= 512 tiles) needs the split on a 128-row edge,
    the 64 x 32-tile form on a 64-row edge."""
    split_ok = lambda bm: dir_split >= rows or dir_split % bm == 0      # noqa: E731
    if H % 16 == 0 and D % 64 == 0 and -(-rows // 128) * (D // 64) >= 512:
        return split_ok(128)
    return split_ok(64)


def rank_chains(cfg, world, rank, scaling):
    """Chains [lo, hi) of the job's global batch that `rank` integrates, and the global batch.
    weak: per-GPU batch fixed, global = world x per_gpu.  strong: the config's global batch cut into `world`
    contiguous shards (l2hmc_amd/dist.py:shard_bounds; SURVEY.md 8e: 1024 / 2048 per GPU at 8 GPUs)."""
    from l2hmc_amd.dist import shard_bounds
    c = CONFIGS[cfg]
    if scaling == "weak":
        return rank * c["per_gpu"], (rank + 1) * c["per_gpu"], world * c["per_gpu"]
    lo, hi = shard_bounds(c["glob"], world, rank)
    return lo, hi, c["glob"]


def committed_traffic():
    """(bytes per launch, file) from the newest profiles/r*_pmc_fused_kernel.json, or (None, None)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fused_kernel.json")), reverse=True):
        try:
            with open(path) as f:
                return int(json.load(f)["hbm_side"]["traffic_bytes_per_launch"]), os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None, None


def launch_command(gpus, env, argv, port=None):
    """The decision the driver's bare `python bench.py --gpus N` needs (gauge_model.py:2041 relies on mpirun to
    start the ranks; here the script starts its own): None when this process IS a rank (WORLD_SIZE is set by
    torch.distributed.run) or when N == 1; otherwise the command that starts N fresh ranks of this script,
    one per GPU, rendezvous on 127.0.0.1."""
    if gpus <= 1 or "WORLD_SIZE" in env:
        return None
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def self_launch(cmd):
    """Parent of the ranks: never touches the GPU (no HIP call has been made in this process), starts the ranks
    as CHILD processes (no exec), relays their output -- rank 0 prints the JSON line -- and returns the launcher's
    exit code, non-zero if any rank failed."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def rendezvous_only(world, rank, cfg, scaling):
    """The cross-rank plumbing of the timed region without the GPU work (CPU test hook): barrier on both sides,
    MAX over ranks of the elapsed time, every rank's chain block [lo, hi) gathered; rank 0 prints one line."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    lo, hi, glob = rank_chains(cfg, world, rank, scaling)
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    ranks = torch.ones(1, dtype=torch.float64)
    bounds = torch.zeros(world, 2, dtype=torch.int64)
    bounds[rank, 0], bounds[rank, 1] = lo, hi
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(ranks, op=dist.ReduceOp.SUM)
        dist.all_reduce(bounds, op=dist.ReduceOp.SUM)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": world, "ranks_seen": int(ranks.item()),
                          "max_elapsed_s": float(t.item()), "config": cfg, "scaling": scaling,
                          "global_batch": glob, "chains": bounds.tolist()}), flush=True)
    return 0


# ---------------------------------------------------------------------------------------------------------------
# workload builders (the product's own constructors; reference initialisation under fixed NumPy seeds)
# ---------------------------------------------------------------------------------------------------------------
def build_gauge(cfg, batch, both_directions=True, arch=None):
    import l2hmc_amd as la
    c = CONFIGS[cfg]
    np.random.seed(106)
    lat = la.GaugeLattice(c["L"], c["L"], 2, 'U1', num_samples=batch, rand=False)
    return la.GaugeDynamics(lat, lat.get_energy_function(), eps=c["eps"], hmc=False,
                            network_arch=arch or c["arch"], num_steps=c["N"], eps_trainable=True,
                            data_format='channels_last', both_directions=both_directions)


def net_state_numpy(net):
    return {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}


def build_toy(cfg):
    import l2hmc_amd as la
    c = CONFIGS[cfg]
    np.random.seed(106)
    if c["target"] == "mog":           # mog_model.py:1063-1067, :1120
        target = la.GMM([np.array([1., 0.]), np.array([0., 1.])], [0.025 * np.eye(2)] * 2, [0.5, 0.5])
    else:                              # SCGExperiment.ipynb cell 3
        target = la.Gaussian(np.zeros(2), np.array([[50.05, -49.95], [-49.95, 50.05]]))
    H = c["H"]
    dyn = la.Dynamics(c["D"], target.get_energy_function(), trajectory_length=c["N"], eps=c["eps"],
                      net_factory=lambda d, scope, factor: la.network(d, scope, factor, num_nodes=H))
    return dyn, target


def profile_class(Lh, cls, run, _lib):
    """(avg us, launches) of kernel class `cls` over `run()`, HIP events on the launch stream."""
    _lib.check(Lh.l2hmc_profile_begin(cls))
    run()
    ms, n = C.c_double(), C.c_int64()
    _lib.check(Lh.l2hmc_profile_end(C.byref(ms), C.byref(n)))
    return (1e3 * ms.value / n.value if n.value else 0.0), int(n.value)


def gauge_kernel_classes(cfg, rows, fused, active_cols=False):
    """(class id, kernel name, EXECUTED FLOPs per launch, the reference graph's FLOPs for the same launch or None)
    of the kernels a lattice step launches."""
    c = CONFIGS[cfg]
    macs, d = config_macs(cfg)
    D, H = d["D"], d["H"]
    # The layered path keeps the first-layer products a leapfrog step repeats (csrc/leapfrog.hip): of the 4 first
    # layers (and conv front-ends) per step it launches 3 -- two whole ones and the second position sub-update's
    # half -- plus the very first momentum update of a trajectory: the AVERAGE launch carries this share of one
    # whole first layer's FLOPs (the whole-step figures always use the algorithmic count, 4 per step).
    kept = (2.5 * c["N"] + 1) / (3 * c["N"] + 1)
    if fused:
        name = ("gauge_traj_fused_kernel<128,512> (whole MCMC step in one launch: draws, both trajectories, "
                "mix / MH, observables, wrap)" if c["arch"] == "generic" else
                "gauge_traj_fused_kernel<128,256,64,conv> (whole MCMC step in one launch, conv front-end in LDS)")
        # executed: the kernel keeps the recurring first-layer products in registers (2.5 N + 1 first layers per
        # trajectory instead of 4 N); reference: every network call formed in full
        return [(5, name, 2.0 * executed_macs_per_lf(cfg, True, False) * rows * c["N"], 8.0 * macs * rows * c["N"])]
    out = [(1, "gemm_relu_kernel<.,1> (first dense layer; average over whole and half-K launches)",
            kept * 2.0 * rows * H * (d["Ka"] + d["Kb"]), None),
           (2, "gemm_relu_kernel<.,2> (hidden dense layer)", 2.0 * rows * H * H, None),
           (3, "heads_kernel (S/T/Q + sub-update + log-det; average over momentum updates -- all columns -- and "
               "position sub-updates -- the columns their keep mask lets move, where the rows' directions are known "
               "per row tile)", (0.75 if active_cols else 1.0) * 2.0 * rows * 3 * D * H, None)]
    if c["arch"] == "conv3D":
        out.append((6, "conv3d_front_kernel (both inputs: conv1+relu+pool, conv2+relu+pool; VALU, priced at the "
                       "fp32 MFMA rate; average over two-input and one-input launches)",
                    kept * 2.0 * rows * conv_front_macs(c["L"], d["F"]), None))
    return out


# ---------------------------------------------------------------------------------------------------------------
# one workload: the timed region + its roofline; used for the selected config and for the `configs` table
# ---------------------------------------------------------------------------------------------------------------
class Job:
    """Distributed context of this process."""

    def __init__(self, world, rank, dev, dist):
        self.world, self.rank, self.dev, self.dist = world, rank, dev, dist

    def barrier(self):
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, dt):
        if self.dist is None:
            return dt
        t = torch.tensor([dt], device=self.dev, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def run_workload(job, cfg, scaling, steps, warmup, both=True, layered=False, roofline=True, keep=None):
    """Builds config `cfg` at this rank's share, runs prewarm + W warm-up + EXACTLY K timed steps bracketed by
    barrier + synchronize, MAX over ranks.  Returns (result dict, state for the secondary legs)."""
    from l2hmc_amd import _lib
    import l2hmc_amd as la
    c = CONFIGS[cfg]
    lo, hi, glob = rank_chains(cfg, job.world, job.rank, scaling)
    B = hi - lo
    if B <= 0:
        raise SystemExit(f"bench.py: rank {job.rank} has no chains (config {cfg}, {scaling} scaling, "
                         f"{job.world} ranks)")
    macs, dims = config_macs(cfg)
    Lh = _lib.lib()
    state = {}
    if c["kind"] == "gauge":
        from l2hmc_amd import GaugeSampler
        dyn = build_gauge(cfg, B, both)
        dyn._seed = 1000 + job.rank                    # independent chains per rank
        dyn.fused = not layered
        D = dims["D"]
        x = torch.empty(B, D, device=job.dev)
        # hot start (lattice.py:131-135); Philox stream `lo`: every shard of the global batch has its own start
        _lib.check(Lh.l2hmc_fill_uniform(x.data_ptr(), x.numel(), 103, lo, _lib.stream_ptr()))
        x.mul_(2 * np.pi)
        sampler = GaugeSampler(dyn, dist=job.dist)     # transition + wrap + observables; one fused all-reduce
        stats = sampler.stats
        beta = c["beta"]

        def step(xc):
            return sampler.step(xc, beta)[0]
        state.update(dyn=dyn, sampler=sampler, stats=stats)
    else:
        dyn, target = build_toy(cfg)
        dyn._seed = 1000 + job.rank
        np.random.seed(102 + job.rank)
        x = _lib.as_dev(np.asarray(target.get_samples(B), dtype=np.float32), job.dev)   # exact draws of the target
        stats = None

        def step(xc):
            return la.propose(xc, dyn, do_mh_step=True)[3][0]
        state.update(dyn=dyn, target=target)

    # untimed pre-warm, in ADDITION to the W warm-up steps: after the idle seconds of start-up the GPU clock takes
    # ~20 ms of work to ramp (tools/warmup_probe.py -> profiles/r02_warmup_probe.txt), which a 5-step warm-up would
    # leak into a 20-step timing (a FIXED count per config: every lattice step carries a collective when sharded)
    prewarm = c["prewarm"]
    for i in range(prewarm):
        x = step(x)
        if i % 16 == 15:
            torch.cuda.synchronize()
    for _ in range(warmup):
        x = step(x)
    if stats is not None:
        stats.wait()                               # the untimed steps' scalars are folded before the clock starts
    job.barrier()
    history = []                                   # references only: no device work in the timed region
    t0 = time.perf_counter()
    for _ in range(steps):
        x = step(x)
        if keep:
            history.append(x)
    if stats is not None:
        stats.join()                               # every step's all-reduce is ordered before the synchronize
    job.barrier()
    dt = time.perf_counter() - t0
    if stats is not None:
        stats.wait()                               # host-side bookkeeping of the step scalars: not part of a step
    dt = job.max_over_ranks(dt)

    ndir = 2 if both else 1
    useful = glob * c["N"] * steps
    fused, active_cols = False, False
    if c["kind"] == "gauge":
        plan = state["dyn"]._plan()
        fused = (not layered) and Lh.l2hmc_gauge_plan_fused(C.byref(plan)) == 1
        # both directions: rows [0, B) forward, [B, 2B) backward; selected-only rows carry per-row directions
        # (no split known per tile: every column is formed)
        active_cols = (not fused) and both and heads_use_active_columns(2 * B, dims["D"], dims["H"], B)
    ex = 2.0 * executed_macs_per_lf(cfg, fused, active_cols)          # FLOPs per chain-LF step the kernels execute
    per_s = ndir * glob * c["N"] / (dt / steps) / 1e12 / job.world    # executed chain-LF steps per second per GPU / 1e12
    res = {"cfg": cfg, "workload": c["name"], "value": useful / dt, "ms_per_step": 1e3 * dt / steps,
           "steps": steps, "warmup": warmup, "prewarm_steps_untimed": prewarm, "scaling": scaling,
           "global_batch": glob, "chains_this_rank": [lo, hi], "num_steps": c["N"], "eps": c["eps"],
           "directions_integrated": ndir, "executed_chain_lf_per_step": ndir * glob * c["N"],
           "algorithmic_flops_per_chain_lf": 8 * macs, "executed_flops_per_chain_lf": ex,
           "whole_step_tflops": per_s * ex, "whole_step_frac": per_s * ex / PEAK_F32_MFMA_TFLOPS,
           "reference_flops_equiv_tflops": per_s * 8 * macs,
           "reference_flops_equiv_frac": per_s * 8 * macs / PEAK_F32_MFMA_TFLOPS}
    if c["kind"] == "gauge":
        res["lattice"], res["beta"] = [c["L"], c["L"]], c["beta"]
        res["mean_accept_prob"] = stats.mean_accept()
    state.update(x=x, step=step, history=history, B=B, dt=dt)

    # ---- roofline of the dominant kernel, HIP events on the launch stream (every rank runs the profiled steps
    #      -- lattice steps contain the per-step collective -- only rank 0 reports)
    if roofline:
        rows = ndir * B
        if c["kind"] == "toy":
            classes = [(7, "small_traj_mfma_kernel (one launch per propose: Philox draws, both trajectories, mix, MH)",
                        8.0 * macs * 2 * B * c["N"], None)]
        else:
            classes = gauge_kernel_classes(cfg, rows, fused, active_cols)
        nprof = max(1, min(steps, 20))
        per = []
        for cls, name, flops, ref_flops in classes:
            def run():
                xs = x
                for _ in range(nprof):
                    xs = step(xs)
            us, n = profile_class(Lh, cls, run, _lib)
            if n:
                per.append(dict(kernel=name, launches=n, avg_us=us, flops_per_launch=flops,
                                tflops=flops / (us * 1e-6) / 1e12, frac=flops / (us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS))
                if ref_flops is not None:
                    per[-1].update(reference_flops_per_launch=ref_flops,
                                   reference_flops_equiv_frac=ref_flops / (us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS)
        if stats is not None:
            stats.wait()
        if per:
            dom = max(per, key=lambda d: d["avg_us"] * d["launches"])
            res["roofline"] = {"bound": "mfma", "achieved": dom["tflops"], "peak": PEAK_F32_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "frac": dom["frac"], "traffic": None, "kernel": dom["kernel"],
                               "avg_launch_us": dom["avg_us"], "executed_flops_per_launch": dom["flops_per_launch"],
                               "flops_counted": "the FLOPs the kernel executes (recurring first-layer products kept, "
                                                "masked head columns skipped); the reference graph's count is under "
                                                "reference_flops_equiv_*",
                               "all_kernels": per, "whole_step_tflops": res["whole_step_tflops"],
                               "whole_step_frac": res["whole_step_frac"],
                               "reference_flops_equiv_tflops": res["reference_flops_equiv_tflops"],
                               "reference_flops_equiv_frac": res["reference_flops_equiv_frac"]}
            if "reference_flops_per_launch" in dom:
                res["roofline"]["reference_flops_per_launch"] = dom["reference_flops_per_launch"]
                res["roofline"]["kernel_reference_flops_equiv_frac"] = dom["reference_flops_equiv_frac"]
    return res, state


def cpu_baseline_for(cfg, state):
    """The reference's CPU path (op-for-op torch-CPU port, oracle/cpu_baseline.py) on a bounded sample of the
    same workload, rank 0 at N = 1 only."""
    from oracle import cpu_baseline as cb
    c = CONFIGS[cfg]
    dyn = state["dyn"]
    if c["kind"] == "toy":
        from oracle import dynamics as ogen
        if c["target"] == "mog":
            target = ogen.GMM([np.array([1., 0.]), np.array([0., 1.])], [0.025 * np.eye(2)] * 2, [0.5, 0.5])
        else:
            target = ogen.Gaussian(np.zeros(2), np.array([[50.05, -49.95], [-49.95, 50.05]]))
        r = cb.time_cpu_toy_baseline(target, c["N"], c["eps"], c["per_gpu"], net_state_numpy(dyn.XNet),
                                     net_state_numpy(dyn.VNet), dyn.mask.cpu().numpy(), budget_s=10.0)
        sample, what = c["per_gpu"], "propose calls (both directions)"
    else:
        # bounded sample: the full 2048 chains at cfg 3 (0.3 s per call); fewer where one call would take minutes
        sample = {3: 2048, 4: 128, 5: 32}[cfg]
        r = cb.time_cpu_baseline(c["L"], c["L"], c["N"], c["eps"], c["beta"], sample,
                                 net_state_numpy(dyn.position_fn), net_state_numpy(dyn.momentum_fn),
                                 dyn.mask.cpu().numpy(), budget_s={3: 15.0, 4: 12.0, 5: 10.0}[cfg],
                                 min_calls={3: 2, 4: 2, 5: 1}[cfg], arch=c["arch"],
                                 threads=None if cfg != 5 else "all")
        what = "apply_transition calls (both directions)"
    return {"value": r["value"], "unit": "chain-leapfrog-steps/s", "cores": r["cores"], "kind": "port",
            "sample": f"{r['calls']} timed {what} on {sample} chains of the same lattice/net/LF config, torch-CPU "
                      f"fp32 op-for-op port of the reference graph, median {r['seconds']:.3f} s per call; "
                      f"{r['cores']} threads of {r['cpus_available']} usable CPUs"}


def ess_of(history, chain_stats):
    """ESS per MCMC step with the reference's estimator (func_utils.py:45-54,114-120) on (cos, sin) of the links."""
    X = torch.stack(history).cpu().numpy()
    feats = np.concatenate([np.cos(X), np.sin(X)], axis=2)
    feats = feats - feats.mean(axis=(0, 1), keepdims=True)
    A = chain_stats.acl_spectrum(feats, 1.0)
    return float(chain_stats.ESS(A / A[0]))


def claim_stdout():
    """The driver parses ONE JSON line from stdout, and RCCL prints a version banner there when a communicator is
    created: keep a private handle on the real stdout for the line and point file descriptor 1 at stderr for everything
    else (this process and the libraries it loads)."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default per config: 50 / 50 / 50 / 10 / 3)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS),
                    help="BASELINE.json configs[c-1]; 3 = the configuration the metric is quoted on")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="weak: per-GPU batch fixed; strong: the config's global batch sharded over the ranks "
                         "(default: weak for configs 1-3, strong for 4 and 5)")
    ap.add_argument("--selected-only", action="store_true",
                    help="integrate only the direction each chain's coin selects (not the reference's work)")
    ap.add_argument("--layered", action="store_true", help="use the layer-by-layer kernels (no fused trajectory)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-configs-table", action="store_true",
                    help="skip the compact table of the other BASELINE configs in the default line")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step timing")
    ap.add_argument("--no-trained-ess", action="store_true",
                    help="skip the 250-step training run + ESS/sec of the trained sampler (N = 1 only)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="test hook (CPU, gloo): start the ranks, barrier, max-over-ranks reduction, print "
                         "{'rendezvous_only': true, ..., 'chains': [[lo, hi) per rank]} and exit -- no GPU")
    args = ap.parse_args()

    cmd = launch_command(args.gpus, os.environ, sys.argv[1:])
    if cmd is not None:                           # before anything initialises the GPU in this process
        raise SystemExit(self_launch(cmd))

    cfg = args.config
    c = CONFIGS[cfg]
    scaling = args.scaling or c["scaling"]
    steps = args.steps if args.steps is not None else c["steps"]
    warmup = args.warmup if args.warmup is not None else c["warmup"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (start it as `python bench.py "
                         f"--gpus N` or through torch.distributed.run with --nproc-per-node N)")
    if args.rendezvous_only:
        raise SystemExit(rendezvous_only(world, rank, cfg, scaling))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    json_out = claim_stdout()
    # Rehearsal knobs for a 1-GPU box (never set by the driver): all ranks on device 0 over gloo.
    backend = os.environ.get("L2HMC_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
    if os.environ.get("L2HMC_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    # (L2HMC_COLLECTIVES_AT_WORLD1=1: one-GPU rehearsal of the sharded flow -- a one-rank RCCL group is created and
    #  every collective below is really issued, see l2hmc_amd/dist.py:active)
    rehearse1 = world == 1 and os.environ.get("L2HMC_COLLECTIVES_AT_WORLD1") == "1"
    if rehearse1:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or rehearse1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    from l2hmc_amd import _lib, GaugeSampler, stats as chain_stats
    dev = torch.device("cuda", local_rank)
    job = Job(world, rank, dev, dist)
    both = not args.selected_only
    failed = []                                   # secondary legs that threw: reported AND a non-zero exit

    res, st = run_workload(job, cfg, scaling, steps, warmup, both=both, layered=args.layered,
                           roofline=not args.no_roofline, keep=(c["kind"] == "gauge" and cfg == 3))
    lo, hi = res["chains_this_rank"]
    short = {1: f"2-D SCG batch {res['global_batch']}, 5 LF", 2: f"2-D MoG batch {res['global_batch']}, 10 LF",
             3: f"8x8 U(1) batch {res['global_batch']}, 10 LF", 4: f"16x16 U(1) conv3D batch {res['global_batch']}, 15 LF",
             5: f"32x32 U(1) batch {res['global_batch']}, 25 LF"}
    out = {
        "metric": HEADLINE_METRIC if (cfg == 3 and scaling == "weak") else f"leapfrog-steps/sec (whole node), {short[cfg]}",
        "value": res["value"], "unit": "chain-leapfrog-steps/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "prewarm_steps_untimed": res["prewarm_steps_untimed"], "ms_per_step": res["ms_per_step"],
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": res["workload"], "baseline_config": cfg, "global_batch": res["global_batch"],
                   "chains_rank0": [lo, hi], "num_steps": res["num_steps"], "eps": res["eps"],
                   "directions_integrated": res["directions_integrated"],
                   "executed_chain_lf_per_step": res["executed_chain_lf_per_step"],
                   "algorithmic_flops_per_chain_lf": res["algorithmic_flops_per_chain_lf"],
                   "executed_flops_per_chain_lf": res["executed_flops_per_chain_lf"],
                   "parallelism": f"chains sharded over {world} GPU(s) ({scaling} scaling), weights replicated"},
    }
    for k in ("lattice", "beta", "mean_accept_prob"):
        if k in res:
            out["config"][k] = res[k]
    if "roofline" in res:
        out["roofline"] = res["roofline"]
    x, step, dyn = st["x"], st["step"], st["dyn"]

    if cfg == 3:
        sampler, stats, BATCH = st["sampler"], st["stats"], st["B"]
        beta, n_lf = c["beta"], c["N"]
        dt = st["dt"]
        # ---- secondary metric: ESS/sec with the reference's estimator on rank 0's chains over the timed steps
        if rank == 0 and steps >= 8:
            ess = ess_of(st["history"][-min(steps, 64):], chain_stats)
            out["config"]["ess_per_mcmc_step"] = ess
            out["config"]["ess_per_sec_whole_job"] = ess * res["global_batch"] * steps / dt
        st["history"].clear()

        if "roofline" in out and not args.layered:
            Lh = _lib.lib()
            # the ESS estimate above was host work (seconds of idle GPU): ramp the clock again and re-time the step
            # kernel, so that `roofline` describes the kernel at the clocks of the timed region
            xs = x
            for _ in range(100):
                xs = step(xs)
            stats.wait()
            D, Hd = 2 * c["L"] ** 2, 8 * c["L"] ** 2
            rows = (2 if both else 1) * BATCH

            def run():
                xs = x
                for _ in range(steps):
                    xs = step(xs)
            us, n = profile_class(Lh, 5, run, _lib)
            stats.wait()
            ex_lf, ref_lf = 2.0 * executed_macs_per_lf(3, True, False), 8.0 * net_macs(D, Hd)
            if n:
                tf = ex_lf * rows * n_lf / (us * 1e-6) / 1e12
                rtf = ref_lf * rows * n_lf / (us * 1e-6) / 1e12
                out["roofline"].update(achieved=tf, frac=tf / PEAK_F32_MFMA_TFLOPS, avg_launch_us=us,
                                       kernel_reference_flops_equiv_frac=rtf / PEAK_F32_MFMA_TFLOPS)
                out["roofline"]["all_kernels"][0].update(avg_us=us, launches=n, tflops=tf, frac=tf / PEAK_F32_MFMA_TFLOPS,
                                                         reference_flops_equiv_frac=rtf / PEAK_F32_MFMA_TFLOPS)
            # The step kernel also draws the momenta and finishes the step (mix, MH, observables, wrap).  The same
            # kernel launched for the trajectories alone (l2hmc_gauge_trajectory: same rows, same FLOPs, no step
            # prologue / epilogue) separates the integrator's MFMA efficiency from that fixed per-step work.
            x2 = torch.cat([x, x]).contiguous()
            v2 = dyn._normal(tuple(x2.shape))
            for _ in range(3):
                dyn.transition_kernel(x2, beta, forward=True, momentum=v2)
            us, n = profile_class(Lh, 5, lambda: [dyn.transition_kernel(x2, beta, forward=True, momentum=v2)
                                                  for _ in range(20)], _lib)
            if n:
                tf = ex_lf * x2.shape[0] * n_lf / (us * 1e-6) / 1e12
                out["roofline"]["trajectory_only"] = {
                    "what": "the same kernel launched for the two trajectories alone (l2hmc_gauge_trajectory, "
                            f"{x2.shape[0]} rows x {n_lf} LF): no draws, no mix / MH / observables / wrap",
                    "avg_launch_us": us, "tflops": tf, "frac": tf / PEAK_F32_MFMA_TFLOPS,
                    "reference_flops_equiv_frac": tf * ref_lf / ex_lf / PEAK_F32_MFMA_TFLOPS}
            # fabric-side bytes per launch of the fused kernel at this exact shape: read from the newest committed
            # PMC summary (tools/pmc_collect.sh + tools/pmc_summary.py -> profiles/rNN_pmc_fused_kernel.json);
            # null if no summary is committed or the shape differs -- never a literal
            if both and scaling == "weak":
                traffic, traffic_src = committed_traffic()
                out["roofline"]["traffic"] = traffic
                out["roofline"]["traffic_unit"] = ("bytes per launch, fabric side: 2 x FETCH_SIZE + WRITE_SIZE from "
                                                   f"separate rocprofv3 --pmc passes ({traffic_src})")

        # ---- the same workload with the reference's CLI-default architecture (conv3D, gauge_model.py:2307) ----
        if rank == 0 and world == 1 and not args.no_roofline:
            try:
                cdyn = build_gauge(3, BATCH, both, arch='conv3D')
                cdyn.set_masks(dyn.mask.cpu().numpy())
                csmp = GaugeSampler(cdyn)
                xc = x.clone()
                for _ in range(3):
                    xc = csmp.step(xc, beta)[0]
                torch.cuda.synchronize()
                tc0 = time.perf_counter()
                for _ in range(20):
                    xc = csmp.step(xc, beta)[0]
                torch.cuda.synchronize()
                tcd = (time.perf_counter() - tc0) / 20
                out["config"]["conv3D_arch"] = {"net": "ConvNet3D F=8, H=256", "ms_per_step": 1e3 * tcd,
                                                "chain_leapfrog_steps_per_s": BATCH * n_lf / tcd}
                if not args.no_train:             # its training step (taped K1c forward, layered reverse pass)
                    from l2hmc_amd.gauge_trainer import GaugeTrainer
                    ctr = GaugeTrainer(cdyn, lr_init=1e-4)
                    for _ in range(2):
                        ctr.train_step(x, beta)
                    torch.cuda.synchronize()
                    tq0 = time.perf_counter()
                    for _ in range(10):
                        ctr.train_step(x, beta)
                    torch.cuda.synchronize()
                    out["config"]["conv3D_arch"]["train_ms_per_step"] = 1e3 * (time.perf_counter() - tq0) / 10
                del cdyn, csmp
            except Exception as e:                 # noqa: BLE001 -- reported in the JSON line, exit code non-zero
                out["config"].setdefault("conv3D_arch", {})["error"] = repr(e)
                failed.append("conv3D_arch")

        # ---- the same dynamics at twice the chains per GPU: more than one round of 16-row workgroups, so the step
        #      kernel runs in its 32-row form (csrc/fused_traj32.hip, every weight fragment feeds two MFMAs) ----
        if rank == 0 and world == 1 and not args.no_roofline and both and not args.layered:
            try:
                B2 = 2 * BATCH
                bdyn = build_gauge(3, B2, both)
                bdyn.set_masks(dyn.mask.cpu().numpy())
                bsmp = GaugeSampler(bdyn)
                xb = torch.cat([x, x]).contiguous()
                for _ in range(10):
                    xb = bsmp.step(xb, beta)[0]
                torch.cuda.synchronize()
                tb0 = time.perf_counter()
                for _ in range(20):
                    xb = bsmp.step(xb, beta)[0]
                torch.cuda.synchronize()
                tbd = (time.perf_counter() - tb0) / 20
                fl2 = 2.0 * executed_macs_per_lf(3, True, False) * (2 * B2) * n_lf           # executed
                ref2 = 8.0 * net_macs(2 * c["L"] ** 2, 8 * c["L"] ** 2) * (2 * B2) * n_lf     # the reference graph's count
                out["config"]["twice_the_chains"] = {
                    "what": f"{B2} chains on one GPU, same dynamics: gauge_traj_fused32_kernel (32 rows per workgroup), "
                            "results bit-identical to the 16-row form",
                    "ms_per_step": 1e3 * tbd, "chain_leapfrog_steps_per_s": B2 * n_lf / tbd,
                    "whole_step_tflops": fl2 / tbd / 1e12, "whole_step_frac": fl2 / tbd / 1e12 / PEAK_F32_MFMA_TFLOPS,
                    "reference_flops_equiv_frac": ref2 / tbd / 1e12 / PEAK_F32_MFMA_TFLOPS}
                bsmp.stats.wait()

                def run2():
                    xs = xb
                    for _ in range(20):
                        xs = bsmp.step(xs, beta)[0]
                us2, n2 = profile_class(_lib.lib(), 5, run2, _lib)       # HIP events on the launch stream
                bsmp.stats.wait()
                if n2:
                    out["config"]["twice_the_chains"].update(
                        kernel_avg_us=us2, kernel_launches=n2, kernel_tflops=fl2 / (us2 * 1e-6) / 1e12,
                        kernel_frac=fl2 / (us2 * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                        kernel_reference_flops_equiv_frac=ref2 / (us2 * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS)
                del bdyn, bsmp, xb
            except Exception as e:                 # noqa: BLE001 -- reported in the JSON line, exit code non-zero
                out["config"]["twice_the_chains"] = {"error": repr(e)}
                failed.append("twice_the_chains")

        # ---- secondary: the same chains with only the direction each chain's coin selects integrated.  Bit-identical
        #      chains (tests), but it SKIPS half of the work the reference does: evidence, never the headline ----
        if rank == 0 and world == 1 and not args.no_roofline and both and not args.layered:
            try:
                sdyn = build_gauge(3, BATCH, both_directions=False)
                sdyn.set_masks(dyn.mask.cpu().numpy())
                ssmp = GaugeSampler(sdyn)
                xs_ = x.clone()
                for _ in range(10):
                    xs_ = ssmp.step(xs_, beta)[0]
                torch.cuda.synchronize()
                ts0 = time.perf_counter()
                for _ in range(20):
                    xs_ = ssmp.step(xs_, beta)[0]
                torch.cuda.synchronize()
                tsd = (time.perf_counter() - ts0) / 20
                ssmp.stats.wait()
                ex1 = 2.0 * executed_macs_per_lf(3, True, False) * BATCH * n_lf
                out["config"]["selected_only"] = {
                    "what": f"{BATCH} chains, L2HMC_PLAN_SELECTED_ONLY: each chain integrated only in the direction its "
                            "coin picks (same Philox streams, bit-identical chains whenever the other direction is finite)",
                    "skips": "unselected direction (half of the reference's work per step)",
                    "ms_per_step": 1e3 * tsd, "chain_leapfrog_steps_per_s": BATCH * n_lf / tsd,
                    "whole_step_tflops": ex1 / tsd / 1e12, "whole_step_frac": ex1 / tsd / 1e12 / PEAK_F32_MFMA_TFLOPS}
                del sdyn, ssmp, xs_
            except Exception as e:                 # noqa: BLE001 -- reported in the JSON line, exit code non-zero
                out["config"]["selected_only"] = {"error": repr(e)}
                failed.append("selected_only")

        # ---- secondary: what the per-step collective costs on this one GPU.  A one-rank RCCL group is created and the
        #      sampler issues its fused 12-byte all-reduce on the side stream after every step, exactly as at N > 1
        #      (l2hmc_amd/dist.py); timed plain / with RCCL / plain again on the same clocks ----
        if rank == 0 and world == 1 and dist is None and not args.no_roofline and not args.layered:
            try:
                import torch.distributed as tdist
                with socket.socket() as sck:
                    sck.bind(("127.0.0.1", 0))
                    port = sck.getsockname()[1]
                tdist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                         device_id=dev)
                os.environ["L2HMC_COLLECTIVES_AT_WORLD1"] = "1"
                try:
                    rsmp = GaugeSampler(dyn, dist=tdist)                       # shipped: 16 steps per all-reduce
                    rsmp1 = GaugeSampler(dyn, dist=tdist, reduce_every=1)      # one all-reduce per step
                finally:
                    del os.environ["L2HMC_COLLECTIVES_AT_WORLD1"]
                assert rsmp.stats.dist is not None and rsmp1.stats.dist is not None

                def timed(smp, n):
                    xs_ = x
                    for _ in range(20):
                        xs_ = smp.step(xs_, beta)[0]
                    smp.stats.wait()
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                    for _ in range(n):
                        xs_ = smp.step(xs_, beta)[0]
                    smp.stats.join()
                    torch.cuda.synchronize()
                    d_ = (time.perf_counter() - t0_) / n
                    smp.stats.wait()
                    return d_
                nw = max(steps, 64)
                plain_a, per_step, plain_b, grouped, plain_c = (timed(sampler, nw), timed(rsmp1, nw), timed(sampler, nw),
                                                                timed(rsmp, nw), timed(sampler, nw))
                plain = (plain_a + plain_b + plain_c) / 3
                out["config"]["world1_rccl"] = {
                    "what": "the headline step with a one-rank RCCL group issuing the collective of N > 1 (fused "
                            "all_reduce(SUM) of [sum p, sum |dQ|, n], side stream; gauge_model.py:795), same GPU, same clocks, "
                            f"{nw} steps each: plain / one all-reduce per step / plain / one all-reduce per "
                            f"{rsmp.stats.reduce_every} steps (shipped) / plain",
                    "ms_per_step_plain": [1e3 * plain_a, 1e3 * plain_b, 1e3 * plain_c],
                    "ms_per_step_one_allreduce_per_step": 1e3 * per_step,
                    "ms_per_step_shipped_grouping": 1e3 * grouped, "steps_per_allreduce_shipped": rsmp.stats.reduce_every,
                    "collective_cost_frac_of_step_per_step": (per_step - plain) / plain,
                    "collective_cost_frac_of_step_shipped": (grouped - plain) / plain}
                del rsmp, rsmp1
                tdist.destroy_process_group()
            except Exception as e:                 # noqa: BLE001 -- reported in the JSON line, exit code non-zero
                out["config"]["world1_rccl"] = {"error": repr(e)}
                failed.append("world1_rccl")

        # ---- secondary: one training step (loss + gradients + all-reduce + Adam) on the same shape; every rank
        #      takes part because the gradient bucket is all-reduced (f1/f2) ----
        if not args.no_train:
            from l2hmc_amd.gauge_trainer import GaugeTrainer
            ok, err, tr = 1, "", None
            try:
                tdyn = build_gauge(3, BATCH, both)
                tdyn._seed = 2000 + rank
                tr = GaugeTrainer(tdyn, lr_init=1e-4, dist=dist)
                saved, tr.dist = tr.dist, None
                tr.train_step(x, beta)                # local rehearsal: no collective yet
                tr.dist = saved
                torch.cuda.synchronize()
            except Exception as e:                    # noqa: BLE001 -- reported in the JSON line
                ok, err = 0, repr(e)
            if dist is not None:                      # only enter the collective steps if every rank can
                flag = torch.tensor([ok], device=dev, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item()) if ok else 0
            if ok:
                nt = 10
                tr.train_step(x, beta)
                job.barrier()
                tt0 = time.perf_counter()
                for _ in range(nt):
                    loss = tr.train_step(x, beta)[0]
                job.barrier()
                ttd = job.max_over_ranks((time.perf_counter() - tt0) / nt)
                out["config"]["train_step"] = {
                    "what": "loss + hand-written reverse pass + gradient all-reduce + Adam, 2048 x-chains and 2048 "
                            "auxiliary chains per GPU, 10 LF (gauge_model.py:799-830, :942-969)",
                    "ms_per_step": 1e3 * ttd, "train_chains_per_s": world * BATCH / ttd,
                    "grad_bucket_bytes": int(tr.grads.numel() * 4), "loss": float(loss),
                    "grad_buckets": int(tr.last_bucket_count) if tr.dist is not None else 0}
                if world == 1 and not args.no_trained_ess:
                    # the secondary metric with TRAINED networks: a short training run (same shape), then ESS/sec
                    # of the sampler with the estimator used above (untrained networks barely move a chain)
                    try:
                        xt = x
                        for i in range(250):
                            _, xo, _, _ = tr.train_step(xt, beta)
                            xt = sampler.wrap(xo)
                        tsm = GaugeSampler(tdyn)
                        for _ in range(20):
                            xt = tsm.step(xt, beta)[0]
                        torch.cuda.synchronize()
                        te0 = time.perf_counter()
                        hist2 = []
                        for _ in range(64):
                            xt = tsm.step(xt, beta)[0]
                            hist2.append(xt)
                        tsm.stats.wait()
                        torch.cuda.synchronize()
                        ted = time.perf_counter() - te0
                        ess2 = ess_of(hist2, chain_stats)
                        out["config"]["after_260_train_steps"] = {
                            "mean_accept_prob": tsm.stats.mean_accept(), "eps": float(tdyn.eps),
                            "ess_per_mcmc_step": ess2, "ess_per_sec_whole_job": ess2 * BATCH * 64 / ted,
                            "ms_per_mcmc_step": 1e3 * ted / 64}
                        del hist2, tsm
                    except Exception as e:                # noqa: BLE001
                        out["config"]["after_260_train_steps"] = {"error": repr(e)}
                        failed.append("after_260_train_steps")
            else:
                out["config"]["train_step"] = {"error": err or "another rank failed"}
                failed.append("train_step")
            del tr

    # ---- CPU baseline: op-for-op torch-CPU port of the reference graph, bounded sample, rank 0 at N = 1 ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_for(cfg, st)
        # BASELINE.md publishes no number for this metric (the reference has no benchmark): the only baseline that can
        # exist here is the one timed beside it, so the ratio is against THAT and labelled as such
        out["vs_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        out["vs_baseline_what"] = (f"value / cpu_baseline.value: the torch-CPU op-for-op port of the reference graph on "
                                   f"{out['cpu_baseline']['cores']} host threads of this box (north_star target >= 50x at "
                                   "1 GPU); BASELINE.md holds no published number for this metric")

    # ---- the other BASELINE configs at their per-GPU size (N = 1, default line only): compact table ----
    if cfg == 3 and world == 1 and not args.no_configs_table and not args.layered and both:
        table = [{"cfg": 3, "workload": c["name"], "chains": st["B"], "ms_per_step": out["ms_per_step"],
                  "value": out["value"], "tflops": res["whole_step_tflops"], "frac": res["whole_step_frac"],
                  "reference_flops_equiv_frac": res["reference_flops_equiv_frac"],
                  "kernel": out.get("roofline", {}).get("kernel"),
                  "kernel_avg_us": out.get("roofline", {}).get("avg_launch_us"),
                  "kernel_frac": out.get("roofline", {}).get("frac")}]
        del st, x, dyn
        torch.cuda.empty_cache()
        for oc, (k, w) in ((1, (50, 5)), (2, (50, 5)), (4, (5, 2)), (5, (3, 1))):
            try:
                r, s2 = run_workload(job, oc, "weak", k, w, roofline=not args.no_roofline)
                rf = r.get("roofline", {})
                table.append({"cfg": oc, "workload": r["workload"], "chains": s2["B"], "ms_per_step": r["ms_per_step"],
                              "value": r["value"], "tflops": r["whole_step_tflops"], "frac": r["whole_step_frac"],
                              "reference_flops_equiv_frac": r["reference_flops_equiv_frac"],
                              "kernel": rf.get("kernel"), "kernel_avg_us": rf.get("avg_launch_us"),
                              "kernel_frac": rf.get("frac"),
                              "kernels": [{"kernel": q["kernel"].split(" ")[0], "avg_us": q["avg_us"], "frac": q["frac"],
                                           "launches_per_step": q["launches"] / max(1, min(k, 20))}
                                          for q in rf.get("all_kernels", [])]})
                del s2, r
            except Exception as e:                # noqa: BLE001
                table.append({"cfg": oc, "error": repr(e)})
                failed.append(f"configs[{oc}]")
            torch.cuda.empty_cache()
            torch.cuda.synchronize()
        table.sort(key=lambda t: t["cfg"])
        out["configs"] = table
        out["configs_note"] = ("every BASELINE.json config on ONE GPU at its per-GPU size (configs 4 and 5: the 1/8 "
                               "shard of the 8-GPU partition, 1024 and 2048 chains); value = useful chain-LF/s, "
                               "tflops / frac = the FLOPs the kernels EXECUTE over the whole step (kept first-layer "
                               "products, position sub-updates on the columns they move) per second against the fp32 "
                               "MFMA peak, reference_flops_equiv_frac = the reference graph's count (every first layer "
                               "and every head column formed anew) over the same time -- work equivalent, not "
                               "utilisation; `python bench.py --config c [--gpus N]` runs one of them as the headline")
    if rank == 0:
        print(json.dumps(out), file=json_out, flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if failed:
        print(f"bench.py: secondary leg(s) failed: {failed}", file=sys.stderr, flush=True)
        raise SystemExit(1)


if __name__ == "__main__":
    main()
