#!/usr/bin/env python
"""Throughput of the L2HMC hot path on MI355X: chain-leapfrog-steps per second.

    python bench.py --gpus N --steps K --warmup W

One "step" is one `apply_transition` (dynamics/gauge_dynamics.py:195-259) over a
batch of synthetic chains: momenta, direction coin and MH uniform drawn on the
device, BOTH directions integrated (as the reference does), accept/reject,
wrap to [0, 2pi) and per-step observables -- all resident in HBM.  Workload:
BASELINE.json configs[2], the configuration its metric is quoted on (2D U(1)
8x8, beta 2.0, batch 2048 per GPU, 10 leapfrog steps, GenericNet H=512, fp32).

N > 1: a bare `python bench.py --gpus N` starts its own N ranks (child processes through
torch.distributed.run, before this process touches the GPU); under torch.distributed.run (WORLD_SIZE set) the
process is a rank.  One rank per GPU: chains are
independent, so every rank integrates its own 2048 chains (weak scaling) and
the only exchange is one small RCCL all-reduce of the per-step scalar sums
(accept probability, |dQ|, count), issued on a side stream.

Prints ONE JSON line (rank 0).  `value` counts USEFUL chain-leapfrog steps
(B*N_LF per transition); the executed count is twice that and is what the
roofline FLOPs use.
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

L, BETA, EPS, N_LF, BATCH, HID_MULT = 8, 2.0, 0.25, 10, 2048, 4
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, chip-level parameters


def net_macs(D, H):
    """generic net MACs per call per chain = 2DH + 2H + H^2 + 3HD."""
    return 2 * D * H + 2 * H + H * H + 3 * H * D


def build_dynamics(batch, both_directions=True, arch='generic'):
    """The product's own constructor (reference initialisation, generic_net.py:39-90) under fixed NumPy seeds;
    returns the weights/masks as NumPy too, for the cpu_baseline leg."""
    import l2hmc_amd as la
    np.random.seed(106)
    lat = la.GaugeLattice(L, L, 2, 'U1', num_samples=batch, rand=False)
    dyn = la.GaugeDynamics(lat, lat.get_energy_function(), eps=EPS, hmc=False, network_arch=arch, num_steps=N_LF,
                           eps_trainable=True, data_format='channels_last', both_directions=both_directions)
    if arch != 'generic':
        return dyn, None, None, None
    xp = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in dyn.position_fn.state_dict().items()}
    vp = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in dyn.momentum_fn.state_dict().items()}
    return dyn, xp, vp, dyn.mask.cpu().numpy()


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


def rendezvous_only(world, rank):
    """The cross-rank plumbing of the timed region without the GPU work (CPU test hook): barrier on both sides,
    MAX over ranks of the elapsed time, rank 0 prints one line."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    ranks = torch.ones(1, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(ranks, op=dist.ReduceOp.SUM)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": world, "ranks_seen": int(ranks.item()),
                          "max_elapsed_s": float(t.item())}), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--selected-only", action="store_true",
                    help="integrate only the direction each chain's coin selects (not the reference's work)")
    ap.add_argument("--layered", action="store_true", help="use the layer-by-layer kernels (no fused trajectory)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step timing")
    ap.add_argument("--no-trained-ess", action="store_true",
                    help="skip the 250-step training run + ESS/sec of the trained sampler (N = 1 only)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="test hook (CPU, gloo): start the ranks, barrier, max-over-ranks reduction, print "
                         "{'rendezvous_only': true, ...} and exit -- no measurement, no GPU")
    args = ap.parse_args()

    cmd = launch_command(args.gpus, os.environ, sys.argv[1:])
    if cmd is not None:                           # before anything initialises the GPU in this process
        raise SystemExit(self_launch(cmd))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (start it as `python bench.py "
                         f"--gpus N` or through torch.distributed.run with --nproc-per-node N)")
    if args.rendezvous_only:
        raise SystemExit(rendezvous_only(world, rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
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
    both = not args.selected_only
    dyn, xp, vp, masks = build_dynamics(BATCH, both)
    dyn._seed = 1000 + rank                       # independent chains per rank
    dyn.fused = not args.layered
    D = 2 * L * L
    x = torch.empty(BATCH, D, device=dev)
    _lib.check(_lib.lib().l2hmc_fill_uniform(x.data_ptr(), x.numel(), 103 + rank, 0, _lib.stream_ptr()))
    x.mul_(2 * np.pi)                             # hot start, lattice.py:131-135
    sampler = GaugeSampler(dyn, dist=dist)       # transition + device-side wrap + per-step observables
    stats = sampler.stats                        # one fused all-reduce of the step's scalar sums

    def step(x):
        return sampler.step(x, BETA)[0]

    # untimed pre-warm, in ADDITION to the W warm-up steps: after the idle seconds of start-up the GPU clock takes
    # ~20 ms of work to ramp (tools/warmup_probe.py -> profiles/r02_warmup_probe.txt: the first 10 steps after an
    # idle period run 2.14 -> 1.70 ms, steady state 1.68), which a 5-step warm-up would leak into a 20-step timing
    # (a FIXED count: every step carries a collective when sharded, so all ranks must run the same number)
    prewarm = 150
    for i in range(prewarm):
        x = step(x)
        if i % 16 == 15:
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        x = step(x)
    stats.wait()                                  # the untimed steps' scalars are folded before the clock starts
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    history = []                                  # references only: no device work in the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = step(x)
        history.append(x)
    stats.join()                                  # every step's all-reduce is ordered before the synchronize below
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats.wait()                                  # host-side bookkeeping of the step scalars: not part of a step
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    accept_rate = stats.mean_accept()

    useful = world * BATCH * N_LF * args.steps
    value = useful / dt
    out = {
        "metric": "leapfrog-steps/sec (whole node), 8x8 U(1) batch 2048, 10 LF",
        "value": value, "unit": "chain-leapfrog-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "prewarm_steps_untimed": prewarm, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "U(1) 8x8 lattice, beta=2.0, batch 2048 per GPU, 10 LF steps, GenericNet H=512 "
                               "(BASELINE.json configs[2])",
                   "global_batch": world * BATCH, "lattice": [L, L], "num_steps": N_LF, "eps": EPS, "beta": BETA,
                   "directions_integrated": 2 if both else 1,
                   "executed_chain_lf_per_step": (2 if both else 1) * BATCH * N_LF * world,
                   "parallelism": f"chains sharded over {world} GPU(s), weights replicated",
                   "mean_accept_prob": accept_rate},
    }

    # ---- secondary metric: ESS/sec with the reference's estimator (func_utils.py:45-54,114-120) on
    #      (cos, sin) of the links of rank 0's chains over the timed steps, normalised by the lag-0 term
    if rank == 0 and args.steps >= 8:
        X = torch.stack(history[-min(args.steps, 64):]).cpu().numpy()
        feats = np.concatenate([np.cos(X), np.sin(X)], axis=2)
        feats = feats - feats.mean(axis=(0, 1), keepdims=True)
        A = chain_stats.acl_spectrum(feats, 1.0)
        ess = float(chain_stats.ESS(A / A[0]))
        out["config"]["ess_per_mcmc_step"] = ess
        out["config"]["ess_per_sec_whole_job"] = ess * world * BATCH * args.steps / dt
    history.clear()

    # ---- roofline of the dominant kernel, HIP events on the launch stream ----
    # (every rank runs the profiled steps -- they contain the per-step collective -- only rank 0 reports)
    if not args.no_roofline:
        Lh = _lib.lib()
        rows = (2 if both else 1) * BATCH
        Hd = HID_MULT * D
        per_class = {}
        # the ESS estimate above was host work (seconds of idle GPU): ramp the clock again before the event timing
        xs = x
        for i in range(100):
            xs = step(xs)
        stats.wait()
        # algorithmic FLOPs per launch: SURVEY.md 8 sizes table (8 x net MACs per chain-LF step)
        for cls, name, flops in ((5, "gauge_traj_fused_kernel<128,512> (whole MCMC step in one launch: draws, both "
                                     "trajectories, mix / MH, observables, wrap)",
                                  8.0 * net_macs(D, Hd) * rows * N_LF),
                                 (1, "gemm_relu_kernel<64,1> (first layer)", 2.0 * rows * Hd * 2 * D),
                                 (2, "gemm_relu_kernel<64,2> (hidden layer)", 2.0 * rows * Hd * Hd),
                                 (3, "heads_kernel (S/T/Q + update)", 2.0 * rows * 3 * D * Hd)):
            _lib.check(Lh.l2hmc_profile_begin(cls))
            xs = x
            for _ in range(args.steps):
                xs = step(xs)
            ms, n = C.c_double(), C.c_int64()
            _lib.check(Lh.l2hmc_profile_end(C.byref(ms), C.byref(n)))
            per_class[cls] = dict(kernel=name, launches=n.value, avg_us=1e3 * ms.value / max(n.value, 1),
                                  flops_per_launch=flops,
                                  tflops=flops / (ms.value / max(n.value, 1) * 1e-3) / 1e12 if n.value else 0.0)
        stats.wait()
        per_class = {k: v for k, v in per_class.items() if v["launches"]}
        dom = max(per_class.values(), key=lambda d: d["avg_us"] * d["launches"])
        # The step kernel also draws the momenta and finishes the step (mix, MH, observables, wrap).  The same kernel
        # launched for the trajectories alone (l2hmc_gauge_trajectory: same rows, same FLOPs, no step prologue /
        # epilogue) separates the integrator's MFMA efficiency from that fixed per-step work.
        traj_only = None
        if not args.layered:
            x2 = torch.cat([x, x]).contiguous()
            v2 = dyn._normal(tuple(x2.shape))
            for _ in range(3):
                dyn.transition_kernel(x2, BETA, forward=True, momentum=v2)
            _lib.check(Lh.l2hmc_profile_begin(5))
            for _ in range(20):
                dyn.transition_kernel(x2, BETA, forward=True, momentum=v2)
            ms, n = C.c_double(), C.c_int64()
            _lib.check(Lh.l2hmc_profile_end(C.byref(ms), C.byref(n)))
            if n.value:
                us = 1e3 * ms.value / n.value
                tf = 8.0 * net_macs(D, Hd) * x2.shape[0] * N_LF / (us * 1e-6) / 1e12
                traj_only = {"what": "the same kernel launched for the two trajectories alone (l2hmc_gauge_trajectory, "
                                     f"{x2.shape[0]} rows x {N_LF} LF): no draws, no mix / MH / observables / wrap",
                             "avg_launch_us": us, "tflops": tf, "frac": tf / PEAK_F32_MFMA_TFLOPS}
        # fabric-side bytes per launch of the fused kernel at this exact shape: read from the newest committed PMC
        # summary (tools/pmc_collect.sh + tools/pmc_summary.py -> profiles/rNN_pmc_fused_kernel.json); null if the
        # dominant kernel is another one or no summary is committed -- never a literal
        traffic, traffic_src = None, None
        if dom["kernel"].startswith("gauge_traj_fused") and both:
            traffic, traffic_src = committed_traffic()
        out["roofline"] = {"bound": "mfma", "achieved": dom["tflops"], "peak": PEAK_F32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": dom["tflops"] / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                           "traffic_unit": "bytes per launch, fabric side: 2 x FETCH_SIZE + WRITE_SIZE from separate "
                                           f"rocprofv3 --pmc passes ({traffic_src})",
                           "kernel": dom["kernel"], "avg_launch_us": dom["avg_us"],
                           "algorithmic_flops_per_launch": dom["flops_per_launch"],
                           "all_kernels": list(per_class.values()), "trajectory_only": traj_only,
                           "whole_step_tflops": (2 if both else 1) * BATCH * N_LF * 8 * net_macs(D, Hd)
                           / (dt / args.steps) / 1e12}

    # ---- the same workload with the reference's CLI-default architecture (conv3D, gauge_model.py:2307) ----
    if rank == 0 and world == 1 and not args.no_roofline:
        cdyn = build_dynamics(BATCH, both, arch='conv3D')[0]
        cdyn.set_masks(masks)
        csmp = GaugeSampler(cdyn)
        xc = x.clone()
        for _ in range(3):
            xc = csmp.step(xc, BETA)[0]
        torch.cuda.synchronize()
        tc0 = time.perf_counter()
        for _ in range(20):
            xc = csmp.step(xc, BETA)[0]
        torch.cuda.synchronize()
        tcd = (time.perf_counter() - tc0) / 20
        out["config"]["conv3D_arch"] = {"net": "ConvNet3D F=8, H=256", "ms_per_step": 1e3 * tcd,
                                        "chain_leapfrog_steps_per_s": BATCH * N_LF / tcd}
        if not args.no_train:
            try:                                   # its training step (taped K1c forward, layered reverse pass)
                from l2hmc_amd.gauge_trainer import GaugeTrainer
                ctr = GaugeTrainer(cdyn, lr_init=1e-4)
                for _ in range(2):
                    ctr.train_step(x, BETA)
                torch.cuda.synchronize()
                tq0 = time.perf_counter()
                for _ in range(10):
                    ctr.train_step(x, BETA)
                torch.cuda.synchronize()
                out["config"]["conv3D_arch"]["train_ms_per_step"] = 1e3 * (time.perf_counter() - tq0) / 10
            except Exception as e:                 # noqa: BLE001 -- reported in the JSON line
                out["config"]["conv3D_arch"]["train_error"] = repr(e)

    # ---- secondary: BASELINE.json configs[1], the 2-D mixture of Gaussians (mog_model.py): 4096 chains per GPU,
    #      10 LF steps, `propose` = forward + backward trajectories of every chain in one launch + mix/accept.
    #      Chains are independent: every rank runs its own 4096 (no collective); rank 0 reports its own time.
    try:
        import l2hmc_amd as la
        np.random.seed(106)
        gmm = la.GMM([np.array([1., 0.]), np.array([0., 1.])], [0.025 * np.eye(2)] * 2, [0.5, 0.5])
        mdyn = la.Dynamics(2, gmm.get_energy_function(), trajectory_length=10, eps=0.1,
                           net_factory=lambda d, scope, factor: la.network(d, scope, factor, num_nodes=50))
        mx = torch.randn(4096, 2, device=dev)
        for _ in range(5):
            mx = la.propose(mx, mdyn, do_mh_step=True)[3][0]
        torch.cuda.synchronize()
        tm0 = time.perf_counter()
        for _ in range(50):
            mx = la.propose(mx, mdyn, do_mh_step=True)[3][0]
        torch.cuda.synchronize()
        tmd = (time.perf_counter() - tm0) / 50
        out["config"]["mog_cfg2"] = {
            "workload": "2-D mixture of Gaussians, 4096 chains per GPU, 10 LF steps, MLP H=50 (BASELINE.json configs[1])",
            "ms_per_propose": 1e3 * tmd, "chain_leapfrog_steps_per_s": world * 4096 * 10 / tmd,
            # 4 network calls per LF step and direction; MACs per call and chain: two 2 x 50 input layers, the time
            # layer, 50 x 50 hidden, three 50 x 2 heads
            "algorithmic_tflops": 2 * 4096 * 40 * 2 * (2 * 2 * 50 + 2 * 50 + 50 * 50 + 3 * 2 * 50) / tmd / 1e12,
            "frac_of_fp32_mfma_peak": 2 * 4096 * 40 * 2 * (2 * 2 * 50 + 2 * 50 + 50 * 50 + 3 * 2 * 50) / tmd / 1e12
            / PEAK_F32_MFMA_TFLOPS,
            "bound": "latency (one launch per propose: Philox draws, both trajectories, mix and MH in the kernel; a "
                     "wave walks 40 dependent network calls for its 8 chains x 2 directions -- 512 waves on 1024 "
                     "SIMDs; hidden layer and heads on 16x16x4 fp32 MFMAs with register-resident weights, DESIGN.md K4)"}
    except Exception as e:                        # noqa: BLE001
        out["config"]["mog_cfg2"] = {"error": repr(e)}

    # ---- secondary: one training step (loss + gradients + all-reduce + Adam) on the same shape; every rank
    #      takes part because the gradient bucket is all-reduced (f1/f2) ----
    if not args.no_train:
        from l2hmc_amd.gauge_trainer import GaugeTrainer
        ok, err, tr = 1, "", None
        try:
            tdyn = build_dynamics(BATCH, both)[0]
            tdyn._seed = 2000 + rank
            tr = GaugeTrainer(tdyn, lr_init=1e-4, dist=dist)
            saved, tr.dist = tr.dist, None
            tr.train_step(x, BETA)                # local rehearsal: no collective yet
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
            tr.train_step(x, BETA)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            tt0 = time.perf_counter()
            for _ in range(nt):
                loss = tr.train_step(x, BETA)[0]
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            ttd = (time.perf_counter() - tt0) / nt
            if dist is not None:
                t = torch.tensor([ttd], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ttd = float(t.item())
            out["config"]["train_step"] = {
                "what": "loss + hand-written reverse pass + gradient all-reduce + Adam, 2048 x-chains and 2048 "
                        "auxiliary chains per GPU, 10 LF (gauge_model.py:799-830, :942-969)",
                "ms_per_step": 1e3 * ttd, "train_chains_per_s": world * BATCH / ttd,
                "grad_bucket_bytes": int(tr.grads.numel() * 4), "loss": float(loss),
                "grad_buckets": int(tr.last_bucket_count) if tr.dist is not None else 0}
            if world == 1 and not args.no_trained_ess:
                # the secondary metric with TRAINED networks: a short training run (same shape), then ESS/sec of
                # the sampler with the estimator used above (untrained networks barely move a chain)
                try:
                    xt = x
                    for i in range(250):
                        _, xo, _, _ = tr.train_step(xt, BETA)
                        xt = sampler.wrap(xo)
                    tsm = GaugeSampler(tdyn)
                    for _ in range(20):
                        xt = tsm.step(xt, BETA)[0]
                    torch.cuda.synchronize()
                    te0 = time.perf_counter()
                    hist2 = []
                    for _ in range(64):
                        xt = tsm.step(xt, BETA)[0]
                        hist2.append(xt)
                    tsm.stats.wait()
                    torch.cuda.synchronize()
                    ted = time.perf_counter() - te0
                    X2 = torch.stack(hist2).cpu().numpy()
                    f2 = np.concatenate([np.cos(X2), np.sin(X2)], axis=2)
                    f2 = f2 - f2.mean(axis=(0, 1), keepdims=True)
                    A2 = chain_stats.acl_spectrum(f2, 1.0)
                    ess2 = float(chain_stats.ESS(A2 / A2[0]))
                    out["config"]["after_260_train_steps"] = {
                        "mean_accept_prob": tsm.stats.mean_accept(), "eps": float(tdyn.eps),
                        "ess_per_mcmc_step": ess2, "ess_per_sec_whole_job": ess2 * BATCH * 64 / ted,
                        "ms_per_mcmc_step": 1e3 * ted / 64}
                except Exception as e:                # noqa: BLE001
                    out["config"]["after_260_train_steps"] = {"error": repr(e)}
        else:
            out["config"]["train_step"] = {"error": err or "another rank failed"}

    # ---- CPU baseline: op-for-op torch-CPU port of the reference graph, bounded sample ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.cpu_baseline import time_cpu_baseline
        sample_b = BATCH          # the whole per-GPU batch: ~0.5 s per call with a sane thread count
        cb = time_cpu_baseline(L, L, N_LF, EPS, BETA, sample_b, xp, vp, masks, budget_s=15.0)
        out["cpu_baseline"] = {"value": cb["value"], "unit": "chain-leapfrog-steps/s", "cores": cb["cores"],
                               "kind": "port",
                               "sample": f"{cb['calls']} timed apply_transition calls (both directions) on "
                                         f"{sample_b} chains (the full per-GPU batch), same lattice/net/LF config, torch-CPU fp32, "
                                         f"median {cb['seconds']:.3f} s per call; thread count probed for the best "
                                         f"throughput ({cb['cores']} of {cb['cpus_available']} usable CPUs)"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
