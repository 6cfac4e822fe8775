"""Diagnostic (not shipped): the per-step collective on ONE GPU under `rocprofv3 --kernel-trace`.

    rocprofv3 --kernel-trace -d OUT -o t --output-format csv -- python3 tools/world1_rccl_trace.py run [reduce_every]
    python3 tools/world1_rccl_trace.py summarise OUT/..._kernel_trace.csv

`run`: a one-rank RCCL group, GaugeSampler at the headline shape issuing its fused all-reduce of [sum p, sum |dQ|, n]
every `reduce_every` steps on the side stream (l2hmc_amd/dist.py), 48 MCMC steps.  `summarise`: where the collective's
kernel sits between the step kernels and what it does to the step that follows."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(reduce_every):
    import numpy as np
    import torch
    import torch.distributed as dist
    import l2hmc_amd as la
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    os.environ["L2HMC_COLLECTIVES_AT_WORLD1"] = "1"
    np.random.seed(106)
    lat = la.GaugeLattice(8, 8, 2, 'U1', num_samples=2048, rand=False)
    dyn = la.GaugeDynamics(lat, lat.get_energy_function(), eps=0.25, hmc=False, network_arch='generic', num_steps=10,
                           eps_trainable=True, data_format='channels_last')
    smp = la.GaugeSampler(dyn, dist=dist, reduce_every=reduce_every)
    assert smp.stats.dist is not None
    x = torch.rand(2048, 128, device="cuda") * 6.28
    for _ in range(64):
        x = smp.step(x, 2.0)[0]
    smp.stats.wait()
    torch.cuda.synchronize()
    for _ in range(48):
        x = smp.step(x, 2.0)[0]
    smp.stats.wait()
    torch.cuda.synchronize()
    dist.destroy_process_group()


def summarise(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    steps = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "gauge_traj_fused_kernel" in r["Kernel_Name"]]
    coll = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows
            if "ccl" in r["Kernel_Name"].lower() or "AllReduce" in r["Kernel_Name"]]
    steps, coll = steps[-48:], [c for c in coll if c[0] >= steps[-48][0]] if len(steps) >= 48 else coll
    dur = [(e - s) / 1e3 for s, e in steps]
    print(f"step kernels: {len(steps)}, median {sorted(dur)[len(dur) // 2]:.1f} us, min {min(dur):.1f}, max {max(dur):.1f}")
    print(f"collective kernels in the same window: {len(coll)}" + (f"  ({coll[0][2][:70]})" if coll else ""))
    for cs, ce, _ in coll[:12]:
        prev = max((i for i, (s, e) in enumerate(steps) if e <= cs + 2000), default=None)
        inside = [i for i, (s, e) in enumerate(steps) if s < ce and e > cs]
        msg = f"  collective {(ce - cs) / 1e3:6.1f} us"
        if prev is not None:
            msg += f" | starts {(cs - steps[prev][1]) / 1e3:+7.1f} us after step {prev} ends"
        if inside:
            i = inside[0]
            msg += f" | overlaps step {i}: that step took {dur[i]:.1f} us"
        print(msg)
    with_c = [dur[i] for i, (s, e) in enumerate(steps) if any(cs < e and ce > s for cs, ce, _ in coll)]
    without = [d for d in dur if d not in with_c]
    if with_c and without:
        print(f"steps overlapping a collective: {len(with_c)}, median {sorted(with_c)[len(with_c) // 2]:.1f} us; "
              f"others: {len(without)}, median {sorted(without)[len(without) // 2]:.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    else:
        summarise(sys.argv[2])
