"""Condenses the rocprofv3 passes of tools/pmc_collect.sh into the JSON bench.py reads for `roofline.traffic`
and DESIGN.md quotes:  python3 tools/pmc_summary.py gpurun_out/<tag> profiles/rNN_pmc_fused_kernel.json

Per launch of the dominant kernel (name contains KERNEL): mean of every counter, the fabric-side traffic
2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes; FETCH doubled per MI355X_MICROARCH.md: gfx950 tallies the 128-byte
requests of wide coalesced reads at 64 bytes), the L2 hit rate, and the MFMA instruction count obtained from the
SQ counters through the calibration kernel (tools/mfma_count_calib.hip: exactly 3.2768e8 v_mfma_f32_16x16x4_f32
per launch), set against the algorithmic count."""
import csv
import glob
import json
import os
import sys

KERNEL = "gauge_traj_fused_kernel"
CALIB_MFMA = 1024.0 * 10000 * 32          # tools/mfma_count_calib.hip
# bench.py workload: 4096 rows (2048 chains x 2 directions), 10 LF steps, D=128, H=512
ROWS, N_LF, D, H = 4096, 10, 128, 512
MFMA_PER_CALL_PER_GROUP = (2 * D // 4) * (H // 16) + (H // 4) * (H // 16) + (H // 4) * (3 * D // 16)   # 16 rows
ALG_MFMA = ROWS // 16 * N_LF * 4 * MFMA_PER_CALL_PER_GROUP
# per LF step the kernel skips one half first-layer product (XNet . v, second position sub-update) and one whole
# first-layer product (VNet at the start of step s+1 = VNet at the end of step s); the very first VNet call of a
# trajectory has nothing to reuse
HALF, WHOLE = (D // 4) * (H // 16), (2 * D // 4) * (H // 16)
ISSUED_MFMA = ROWS // 16 * (N_LF * (4 * MFMA_PER_CALL_PER_GROUP - HALF - WHOLE) + WHOLE)


def read(dirname, want=None):
    """{counter: (launches, mean per launch)} over the dispatches whose kernel name contains `want`."""
    acc = {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if want and want not in row["Kernel_Name"]:
                    continue
                if want == KERNEL and int(row["Grid_Size"]) not in (ROWS // 16 * 256, ROWS // 16 * 512):   # (4 or 8 waves per workgroup)
                    continue                      # only the benchmark-shaped launches (4096 rows)
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    # median: a launch that overlaps another stream's work (an asynchronous copy, the side-stream all-reduce
    # buffer) picks up that traffic in the chip-wide TCC counters; SQ counters are identical across launches
    return {k: {"launches": len(v), "median_per_launch": float(sorted(v)[len(v) // 2]),
                "min_per_launch": min(v), "max_per_launch": max(v)} for k, v in acc.items() if v}


def kernel_stats(dirname):
    out = {}
    for f in glob.glob(os.path.join(dirname, "**", "*kernel_stats.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if KERNEL in row["Name"] and "false, false>" in row["Name"]:      # sampling instantiation, GenericNet
                    out = {"name": row["Name"], "calls": int(row["Calls"]),
                           "avg_us": float(row["AverageNs"]) / 1e3, "min_us": float(row["MinNs"]) / 1e3,
                           "max_us": float(row["MaxNs"]) / 1e3, "percent_of_gpu_time": float(row["Percentage"])}
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    sq, sq2 = read(os.path.join(src, "sq"), KERNEL), read(os.path.join(src, "sq2"), KERNEL)
    tcc = read(os.path.join(src, "tcc"), KERNEL)
    fetch, write = read(os.path.join(src, "fetch"), KERNEL), read(os.path.join(src, "write"), KERNEL)
    calib = read(os.path.join(src, "calib"), "calib_mfma_kernel")
    out = {"kernel": KERNEL + "<128,512,128,false,false>", "workload": "bench.py: 4096 rows x 10 LF steps per launch",
           "sq": sq, "sq_mix": sq2, "l2_cache": tcc, "kernel_trace_stats": kernel_stats(os.path.join(src, "stats"))}
    d = out["derived"] = {}
    if "TCC_HIT_sum" in tcc:
        h, m = tcc["TCC_HIT_sum"]["median_per_launch"], tcc["TCC_MISS_sum"]["median_per_launch"]
        d["l2_hit_rate"] = h / (h + m)
    if "FETCH_SIZE" in fetch and "WRITE_SIZE" in write:
        fk, wk = fetch["FETCH_SIZE"]["median_per_launch"], write["WRITE_SIZE"]["median_per_launch"]
        out["hbm_side"] = {"FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
                           "traffic_bytes_per_launch": int(round((2 * fk + wk) * 1024)),
                           "algorithmic_bytes_per_launch": ROWS * N_LF * 0 + ROWS * (16 * D + 4),
                           "note": "separate --pmc passes; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md). "
                                   "Fabric-side requests of the per-XCD L2s, Infinity-Cache hits included: the weight "
                                   "stream (4.7 MB for both networks) is re-fetched by each of the 8 L2s as it cycles "
                                   "through them; the chain state (x, v in and out = the algorithmic bytes) crosses once."}
    if "SQ_INSTS_VALU_MFMA_MOPS_F32" in sq and "SQ_INSTS_VALU_MFMA_MOPS_F32" in calib:
        per_mfma_mops = calib["SQ_INSTS_VALU_MFMA_MOPS_F32"]["median_per_launch"] / CALIB_MFMA
        per_mfma_busy = calib["SQ_VALU_MFMA_BUSY_CYCLES"]["median_per_launch"] / CALIB_MFMA
        n_mops = sq["SQ_INSTS_VALU_MFMA_MOPS_F32"]["median_per_launch"] / per_mfma_mops
        n_busy = sq["SQ_VALU_MFMA_BUSY_CYCLES"]["median_per_launch"] / per_mfma_busy
        d["mfma_calibration"] = {
            "known_mfma_per_launch": CALIB_MFMA,
            "MOPS_F32_per_v_mfma_f32_16x16x4_f32": per_mfma_mops,
            "MFMA_BUSY_CYCLES_per_v_mfma_f32_16x16x4_f32": per_mfma_busy,
            "note": "counter_defs.yaml defines MOPS as FLOPs / 512, i.e. 4 per 16x16x4 f32 MFMA (2048 FLOP); the "
                    "calibration kernel says what the counter really returns on this chip / profiler"}
        d["mfma_instructions_per_launch"] = {
            "from_MOPS_counter": n_mops, "from_BUSY_CYCLES_counter": n_busy,
            "algorithmic (SURVEY 8 sizes table, nothing skipped)": ALG_MFMA,
            "issued by design (recurring first-layer products reused)": ISSUED_MFMA,
            "counter / issued-by-design": n_mops / ISSUED_MFMA}
    ks = out["kernel_trace_stats"]
    if ks:
        t = ks["avg_us"] * 1e-6
        d["tflops_algorithmic"] = ALG_MFMA * 2048 / t / 1e12
        d["tflops_issued"] = ISSUED_MFMA * 2048 / t / 1e12
        d["frac_of_157.3_algorithmic"] = d["tflops_algorithmic"] / 157.3
        d["frac_of_157.3_issued"] = d["tflops_issued"] / 157.3
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(d, indent=1))


if __name__ == "__main__":
    main()
