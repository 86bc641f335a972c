#!/usr/bin/env python3
"""Copies what tools/collect_profiles.sh measured (gpurun_out/r01) into the tracked profiles/ tree:

  profiles/r01/bench_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary of `python bench.py`
  profiles/r01/bench_pmc_counters.json   per-launch averages of every PMC counter collected (one pass per group)
  profiles/pmc_traffic.json              HBM-side bytes per launch of the dominant kernel, read by bench.py
  profiles/r01/{bench_default.json, configs_1gpu.json, ubench_*.txt, power_clocks.txt}
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "r05")
SRC = os.path.join(ROOT, "gpurun_out", ROUND)
DST = os.path.join(ROOT, "profiles", ROUND)
KERNEL = "dist_kernel_v2"


def main():
    os.makedirs(DST, exist_ok=True)
    stats = (glob.glob(os.path.join(SRC, "kt", "*", "kt_kernel_stats.csv")) + glob.glob(os.path.join(SRC, "kt", "kt_kernel_stats.csv")))[0]
    shutil.copy(stats, os.path.join(DST, "bench_kernel_stats.csv"))
    if os.path.exists(os.path.join(SRC, "bench_kernel_rows.csv")):      # the same trace split by (kernel, grid)
        shutil.copy(os.path.join(SRC, "bench_kernel_rows.csv"), DST)
    line = [l for l in open(os.path.join(SRC, "bench.json")) if l.startswith("{")][-1]
    open(os.path.join(DST, "bench_default.json"), "w").write(line)
    if os.path.exists(os.path.join(SRC, "configs.json")):
        shutil.copy(os.path.join(SRC, "configs.json"), os.path.join(DST, "configs_1gpu.json"))
    ktc = os.path.join(SRC, "ktc", "c_kernel_stats.csv")
    if os.path.exists(ktc):       # every kernel of tools/measure_configs.py (kernel 2, sweeps, kNN, ...)
        shutil.copy(ktc, os.path.join(DST, "configs_kernel_stats.csv"))
    for f in glob.glob(os.path.join(SRC, "ubench_*.txt")) + [os.path.join(SRC, x) for x in (
            "power_clocks.txt", "ab_host.txt", "ab_host_parts.txt", "knn_from_tiles.txt", "two_ranks_one_gpu.json",
            "smalljob.txt", "smalljob_two_pass.txt", "stall_hunt.txt", "ab_pinning.txt", "latency_table.txt", "time_wide.txt",
            "k2_trace.txt")]:
        if os.path.exists(f):
            shutil.copy(f, DST)
    k2 = glob.glob(os.path.join(SRC, "k2", "*", "k2_kernel_stats.csv")) + glob.glob(os.path.join(SRC, "k2", "k2_kernel_stats.csv"))
    if k2:       # kernel 2 on rotating (cold-cache) matrices: per-kernel averages
        shutil.copy(k2[0], os.path.join(DST, "kernel2_kernel_stats.csv"))
    counters = {}
    kname = None
    for d in sorted(glob.glob(os.path.join(SRC, "pmc_*"))):
        for f in glob.glob(os.path.join(d, "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if KERNEL in r["Kernel_Name"]:
                    kname = r["Kernel_Name"].split("(")[0]
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for c, v in agg.items():
                counters[c] = {"avg_per_launch": sum(v) / len(v), "launches": len(v), "pass": os.path.basename(d)}
    bench = json.loads(line)
    doc = {"command": "rocprofv3 --pmc <counters> --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu "
                      " (one pass per counter group; tools/collect_profiles.sh)",
           "kernel": "%s  (%d genomes self, %d pairs per launch)" % (kname, bench["config"]["n_genomes"], bench["config"]["pairs"]),
           "counters": dict(sorted(counters.items()))}
    json.dump(doc, open(os.path.join(DST, "bench_pmc_counters.json"), "w"), indent=1)
    fetch, write = counters["FETCH_SIZE"]["avg_per_launch"], counters["WRITE_SIZE"]["avg_per_launch"]
    import datetime
    sys.path.insert(0, ROOT)
    from poppunk_amd import _lib
    traffic = {"n%d" % bench["config"]["n_genomes"]: (2.0 * fetch + write) * 1024.0,
               # the sources the measured library was built from (ppk_version()): bench.py uses the figure only
               # with a library of the same hash
               "src_hash": _lib.source_hash(),
               "source": "profiles/%s/bench_pmc_counters.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over "
                         "bench.py, %s)" % (ROUND, datetime.date.today().isoformat()),
               "how": "2 x FETCH_SIZE (gfx950 rocprofv3 reports half the bytes of a 16 B/lane stream: "
                      "MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KB -> bytes, averaged per launch of "
                      "dist_kernel_v2; separate --pmc passes",
               "fetch_size_kb": fetch, "write_size_kb": write}
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    # derived per-launch figures of the dominant kernel
    g = lambda k: counters[k]["avg_per_launch"] if k in counters else float("nan")
    pairs = bench["config"]["pairs"]
    cyc = g("GRBM_GUI_ACTIVE") / 8.0          # summed over the 8 XCDs
    derived = {
        "valu_instructions_per_pair": g("SQ_INSTS_VALU") * 64 / pairs,
        "valu_instructions_per_pair_algorithmic": 2400,
        "lds_instructions_per_pair": g("SQ_INSTS_LDS") * 64 / pairs,
        "l2_hit_rate": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")),
        "lds_array_busy_fraction": g("SQ_LDS_IDX_ACTIVE") / 256.0 / cyc,
        "lds_bank_conflict_cycles": g("SQ_LDS_BANK_CONFLICT"),
        "valu_wave_instructions_per_clk_per_simd": g("SQ_INSTS_VALU") / 1024.0 / cyc,
        "wave_wait_any_fraction": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
        "wave_wait_inst_any_fraction": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
        "gpu_cycles_per_launch": cyc,
        "note": "from bench_pmc_counters.json (separate --pmc passes over python bench.py --steps 5); "
                "cycles = GRBM_GUI_ACTIVE / 8 XCDs; a pure v_bitop3 stream issues ~0.41 wave-instr/clk/SIMD",
    }
    json.dump(derived, open(os.path.join(DST, "derived_metrics.json"), "w"), indent=1)
    for r in csv.DictReader(open(os.path.join(DST, "bench_kernel_stats.csv"))):
        if KERNEL in r["Name"]:
            print("rocprof: %s calls=%s avg=%.4f ms" % (KERNEL, r["Calls"], float(r["AverageNs"]) / 1e6))
    print("bench : kernel_ms=%.4f ms_per_step=%.4f value=%.3f Gpairs/s" %
          (bench["roofline"]["kernel_ms"], bench["ms_per_step"], bench["value"] / 1e9))
    print("traffic per launch: %.3f GB" % (traffic["n%d" % bench["config"]["n_genomes"]] / 1e9))


if __name__ == "__main__":
    sys.exit(main())
