#!/usr/bin/env python3
"""Benchmark of the distance hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of kernel 1 over the whole synthetic database: all
self-vs-self core/accessory distances, sketches already resident in HBM,
output [n_pairs, 2] float32 in PopPUNK row order on rank 0.

  N = 1 : BASELINE configs[2]'s workload on one GPU -- 10 000 synthetic genomes,
          s = 1024 (sketchsize64 16, bbits 14), k = 13,17,21,25,29 -> 49 995 000 pairs.
  N > 1 : the SAME 10 000 genomes (strong scaling, BASELINE config 3): the pair space is band-split
          over the ranks and the distance blocks are gathered to rank 0 with grouped RCCL
          send/recv inside the timed region, pipelined under the compute in --chunks sub-bands
          (bands re-cut from measured per-rank rates during the untimed set-up: the root's band
          needs no transfer; --even-bands).  `--weak` instead grows the database to
          round(10 000 * sqrt(N)) genomes (N x 49 995 000 pairs per step).
          Started plainly (`python bench.py --gpus 4`) the script re-executes itself under
          torch.distributed.run, one rank per GPU; started by torch.distributed.run it uses the
          ranks it is given.

Rank 0 prints ONE JSON line (see the driver contract) carrying
  `roofline`     the dominant kernel, HIP-event timed inside libppk_hip.so on its own stream,
                 against the integer-VALU roof that binds it (and, for reference, the HBM figures);
  `cpu_baseline` the oracle, timed on this host on a bounded sample (N = 1 only);
  `host_call`    the PCIe-inclusive ppk_query call PopPUNK itself makes (N = 1 only; never `value`);
  `config5`      BASELINE config 5's shape -- 100 000 genomes self, fused distance -> boundary ->
                 edge list, only the edge lists gathered (engine.edges_sharded) -- timed separately
                 after the headline steps; `multi_gpu` (N > 1) compute vs gather time.
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PAIR = 17928          # 2*5*16*14*8 operand bytes + 8 B result
VALU_OPS_PER_PAIR = 2400             # 5 k * 16 blocks * (28 v_bitop3 + 2 v_bcnt) lane-ops
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9   # CUs * SIMDs * lanes/clk * max clock = 78.6e12 lane-ops/s
VALU_MEASURED_LANE_OPS = 55.6e12     # tools/ubench_valu.hip: v_bitop3_b32 v,v,v sustained on MI355X


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20,
                    help="untimed steps (the GPU clock needs ~50 ms of load to ramp up)")
    ap.add_argument("--genomes", dest="n", type=int, default=10000, help="genomes (at 1 GPU with --weak)")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: grow the database to genomes*sqrt(N) (constant pairs per GPU) instead "
                         "of the default strong scaling on the same genomes")
    ap.add_argument("--strong", action="store_true", help="(default; kept for compatibility)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-host-call", action="store_true", help="skip the host_call leg")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 (fused edge list) leg")
    ap.add_argument("--config5-genomes", type=int, default=100000)
    ap.add_argument("--config5-steps", type=int, default=3)
    ap.add_argument("--spinup-ms", type=float, default=200.0,
                    help="untimed clock spin-up before the warm-up steps (0 disables)")
    ap.add_argument("--chunks", type=int, default=4,
                    help="sub-bands per rank: the gather of chunk c overlaps the compute of c+1")
    ap.add_argument("--even-bands", action="store_true",
                    help="N > 1: keep equal bands (default: re-cut them from measured rates during "
                         "the untimed set-up, so that the root, whose band needs no transfer, takes more)")
    return ap.parse_args()


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` from a plain shell: become N ranks, one per GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota
    (a container that sees 256 hardware threads but has a 16-CPU quota is throttled beyond 16:
    tools/cpu_scaling.py measured 277 Mpairs/s in a 16 ms burst on 64 threads, 27 Mpairs/s
    sustained on 128)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def cpu_baseline(sk, kmers, tbl, seconds):
    """Time the CPU oracle (oracle/ppk_oracle.c, `port`) on a bounded self-vs-self sample,
    repeated until about `seconds` of wall time has been spent, on every CPU the process may use."""
    from oracle import oracle
    threads = max(1, min(usable_cpus(), oracle.max_threads()))
    probe = min(sk.shape[0], 400)
    t0 = time.perf_counter()
    oracle.query(sk[:probe], None, kmers, 16, 14, tbl, threads=threads)
    dt = max(time.perf_counter() - t0, 1e-4)
    rate = probe * (probe - 1) / 2 / dt
    n_s = int(min(sk.shape[0], max(probe, (2 * rate * seconds) ** 0.5)))
    pairs = n_s * (n_s - 1) // 2
    reps, total = 0, 0.0
    while total < seconds and reps < 50:
        t0 = time.perf_counter()
        oracle.query(sk[:n_s], None, kmers, 16, 14, tbl, threads=threads)
        total += time.perf_counter() - t0
        reps += 1
    # one thread, on a smaller sample (~2 s): the per-core rate (SURVEY.md 8d asks for both)
    n_1 = int(min(n_s, max(200, (2 * (rate / max(threads, 1)) * 2.0) ** 0.5)))
    t0 = time.perf_counter()
    oracle.query(sk[:n_1], None, kmers, 16, 14, tbl, threads=1)
    one = n_1 * (n_1 - 1) / 2 / max(time.perf_counter() - t0, 1e-6)
    return {"value": pairs * reps / total, "unit": "pairs/s", "cores": threads, "kind": "port",
            "single_thread_value": one,
            "host_hw_threads": os.cpu_count(),
            "sample": "first %d of the %d synthetic genomes self-vs-self (%d pairs) x %d passes = "
                      "%.1f s wall on %d threads (= the CPUs the container may use: affinity capped "
                      "by the cgroup quota; the host has %d hardware threads); oracle/ppk_oracle.c "
                      "gcc -O3 -mavx2 -fopenmp (in-repo restatement of the pp-sketchlib CPU path, "
                      "not the upstream binary)"
                      % (n_s, sk.shape[0], pairs, reps, total, threads, os.cpu_count() or 0)}


def host_call(sk, kmers, tbl, device, reps=5):
    """The call PopPUNK itself makes (pp_sketchlib.queryDatabase after the file read -> ppk_query):
    host sketches in, a FRESH host result array out, PCIe both ways.  The first call uploads and
    re-lays out the sketches; later calls find them resident (ppk_query's database cache)."""
    from poppunk_amd import _lib, pp_sketchlib
    _lib.lib().ppk_release_scratch()                 # start cold: no cached database, no buffers
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out, _ = pp_sketchlib.query_arrays(sk, None, kmers, 16, 14, tbl, devices=(device,))
        times.append((time.perf_counter() - t0) * 1e3)
        del out
    pairs = sk.shape[0] * (sk.shape[0] - 1) // 2
    warm = sorted(times[1:])
    med = warm[len(warm) // 2]
    return {"first_call_ms": round(times[0], 3), "ms": round(med, 3), "min_ms": round(warm[0], 3),
            "pairs_per_s": pairs / (med * 1e-3), "result_bytes": pairs * 8,
            "note": "ppk_query, host buffers in / fresh host array out (np.zeros pages untouched), "
                    "median of %d calls after the first; the first call also uploads + re-lays out "
                    "the %d MB of sketches" % (reps - 1, sk.nbytes >> 20)}


def config5(args, rank, world, local_rank, dev, barrier):
    """BASELINE config 5's shape on N GPUs: fused distance -> boundary -> edge list per band, only
    the edge lists move (engine.edges_sharded)."""
    import torch
    import torch.distributed as dist
    from poppunk_amd import engine, synth
    n = args.config5_genomes
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    tbl = synth.random_match_table(kmers)
    sk_t = synth.make_sketches_device(n, kmers, device="cuda:%d" % local_rank)
    ref = engine.SketchDB(sk_t, 16, 14, device=local_rank)
    del sk_t
    # boundary through the 2 % quantiles of a 2 000-genome subsample's distances
    sub = engine.SketchDB(synth.make_sketches_device(2000, kmers, device="cuda:%d" % local_rank), 16, 14,
                          device=local_rank)
    d_sub, _ = engine.dist(sub, None, kmers, tbl)
    x_max, y_max = synth.boundary_for_quantile(d_sub.cpu().numpy(), 0.02)
    sub.close()
    del d_sub

    def step():
        return engine.edges_sharded(ref, None, kmers, tbl, rank, world, slope=2, x_max=x_max, y_max=y_max,
                                    cap=16 << 20)

    step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.config5_steps):
        edges, counts = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    pairs = n * (n - 1) // 2
    ref.close()
    torch.cuda.empty_cache()
    return {"workload": "%d synthetic genomes self-vs-self, s=1024, k=13,17,21,25,29, fused distance -> "
                        "slope-2 boundary -> edge list; band-split x%d, only the edge lists gathered to rank 0"
                        % (n, world),
            "pairs": pairs, "n_edges": int(sum(counts)), "steps": args.config5_steps,
            "ms_per_step": elapsed / args.config5_steps * 1e3,
            "value": pairs * args.config5_steps / elapsed, "unit": "pairs/s",
            "gathered_bytes_per_step": int(sum(counts[1:])) * 16}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        relaunch_under_torchrun(args)        # does not return
    import torch
    import torch.distributed as dist
    from poppunk_amd import _lib, engine, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("PPK_BENCH_ONE_GPU"):      # debugging aid: every rank on GPU 0
        local_rank = 0
    if world != args.gpus:
        sys.exit("bench.py --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("PPK_BENCH_BACKEND", "nccl")   # "gloo": debugging aid with PPK_BENCH_ONE_GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    lib = _lib.lib()
    n = int(round(args.n * world ** 0.5)) if (args.weak and world > 1) else args.n
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    sk, _ = synth.make_sketches(n, kmers, sketchsize64=16, bbits=14)
    tbl = synth.random_match_table(kmers)
    ref = engine.SketchDB(sk, 16, 14, device=local_rank)

    job = engine.ShardedQuery(ref, None, rank, world, n_chunks=args.chunks if world > 1 else 1)
    rows = job.band_rows
    total_pairs = int(job.total_rows)

    def step():
        job.run(kmers, tbl)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Device spin-up (setup, not part of the W warm-up steps): after an idle period the GPU clock
    # takes ~50 ms of load to ramp and the first launches run 20-30 % slow; a driver that asks for a
    # short --warmup would otherwise time the governor instead of the kernel.
    band_note = "1 GPU"
    if world == 1:
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < args.spinup_ms * 1e-3:
            step()
            torch.cuda.synchronize()
    else:
        if args.spinup_ms > 0:
            n_spin = 30 if os.environ.get("PPK_BENCH_BACKEND", "nccl") == "nccl" else 1
            for _ in range(n_spin):      # every rank must run the SAME number of gathered steps
                step()
            torch.cuda.synchronize()
        # Set-up, like choosing the band edges at all: with equal bands a step lasts as long as the
        # slowest peer -> root transfer (a GPU produces 8 B per pair faster than its one xGMI link to
        # the root carries them) while the root's own band needs none.  Three measured steps re-cut
        # the bands in proportion to each rank's measured rate (engine.ShardedQuery.rebalance).
        if not args.even_bands:
            def probe(n_steps=3):
                """ms per gathered step, max over ranks (the same number on every rank)."""
                barrier()
                t_p = time.perf_counter()
                for _ in range(n_steps):
                    step()
                barrier()
                t = torch.tensor([(time.perf_counter() - t_p) / n_steps * 1e3], dtype=torch.float64,
                                 device=dev if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item())
            ms_even = probe()
            for _ in range(3):
                job.rebalance(kmers, tbl)
            ms_balanced = probe()
            band_note = "rate-balanced (%.2f ms/step against %.2f with equal bands, set-up probe)" % (ms_balanced, ms_even)
            if ms_balanced > 1.02 * ms_even:      # never keep a cut that measures worse than the equal one
                job._layout(engine.shard_bounds(ref.n, 0, world))
                band_note = "equal (the rate-balanced cut measured %.2f ms/step against %.2f)" % (ms_balanced, ms_even)
            rows = job.band_rows
        else:
            band_note = "equal (--even-bands)"
    for _ in range(args.warmup):
        step()
    barrier()
    lib.ppk_prof_enable(1)
    lib.ppk_prof_read(None, None, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    lib.ppk_prof_enable(0)
    import ctypes as C
    kms, kn = C.c_double(0), C.c_longlong(0)
    lib.ppk_prof_read(C.byref(kms), C.byref(kn), 1)
    kernel_ms = kms.value / max(kn.value, 1)
    kname = lib.ppk_last_kernel_name().decode()

    compute_ms = kms.value / args.steps          # kernel time per step on this rank (HIP events)
    if world > 1:
        t = torch.tensor([elapsed, compute_ms], dtype=torch.float64,
                         device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, compute_ms = float(t[0].item()), float(t[1].item())

    c5 = None
    if not args.no_config5:
        ref.close()
        job.out = None
        torch.cuda.empty_cache()
        c5 = config5(args, rank, world, local_rank, dev, barrier)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_pairs * args.steps / elapsed
        per_launch = rows[0] / job.n_chunks          # pairs one launch of the dominant kernel covers
        k_s = kernel_ms * 1e-3
        lane_ops = VALU_OPS_PER_PAIR * per_launch / k_s if k_s > 0 else 0.0
        algo_gbs = ALGO_BYTES_PER_PAIR * per_launch / k_s / 1e9 if k_s > 0 else 0.0
        traffic, traffic_source = None, None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile) and world == 1:
            try:
                tj = json.load(open(tfile))
                traffic = tj.get("n%d" % n)
                traffic_source = tj.get("source", "profiles/pmc_traffic.json")
            except Exception:
                traffic = None
        roof = {"bound": "valu", "achieved": round(lane_ops / 1e12, 3), "peak": round(VALU_PEAK_LANE_OPS / 1e12, 2),
                "unit": "T lane-op/s", "frac": round(lane_ops / VALU_PEAK_LANE_OPS, 4),
                "frac_of_measured_bitop3_stream": round(lane_ops / VALU_MEASURED_LANE_OPS, 4),
                "traffic": traffic, "traffic_source": traffic_source,
                "hbm_frac_counter": round(traffic / k_s / 1e9 / HBM_PEAK_GBS, 4) if (traffic and k_s > 0) else None,
                "hbm_naive_x": round(algo_gbs / HBM_PEAK_GBS, 2),
                "hbm_naive_GBs": round(algo_gbs, 1),
                "kernel": kname, "kernel_ms": round(kernel_ms, 4),
                "pairs_per_launch": per_launch,
                "note": "Integer set-intersection (no MFMA): the binding roof is VALU issue.  achieved = "
                        "2400 VALU lane-ops/pair (5 k x 16 blocks x (28 v_bitop3 + 2 v_bcnt)) x pairs per launch "
                        "/ HIP-event kernel time; peak = 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz; "
                        "frac_of_measured_bitop3_stream is against the 55.6 T lane-op/s a bare 3-VGPR v_bitop3 "
                        "stream sustains on this chip (tools/ubench_valu.hip).  HBM: the LDS + register tile "
                        "re-uses every sketch row ~250x, so SURVEY 8(d)'s algorithmic bytes (17928 B/pair) "
                        "over the kernel time are hbm_naive_x TIMES the 8 TB/s peak (a reuse factor, not a "
                        "fraction); hbm_frac_counter = PMC-measured fabric bytes per launch (traffic, from "
                        "traffic_source -- a recorded rocprofv3 run, not measured in this process) / kernel "
                        "time / 8 TB/s"}
        cpu = None
        if not args.no_cpu and world == 1:
            cpu = cpu_baseline(sk, kmers, tbl, args.cpu_seconds)
        hc = None
        if not args.no_host_call and world == 1:
            hc = host_call(sk, kmers, tbl, local_rank)
        line = {
            "metric": "genome-pair distances/sec (10k self, s=1024, k=13-29)",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak" if (args.weak and world > 1) else "strong", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "%d synthetic genomes self-vs-self, s=1024 (sketchsize64=16, "
                                   "bbits=14), k=13,17,21,25,29, %d pairs, output [n_pairs,2] f32 "
                                   "on rank 0" % (n, total_pairs),
                       "n_genomes": n, "pairs": total_pairs,
                       "parallelism": "band-split x%d (%s bands), %d-chunk pipelined p2p gather to rank 0"
                                      % (world, band_note.split(" ")[0], args.chunks) if world > 1 else "1 GPU"},
            "roofline": roof, "cpu_baseline": cpu, "host_call": hc, "config5": c5,
        }
        if world > 1:
            # where an N-GPU step goes: the slowest rank's kernel time, and what the root receives
            line["multi_gpu"] = {
                "compute_ms_per_step_max_rank": round(compute_ms, 4),
                "gather_exposed_ms_per_step": round(max(ms_per_step - compute_ms, 0.0), 4),
                "gathered_bytes_per_step": int(sum(job.band_rows[1:])) * 8,
                "band_shares": [round(b / max(total_pairs, 1), 4) for b in job.band_rows],
                "bands": band_note + "; the root's band needs no transfer, so equal bands are not the "
                                     "fastest cut (engine.ShardedQuery.rebalance)",
                "note": "value includes the p2p gather of every peer's distance block into the "
                        "PopPUNK-ordered matrix on rank 0 (pipelined under compute in %d chunks); "
                        "the root's inbound xGMI links bound it.  config5 is the shape that scales: "
                        "only edge lists move" % args.chunks}
        if cpu:
            line["speedup_vs_cpu"] = value / cpu["value"]
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
