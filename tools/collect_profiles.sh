#!/bin/bash
# Regenerates the measured artefacts of a round under gpurun_out/$ROUND (run through gpurun from the repo root):
#   ROUND=r05 tools/collect_profiles.sh     then `ROUND=r05 python tools/update_profiles.py` copies into profiles/
# FULL=1 adds the micro-benchmarks and host A/Bs of earlier rounds.
set -u
ROUND=${ROUND:-r05}
OUT=gpurun_out/$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
# the whole default line (what the driver runs)
python bench.py > $OUT/bench.json 2> $OUT/bench.err
# the same command under the kernel trace: one summary row per kernel name + per-dispatch rows (split by grid below)
B="python bench.py --steps 100 --warmup 20 --no-cpu --no-config5 --no-host-call --no-file-call"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $B > $OUT/bench_under_rocprof.log 2>&1
python tools/kernel_rows.py $(ls $OUT/kt/*/kt_kernel_trace.csv $OUT/kt/kt_kernel_trace.csv 2>/dev/null | head -1) > $OUT/bench_kernel_rows.csv 2>> $OUT/bench_under_rocprof.log
# PMC passes (their own runs, no tracing of anything else), headline kernel only
S="python bench.py --steps 5 --warmup 2 --no-cpu --no-config5 --no-host-call --no-file-call --no-other-configs"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- $S > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -o w -- $S > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -o s -- $S > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_lds -o l -- $S > /dev/null 2>&1
python tools/ab_smalljob.py > $OUT/smalljob.txt 2>&1
python tools/ab_smalljob.py ksplit_fused=0 > $OUT/smalljob_two_pass.txt 2>&1
python tools/ab_pinning.py 2>&1 | grep -v amdgpu > $OUT/ab_pinning.txt
python tools/latency_table.py 2>&1 | grep -v amdgpu > $OUT/latency_table.txt
python tools/time_wide.py 2>&1 | grep -v amdgpu > $OUT/time_wide.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k2 -o k2 -- python tools/k2_trace.py 200 > $OUT/k2_trace.txt 2>&1
PPK_BENCH_ONE_GPU=1 PPK_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --config5-genomes 20000 --no-cpu > $OUT/two_ranks_one_gpu.json 2> $OUT/two_ranks_one_gpu.err
if [ "${FULL:-0}" = "1" ]; then
  timeout 1500 python tools/measure_configs.py $OUT/configs.json > $OUT/configs.log 2>&1
  ./tools/watch_clocks.sh > $OUT/power_clocks.txt 2>&1
  python tools/ab_host.py > $OUT/ab_host.txt 2>&1
  python tools/ab_knn.py > $OUT/knn_from_tiles.txt 2>&1
fi
tail -1 $OUT/bench.json | cut -c1-300
