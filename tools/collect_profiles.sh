#!/bin/bash
# Regenerates every measured artefact under gpurun_out/$ROUND (run through gpurun from the repo root):
#   ROUND=r03 tools/collect_profiles.sh     then `ROUND=r03 python tools/update_profiles.py` copies into profiles/
set -u
ROUND=${ROUND:-r03}
OUT=gpurun_out/$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --steps 100 --warmup 20 --no-cpu --no-config5 --no-host-call"
S="python bench.py --steps 5 --warmup 2 --no-cpu --no-config5 --no-host-call"
timeout 1500 python tools/measure_configs.py $OUT/configs.json > $OUT/configs.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $B > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- $S > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -o w -- $S > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -o s -- $S > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_lds -o l -- $S > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktc -o c -- python tools/measure_configs.py > /dev/null 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err
./tools/ubench_valu.out > $OUT/ubench_valu.txt 2>&1
./tools/ubench_bank.out > $OUT/ubench_bank.txt 2>&1
./tools/ubench_lds_tile.out > $OUT/ubench_lds_tile.txt 2>&1
./tools/ubench_pipe.out > $OUT/ubench_pipe.txt 2>&1
./tools/watch_clocks.sh > $OUT/power_clocks.txt 2>&1
./tools/ubench_host_out.out > $OUT/ubench_host_out.txt 2>&1
python tools/ab_host.py > $OUT/ab_host.txt 2>&1
python tools/ab_host_parts.py > $OUT/ab_host_parts.txt 2>&1
./tools/ubench_hwid.out > $OUT/ubench_hwid.txt 2>&1
python tools/ab_knn.py > $OUT/knn_from_tiles.txt 2>&1
PPK_BENCH_ONE_GPU=1 PPK_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --config5-genomes 20000 --no-cpu > $OUT/two_ranks_one_gpu.json 2> $OUT/two_ranks_one_gpu.err
tail -1 $OUT/bench.json | cut -c1-300
