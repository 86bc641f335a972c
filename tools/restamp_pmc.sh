#!/bin/bash
# After a change to the hashed sources: the four PMC passes of tools/collect_profiles.sh alone, then the default bench
# line (which then carries this build's traffic).  ROUND=r06 tools/restamp_pmc.sh; ROUND=r06 python tools/update_profiles.py
set -u
ROUND=${ROUND:-r06}
OUT=gpurun_out/$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
S="python bench.py --steps 5 --warmup 2 --no-cpu --no-config5 --no-host-call --no-file-call --no-other-configs"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- $S > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -o w -- $S > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -o s -- $S > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_lds -o l -- $S > /dev/null 2>&1
B="python bench.py --steps 100 --warmup 20 --no-cpu --no-config5 --no-host-call --no-file-call"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $B > $OUT/bench_under_rocprof.log 2>&1
python tools/kernel_rows.py $(ls $OUT/kt/*/kt_kernel_trace.csv $OUT/kt/kt_kernel_trace.csv 2>/dev/null | head -1) > $OUT/bench_kernel_rows.csv 2>> $OUT/bench_under_rocprof.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 200 $OUT/bench.json
