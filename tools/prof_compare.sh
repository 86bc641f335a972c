#!/bin/bash
# ubench_pipe vs the product kernel (no epilogue / no DMA) under rocprofv3: durations + SQ counters
set -u
OUT=gpurun_out/cmp
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_u -o u -- ./tools/ubench_pipe.out > $OUT/u.log 2>&1
PPK_ABLATE=5 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_r5 -o r -- python tools/ab_square.py > $OUT/r5.log 2>&1
PPK_ABLATE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_r1 -o r -- python tools/ab_square.py > $OUT/r1.log 2>&1
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_u -o u -- ./tools/ubench_pipe.out > /dev/null 2>&1
PPK_ABLATE=5 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_r5 -o r -- python tools/ab_square.py > /dev/null 2>&1
PPK_ABLATE=1 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_r1 -o r -- python tools/ab_square.py > /dev/null 2>&1
C2="SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH"
rocprofv3 --pmc $C2 --output-format csv -d $OUT/pmc2_u -o u -- ./tools/ubench_pipe.out > /dev/null 2>&1
PPK_ABLATE=5 rocprofv3 --pmc $C2 --output-format csv -d $OUT/pmc2_r5 -o r -- python tools/ab_square.py > /dev/null 2>&1
find $OUT -name "*.csv" | head -30
