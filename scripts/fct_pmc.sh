#!/bin/bash
# SQ counters of the FCT forward (counters only, no trace domains) -> gpurun_out/pmc_fct/
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_fct
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1 -o pmc -- python3 $REPO/scripts/fct_bench.py --no-train --reps 1 --frames 16 > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY --output-format csv -d $OUT/p2 -o pmc -- python3 $REPO/scripts/fct_bench.py --no-train --reps 1 --frames 16 > $OUT/p2.json 2> $OUT/p2.err
ls $OUT/p1 $OUT/p2
