#!/bin/bash
# two SQ counter passes over four eager steps of the bench command (counters only: no trace domains)
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--eager --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-fwd-bwd-only --no-pipeline"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/p1.json 2> $OUT/p1.err
echo pass1 done
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p2 -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/p2.json 2> $OUT/p2.err
echo pass2 done
ls -la $OUT/p1 $OUT/p2
