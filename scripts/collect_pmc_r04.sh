#!/bin/bash
# Round-4 counter collection (run through gpurun): for configs 2 / 4 / 5 in bf16 and config 2 in bf16x3, two PMC passes (FETCH_SIZE, WRITE_SIZE;
# counters only, no trace domains, separate runs) over eager steps of the bench command; for configs 2 and 5 also the two SQ passes
# (matrix-pipe busy, LDS bank conflicts, instruction counts).  Summaries: scripts/pmc_summary_r04.py -> profiles/r04_traffic.json etc.
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_r04
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CD in "2 bf16" "4 bf16" "5 bf16" "2 bf16x3"; do
  set -- $CD; C=$1; D=$2
  ARGS="--config $C --dtype $D --eager --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-fwd-bwd-only --no-pipeline --no-extra-legs"
  for CTR in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $CTR --output-format csv -d $OUT/c${C}_${D}_$CTR -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/c${C}_${D}_$CTR.json 2> $OUT/c${C}_${D}_$CTR.err
    echo "config $C $D $CTR done"
  done
done
for C in 2 5; do
  ARGS="--config $C --eager --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-fwd-bwd-only --no-pipeline --no-extra-legs"
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/c${C}_sq1 -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/c${C}_sq1.json 2> $OUT/c${C}_sq1.err
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/c${C}_sq2 -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/c${C}_sq2.json 2> $OUT/c${C}_sq2.err
  echo "config $C SQ passes done"
done
python3 $REPO/scripts/pmc_summary_r04.py $OUT
