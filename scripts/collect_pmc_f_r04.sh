#!/bin/bash
# Round-4 counter passes for the (f) rows (run through gpurun): FETCH_SIZE and WRITE_SIZE (counters only, separate runs) over one training pass of
# scripts/enc32k_bench.py and scripts/fct_bench.py.  Summary: scripts/pmc_summary_f_r04.py -> profiles/r04_pmc_fetch_write_{enc32k,fct}.csv, r04_traffic_f.json
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_f_r04
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CTR in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CTR --output-format csv -d $OUT/enc32k_$CTR -o pmc -- python3 $REPO/scripts/enc32k_bench.py --frames 16 --reps 1 > $OUT/enc32k_$CTR.json 2> $OUT/enc32k_$CTR.err
  echo "enc32k $CTR done"
  rocprofv3 --pmc $CTR --output-format csv -d $OUT/fct_$CTR -o pmc -- python3 $REPO/scripts/fct_bench.py --reps 1 > $OUT/fct_$CTR.json 2> $OUT/fct_$CTR.err
  echo "fct $CTR done"
done
python3 $REPO/scripts/pmc_summary_f_r04.py $OUT
