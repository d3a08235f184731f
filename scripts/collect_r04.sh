#!/bin/bash
# Round-4 collection on the GPU box (run through gpurun): scripts/collect_profiles.sh's bench lines, kernel-trace summaries and step traces
# (without its config-2-only PMC passes), then scripts/collect_pmc_r04.sh (FETCH / WRITE for configs 2 / 4 / 5 and bf16x3, SQ for 2 and 5).
# Output: gpurun_out/prof_r04/ and gpurun_out/pmc_r04/; scripts/publish_r04.py <state> copies the judged summaries into profiles/.
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r04
rm -rf $OUT; mkdir -p $OUT
cd $REPO
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err          # the driver's command: headline + every extra leg
echo "default bench done"
python3 bench.py --steps 100 --warmup 10 --no-extra-legs > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python3 bench.py --eager --steps 100 --warmup 10 --no-cpu-baseline --no-roofline > $OUT/bench_c2_eager.json 2> $OUT/bench_c2_eager.err
python3 bench.py --config 4 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python3 bench.py --config 5 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
python3 bench.py --dtype fp32 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_fp32.json 2> $OUT/bench_fp32.err
python3 bench.py --dtype bf16x3 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_bf16x3.json 2> $OUT/bench_bf16x3.err
python3 bench.py --dtype mixed --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_mixed.json 2> $OUT/bench_mixed.err
echo "bench lines done"
python3 scripts/attn_microbench.py > $OUT/attention_microbench.json 2> $OUT/attention_microbench.err || true
python3 scripts/fct_bench.py --cpu > $OUT/fct_bench.json 2> $OUT/fct_bench.err || true
python3 scripts/s1_bench.py > $OUT/stage1_bench.json 2> /dev/null || true
python3 scripts/enc32k_bench.py --frames 16 --cpu > $OUT/enc32k_bench.json 2> $OUT/enc32k_bench.err || true
echo "micro benches done"
cd /tmp && export TMPDIR=/tmp
for C in 2 4 5; do
  STEPS=30; [ $C != 2 ] && STEPS=12
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_c$C -o kt -- python3 $REPO/bench.py --config $C --steps $STEPS --warmup 5 --no-cpu-baseline --no-pipeline --no-fwd-bwd-only --no-roofline > $OUT/bench_c${C}_under_rocprof.json 2> $OUT/kt_c$C.err
  rm -f $OUT/kt_c$C/kt_kernel_trace.csv
  echo "kernel trace config $C done"
done
for M in bf16 bf16x3 mixed; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$M -o kt -- python3 $REPO/bench.py --dtype $M --steps 12 --warmup 3 --no-cpu-baseline --no-pipeline --no-fwd-bwd-only --no-roofline --no-extra-legs > $OUT/st_$M.json 2> $OUT/st_$M.err
  python3 $REPO/scripts/step_trace.py $OUT/st_$M/kt_kernel_trace.csv > $OUT/step_trace_$M.txt
  rm -f $OUT/st_$M/kt_kernel_trace.csv
  echo "step trace $M done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_enc32k -o kt -- python3 $REPO/scripts/enc32k_bench.py --frames 16 --reps 5 > $OUT/enc32k_under_rocprof.json 2> $OUT/kt_enc32k.err || true
rm -f $OUT/kt_enc32k/kt_kernel_trace.csv
echo "kernel trace Encoder_32K done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_fct -o kt -- python3 $REPO/scripts/fct_bench.py --reps 5 > $OUT/fct_under_rocprof.json 2> $OUT/kt_fct.err || true
rm -f $OUT/kt_fct/kt_kernel_trace.csv
echo "kernel trace FCT done"
[ "$SKIP_PMC" = "1" ] && { ls $OUT; exit 0; }
bash $REPO/scripts/collect_pmc_r04.sh
ls $OUT
