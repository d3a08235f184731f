#!/bin/bash
# SQ counters of one variant of the fused weight-gradient kernel (scripts/micro/wg/libwg_<name>.so) -> gpurun_out/wg_pmc_<name>/
NAME=${1:-v3p}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/wg_pmc_$NAME
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WGB_FIRST=1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT -o a -- $REPO/scripts/micro/wg/wgrad_bench $REPO/scripts/micro/wg/libwg_$NAME.so - 3 > $OUT/log_a.txt 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM --output-format csv -d $OUT -o b -- $REPO/scripts/micro/wg/wgrad_bench $REPO/scripts/micro/wg/libwg_$NAME.so - 3 > $OUT/log_b.txt 2>&1
python3 - <<PY
import csv, glob, collections
for tag in "ab":
    for f in glob.glob("$OUT/**/%s_counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40] + " grid=" + r.get("Grid_Size", "?")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            if "wgrad_v" not in k: continue
            print(k)
            for c, v in sorted(d.items()): print("   %-28s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
