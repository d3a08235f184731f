#!/bin/bash
# bf16 forward error per kernel generation (one process each: the switches are read once) -> gpurun_out/r3_bf16_drift.jsonl
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3_bf16_drift.jsonl; : > $O
python scripts/diag_bf16_drift.py >> $O
HYB_S1_GRAM=0 python scripts/diag_bf16_drift.py >> $O
HYB_S1_WAVE=0 HYB_S1_WAVE_BWD=0 HYB_S1_GRAM=0 python scripts/diag_bf16_drift.py >> $O
HYB_CONV_V2=0 HYB_WGRAD_V2=0 python scripts/diag_bf16_drift.py >> $O
HYB_S1_WAVE=0 HYB_S1_WAVE_BWD=0 HYB_S1_GRAM=0 HYB_CONV_V2=0 HYB_WGRAD_V2=0 python scripts/diag_bf16_drift.py >> $O
python - <<PY
import json
for l in open("$O"):
    d = json.loads(l)
    print(d["env"])
    for c in d["cases"]: print("   ", c["shape"], "bf16 mean %.2e max %.2e  per seed" % (c["bf16_mean"], c["bf16_max"]), ["%.1e" % e for e in c["bf16_logits_rel_err"]], "fp32 max %.1e" % c["fp32_max"])
PY
