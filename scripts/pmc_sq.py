"""Summarise rocprofv3 SQ counter passes (one directory per pass, same bench command) into profiles/<tag>_pmc_sq_kernels.csv:
per kernel name the average per launch of every counter, plus
  mfma_busy_frac   = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (SQ_BUSY_CYCLES / 32 SEs)      [as in round 1]
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  valu_issue_frac  = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES ... reported raw; interpretation in DESIGN.md"""
import collections, csv, glob, os, re, sys
tag, dirs = sys.argv[1], sys.argv[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d*", "", n)
    return re.sub(r"\(.*$", "", n).replace(",", ";")[:70]
data = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            data[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for k in data for c in data[k]})
rows = []
for k, cs in data.items():
    avg = {c: sum(v) / len(v) for c, v in cs.items()}
    n = max(len(v) for v in cs.values())
    busy = avg.get("SQ_BUSY_CYCLES", 0.0)
    mf = (avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024) / (busy / 32) if busy else 0.0
    lc = avg.get("SQ_LDS_BANK_CONFLICT", 0.0) / avg["SQ_LDS_IDX_ACTIVE"] if avg.get("SQ_LDS_IDX_ACTIVE") else 0.0
    rows.append((busy * n, k, n, mf, lc, avg))
rows.sort(reverse=True)
with open(os.path.join(root, "profiles", f"{tag}_pmc_sq_kernels.csv"), "w") as f:
    f.write("kernel,launches,mfma_busy_frac,lds_conflict_frac," + ",".join(counters) + "\n")
    for _, k, n, mf, lc, avg in rows:
        f.write(f"{k},{n},{mf:.3f},{lc:.3f}," + ",".join(str(round(avg.get(c, 0))) for c in counters) + "\n")
for _, k, n, mf, lc, avg in rows[:24]:
    print(f"{k[:60]:60s} n={n:3d} mfma_busy {mf:.3f} lds_conflict {lc:.3f} valu_active/busy {avg.get('SQ_ACTIVE_INST_VALU',0)/max(avg.get('SQ_BUSY_CYCLES',1),1):.2f} insts_valu {avg.get('SQ_INSTS_VALU',0):.3g}")
