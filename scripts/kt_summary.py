"""Per-step summary of a rocprofv3 kernel_stats.csv: python scripts/kt_summary.py <csv> <steps>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total GPU ms {tot / 1e6:.2f}  per step {tot / steps / 1e3:.1f} us")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    n = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+|void |at::native::", "", r["Name"])[:64]
    c = int(r["Calls"])
    print(f"{n:64s} {c / steps:6.2f}/step  avg {float(r['AverageNs']) / 1e3:7.1f} us  per-step {float(r['TotalDurationNs']) / steps / 1e3:7.1f} us  {float(r['Percentage']):5.2f}%")
