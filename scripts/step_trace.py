"""One training step as the GPU ran it, from a rocprofv3 kernel_trace.csv:  python scripts/step_trace.py <kernel_trace.csv> [anchor-kernel-substring]

Takes the launches between the last two launches of the anchor kernel (default: adamw_kernel, the last kernel of a step) and prints them in
start order with duration and the idle gap before each, then the sums: busy time, idle time, launches."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "adamw_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
if len(idx) < 3:
    sys.exit(f"fewer than three launches of {anchor}")
lo, hi = idx[-3], idx[-2]          # the step before the last one (the last may be followed by the end-of-run copies)
step = rows[lo + 1:hi + 1]
t0 = int(rows[lo]["End_Timestamp"])
busy = idle = 0
prev_end = t0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+|void |at::native::", "", r["Kernel_Name"])[:72]
    gap = s - prev_end
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {name}")
    busy += e - s
    idle += max(gap, 0)
    prev_end = max(prev_end, e)
print(f"step: {len(step)} launches, busy {busy / 1e3:.1f} us, idle {idle / 1e3:.1f} us, wall {(prev_end - t0) / 1e3:.1f} us")
