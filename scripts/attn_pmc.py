"""Summarise a rocprofv3 --pmc pass over scripts/attn_microbench.py: per kernel, counters averaged per launch."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "attention" in r["Kernel_Name"]:
        key = ("bwd" if "bwd" in r["Kernel_Name"] else "fwd") + " grid=" + r.get("Grid_Size", "?") + " wg=" + r.get("Workgroup_Size", "?")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
