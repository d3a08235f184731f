"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into per-kernel HBM
bytes per launch: profiles/<tag>_pmc_fetch_write_step_kernels.csv and profiles/<round>_traffic.json (read by bench.py; <round> = the tag up to its first '_').
FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md HBM section); both counters are in KiB-sized units of 1 KB."""
import collections, csv, glob, json, os, re, sys

def load(d, counter):
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per

def short(n):
    n = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1", "", n)
    return re.sub(r"\(.*$", "", n).replace(",", ";")

fetch_dir, write_dir, tag = sys.argv[1], sys.argv[2], sys.argv[3]
fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows, traffic = [], {}
for k in sorted(fe, key=lambda k: -sum(fe[k])):
    f_avg = sum(fe[k]) / len(fe[k]); w_avg = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0])))
    fb, wb = f_avg * 1024 * 2, w_avg * 1024
    rows.append((short(k), len(fe[k]), round(f_avg), round(fb), round(w_avg), round(fb + wb)))
    m = re.search(r"wgrad_v2_kernel<(true|false), (\d+)[,>]", k) if ("wgrad" in k and "reduce" not in k) else None
    name = "wgrad_v2_kernel<%s, %s>" % (m.group(1), m.group(2)) if m else ("wgrad_v3_kernel" if "wgrad_v3_kernel" in k else short(k))
    if name not in traffic:
        traffic[name] = {"hbm_bytes_per_launch": fb + wb, "fetch_size_kb_raw_avg": f_avg, "write_size_kb_avg": w_avg, "launches_sampled": len(fe[k]),
                         "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --eager --steps 6 "
                                   "--warmup 2 --no-cpu-baseline --no-roofline --no-fwd-bwd-only --no-pipeline` (real steps); FETCH_SIZE "
                                   "doubled per the gfx950 correction in MI355X_MICROARCH.md; average over the launches of this kernel name"}
with open(os.path.join(root, "profiles", f"{tag}_pmc_fetch_write_step_kernels.csv"), "w") as f:
    f.write("kernel,launches,FETCH_SIZE_KB_avg_raw,FETCH_bytes_corrected_x2,WRITE_SIZE_KB_avg,hbm_bytes_per_launch\n")
    for r in rows:
        f.write(",".join(str(x) for x in r) + "\n")
json.dump(traffic, open(os.path.join(root, "profiles", tag.split("_")[0] + "_traffic.json"), "w"), indent=1)
print("step total GB:", sum(r[5] * r[1] for r in rows) / 8 / 1e9, "(8 profiled steps)")
for r in rows[:12]:
    print(r)
