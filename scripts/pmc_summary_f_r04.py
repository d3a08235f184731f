"""Summarise scripts/collect_pmc_f_r04.sh's passes (gpurun_out/pmc_f_r04/) into profiles/r04_pmc_fetch_write_{enc32k,fct}.csv and
profiles/r04_traffic_f.json {"enc32k" | "fct": {kernel: {launches, hbm_bytes_per_launch_avg, hbm_bytes_per_launch_max}}}: HBM bytes per launch
of every kernel of one training pass -- FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, KiB units, two separate --pmc passes; launches are matched
by their order within a kernel name.  bench.py reports the MAX over a kernel name's launches as `kernel_roofline.traffic` of the fct / enc32k legs
(the hooked launch is the heaviest of its name)."""
import collections, csv, glob, json, os, re, sys
out = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d*", "", n)
    return re.sub(r"\(.*$", "", n)


def load(d, counter):
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = sorted((r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter), key=lambda r: int(r.get("Dispatch_Id", 0)))
        for r in rows:
            per[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return per


res = {}
for tag in ("enc32k", "fct"):
    fe, wr = load(os.path.join(out, tag + "_FETCH_SIZE"), "FETCH_SIZE"), load(os.path.join(out, tag + "_WRITE_SIZE"), "WRITE_SIZE")
    if not fe:
        continue
    per, rows = {}, []
    for k in sorted(fe, key=lambda k: -sum(fe[k])):
        w = wr.get(k, [])
        n = min(len(fe[k]), len(w)) if w else len(fe[k])
        tot = [fe[k][i] * 2048.0 + (w[i] * 1024.0 if w else 0.0) for i in range(n)]
        per[k] = {"launches": n, "hbm_bytes_per_launch_avg": sum(tot) / n, "hbm_bytes_per_launch_max": max(tot)}
        rows.append((k.replace(",", ";")[:100], n, round(sum(tot) / n), round(max(tot)), round(sum(tot))))
    res[tag] = per
    with open(os.path.join(root, "profiles", f"r04_pmc_fetch_write_{tag}.csv"), "w") as f:
        f.write("kernel,launches,hbm_bytes_per_launch_avg,hbm_bytes_per_launch_max,hbm_bytes_total\n")
        for r in rows:
            f.write(",".join(str(x) for x in r) + "\n")
    print(f"{tag}: {sum(r[4] for r in rows) / 1e9:.2f} GB over the profiled process (warm-up passes + timed passes); top: {rows[0][0][:50]} {rows[0][2]} avg / {rows[0][3]} max")
json.dump(res, open(os.path.join(root, "profiles", "r04_traffic_f.json"), "w"), indent=1)
