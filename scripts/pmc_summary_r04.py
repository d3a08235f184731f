"""Summarise scripts/collect_pmc_r04.sh's passes (gpurun_out/pmc_r04/) into profiles/:
  r04_traffic.json                      {"config<N>_<dtype>": {kernel: {hbm_bytes_per_launch, fetch/write parts, launches}}}: what bench.py reports as
                                        roofline.traffic for the headline and for the config-4 / config-5 / bf16x3 legs (committed_traffic)
  r04_pmc_fetch_write_c<N>_<dtype>.csv  per-kernel HBM bytes per launch and, from the same directory's kernel names only, nothing else
  r04_pmc_sq_c<N>.csv                   per-kernel SQ counters (matrix-pipe busy fraction, LDS bank-conflict fraction, instruction counts)
FETCH_SIZE is doubled (the gfx950 correction of MI355X_MICROARCH.md's HBM section); both counters count KiB."""
import collections, csv, glob, json, os, re, sys

out = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d*", "", n)
    return re.sub(r"\(.*$", "", n).replace(",", ";")


def bench_name(k):
    """the name bench.py's kernel table uses for this rocprof kernel name"""
    if "reduce" in k:
        return short(k)
    m = re.search(r"wgrad_v2_kernel<(true|false), (\d+)[,>]", k)
    if m:
        return "wgrad_v2_kernel<%s, %s>" % (m.group(1), m.group(2))
    if "wgrad_v3_kernel" in k:
        return "wgrad_v3_kernel"
    m = re.search(r"conv3x3_wgrad_kernel<float, (\d+), (true|false)>", k)
    if m:
        return "conv3x3_wgrad_kernel<float, %s, %s>" % (m.group(1), m.group(2))
    if "conv3x3_nhwc_kernel<float" in k:
        return "conv3x3_nhwc_kernel<float>"
    if "conv3x3_v2_kernel" in k:
        return "conv3x3_v2_kernel"
    if "conv3x3_x3_kernel" in k:
        return "conv3x3_x3_kernel"
    return short(k)


def load(d, counter):
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


traffic = {}
for d in sorted(glob.glob(os.path.join(out, "c*_FETCH_SIZE"))):
    tag = os.path.basename(d)[:-len("_FETCH_SIZE")]                  # c2_bf16
    cfg, dtype = tag[1:].split("_", 1)
    fe, wr = load(d, "FETCH_SIZE"), load(os.path.join(out, tag + "_WRITE_SIZE"), "WRITE_SIZE")
    if not fe:
        continue
    rows, per = [], {}
    for k in sorted(fe, key=lambda k: -sum(fe[k])):
        f_avg = sum(fe[k]) / len(fe[k])
        w_avg = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0])))
        fb, wb = f_avg * 1024 * 2, w_avg * 1024
        rows.append((short(k)[:90], len(fe[k]), round(f_avg), round(fb), round(w_avg), round(fb + wb)))
        name = bench_name(k)
        e = per.setdefault(name, {"hbm_bytes_per_launch": 0.0, "launches_sampled": 0, "_sum": 0.0})
        e["_sum"] += (fb + wb) * len(fe[k]); e["launches_sampled"] += len(fe[k])
    for name, e in per.items():
        e["hbm_bytes_per_launch"] = e.pop("_sum") / e["launches_sampled"]
        e["method"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over `bench.py --config %s --dtype %s --eager --steps 3 --warmup 2` "
                       "(real steps); FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, KiB units; average over every launch of this kernel name" % (cfg, dtype))
    traffic[f"config{cfg}_{dtype}"] = per
    with open(os.path.join(root, "profiles", f"r04_pmc_fetch_write_{tag}.csv"), "w") as f:
        f.write("kernel,launches,FETCH_SIZE_KB_avg_raw,FETCH_bytes_corrected_x2,WRITE_SIZE_KB_avg,hbm_bytes_per_launch\n")
        for r in rows:
            f.write(",".join(str(x) for x in r) + "\n")
    steps = 5.0
    print(f"{tag}: step total {sum(r[5] * r[1] for r in rows) / steps / 1e9:.2f} GB ({steps:.0f} profiled steps); top:", rows[0][0][:40], rows[0][5])
json.dump(traffic, open(os.path.join(root, "profiles", "r04_traffic.json"), "w"), indent=1)

for d1 in sorted(glob.glob(os.path.join(out, "c*_sq1"))):
    tag = os.path.basename(d1)[:-4]
    data = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in (d1, os.path.join(out, tag + "_sq2")):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                data[short(r["Kernel_Name"])[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
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
    with open(os.path.join(root, "profiles", f"r04_pmc_sq_{tag}.csv"), "w") as f:
        f.write("kernel,launches,mfma_busy_frac,lds_conflict_frac," + ",".join(counters) + "\n")
        for _, k, n, mf, lc, avg in rows:
            f.write(f"{k},{n},{mf:.3f},{lc:.3f}," + ",".join(str(round(avg.get(c, 0))) for c in counters) + "\n")
    for _, k, n, mf, lc, avg in rows[:10]:
        print(f"{tag} {k[:56]:56s} n={n:3d} mfma_busy {mf:.3f} lds_conflict {lc:.3f}")
