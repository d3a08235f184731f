"""Copy the judged summaries of gpurun_out/prof_<tag>/ (scripts/collect_profiles.sh) into profiles/ under round-tagged names and
write profiles/<round>_traffic.json (scripts/pmc_traffic.py).  python scripts/publish_profiles.py r02 b"""
import os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, state = sys.argv[1], sys.argv[2]
src = os.path.join(root, "gpurun_out", f"prof_{rnd}")
dst = os.path.join(root, "profiles")
tag = f"{rnd}_{state}"
def cp(a, b):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b)); print(b)
for name in ("default", "c2", "c2_eager", "c4", "c5", "fp32", "bf16x3"):
    cp(f"bench_{name}.json", f"{tag}_bench_{name}.json")
for c in (2, 4, 5):
    cp(f"kt_c{c}/kt_kernel_stats.csv", f"{tag}_kernel_stats_c{c}.csv")
for m in ("bf16", "bf16x3"):
    cp(f"step_trace_{m}.txt", f"{tag}_step_trace_{m}.txt")
cp("st_bf16x3/kt_kernel_stats.csv", f"{tag}_kernel_stats_c2_bf16x3.csv")
cp("attention_microbench.json", f"{tag}_attention_microbench.json")
cp("fct_bench.json", f"{tag}_fct_bench.json")
cp("stage1_bench.json", f"{tag}_stage1_bench.json")
cp("enc32k_bench.json", f"{tag}_enc32k_bench.json")
cp("kt_enc32k/kt_kernel_stats.csv", f"{tag}_kernel_stats_enc32k.csv")
subprocess.check_call([sys.executable, os.path.join(root, "scripts", "pmc_traffic.py"), os.path.join(src, "pmc_fetch"), os.path.join(src, "pmc_write"), tag])
