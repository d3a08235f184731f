"""Copy the judged summaries of gpurun_out/prof_r04/ (scripts/collect_r04.sh) into profiles/ as r04_<state>_* and summarise the counter passes
of gpurun_out/pmc_r04/ (scripts/pmc_summary_r04.py -> profiles/r04_traffic.json, r04_pmc_*).   python scripts/publish_r04.py <state>"""
import os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
state = sys.argv[1]
src = os.path.join(root, "gpurun_out", "prof_r04")
dst = os.path.join(root, "profiles")
tag = f"r04_{state}"
def cp(a, b):
    if os.path.exists(os.path.join(src, a)) and os.path.getsize(os.path.join(src, a)) > 0:
        shutil.copy(os.path.join(src, a), os.path.join(dst, b)); print(b)
for name in ("default", "c2", "c2_eager", "c4", "c5", "fp32", "bf16x3", "mixed"):
    cp(f"bench_{name}.json", f"{tag}_bench_{name}.json")
for c in (2, 4, 5):
    cp(f"kt_c{c}/kt_kernel_stats.csv", f"{tag}_kernel_stats_c{c}.csv")
for m in ("bf16", "bf16x3", "mixed"):
    cp(f"step_trace_{m}.txt", f"{tag}_step_trace_{m}.txt")
cp("st_bf16x3/kt_kernel_stats.csv", f"{tag}_kernel_stats_c2_bf16x3.csv")
cp("attention_microbench.json", f"{tag}_attention_microbench.json")
cp("fct_bench.json", f"{tag}_fct_bench.json")
cp("stage1_bench.json", f"{tag}_stage1_bench.json")
cp("enc32k_bench.json", f"{tag}_enc32k_bench.json")
cp("kt_enc32k/kt_kernel_stats.csv", f"{tag}_kernel_stats_enc32k.csv")
cp("kt_fct/kt_kernel_stats.csv", f"{tag}_kernel_stats_fct.csv")
if "--no-pmc" not in sys.argv:
    subprocess.check_call([sys.executable, os.path.join(root, "scripts", "pmc_summary_r04.py"), os.path.join(root, "gpurun_out", "pmc_r04")])
