"""Regenerate DESIGN.md from scripts/DESIGN.md.tmpl: the @@KEY@@ fields are read from the committed bench lines under profiles/."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02_b"
def L(name):
    return json.loads(open(os.path.join(root, "profiles", f"{tag}_bench_{name}.json")).read().strip().splitlines()[-1])
c2, c4, c5, f32, eg = L("c2"), L("c4"), L("c5"), L("fp32"), L("c2_eager")
rep = {"@@C2MS@@": f"{c2['ms_per_step']:.2f}", "@@C2@@": f"{c2['value']:.0f}", "@@C2FB@@": f"{c2['fwd_bwd_only']['value']:.0f}",
       "@@C4@@": f"{c4['value']:.0f}", "@@C4MS@@": f"{c4['ms_per_step']:.2f}", "@@C5@@": f"{c5['value']:.0f}", "@@C5MS@@": f"{c5['ms_per_step']:.2f}",
       "@@F32@@": f"{f32['value']:.0f}", "@@EAGERMS@@": f"{eg['ms_per_step']:.2f}", "@@PCIE@@": f"{c2['pcie_inclusive']['value']:.0f}"}
s = open(os.path.join(root, "scripts", "DESIGN.md.tmpl")).read()
for k, v in rep.items():
    s = s.replace(k, v)
open(os.path.join(root, "DESIGN.md"), "w").write(s)
print(rep)
