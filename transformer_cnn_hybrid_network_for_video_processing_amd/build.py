"""Build libhybrid_hip.so (gfx950) in-tree with hipcc.  No torch headers are involved:
the library is a plain C-ABI shared object (include/hybrid_hip.h)."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libhybrid_hip.so")
LIB_X3 = os.path.join(PKG, "libhybrid_hip_x3.so")     # the same sources with -DHYB_F32_X3: fp32 storage, split-bf16 products (hyb_common.h)
SOURCES = ["conv_fwd.hip", "conv_v2.hip", "conv_first.hip", "conv_first_wave.hip", "conv_wgrad.hip", "bn_pool.hip", "linear.hip", "attention.hip", "layernorm.hip", "model.hip", "fused.hip", "fct.hip", "fct_bwd.hip", "bn2d.hip", "optim.hip"]


# Per-file compiler flags.  conv_first_wave.hip: MFMA results in VGPRs instead of AGPRs -- its kernels are bound by VALU instruction issue
# and every accumulator value is consumed by vector instructions, which cannot read AGPRs (one v_accvgpr_read per value otherwise).
# conv_wgrad.hip: no SLP vectorisation -- it packs the producers' fp32 arithmetic into v_pk_fma_f32 / v_pk_mul_f32, which issue slower
# than the scalar pairs next to the consumers' MFMAs (measured: fused weight gradient 82.6 -> 78.9 us at stage 4, profiles/r03_wgrad_ablation.txt).
# conv_v2.hip: the same flag, same reason (epilogue arithmetic beside the MFMA stream; -4 us per step same-box A/B); bn_pool.hip, a pure streaming
# file, LOSES 14 us per step with it and keeps the default.
FILE_FLAGS = {"conv_first_wave.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "conv_wgrad.hip": ["-fno-slp-vectorize"], "conv_v2.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(LIB_X3):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(LIB_X3))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "hybrid_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, jobs=8):
    """Compile every HIP source for gfx950 and link the two shared objects next to this file: libhybrid_hip.so and, from the same
    sources with -DHYB_F32_X3, libhybrid_hip_x3.so (compute_dtype="bf16x3")."""
    if not force and not needs_build():
        return LIB
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
             "-Wno-unused-result"] + os.environ.get("HYB_EXTRA_FLAGS", "").split()
    variants = [("build", [], LIB), ("build_x3", ["-DHYB_F32_X3"], LIB_X3)]
    procs, links = [], []
    for objdir_name, extra, out in variants:
        objdir = os.path.join(PKG, objdir_name)
        os.makedirs(objdir, exist_ok=True)
        objs = []
        for src in SOURCES:
            obj = os.path.join(objdir, src.replace(".hip", ".o"))
            objs.append(obj)
            cmd = [_hipcc()] + flags + extra + FILE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            if len(procs) >= jobs:
                _drain(procs, 1)
        links.append((out, objs))
    _drain(procs, 0)
    for out, objs in links:
        # -z defs: a kernel whose host stub was not emitted (a target-checked builtin inside a template, rejected silently in the host pass)
        # must fail here, not at dlopen on the GPU box
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-z,defs", "-o", out] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout.decode())
    return LIB


def _drain(procs, leave):
    while len(procs) > leave:
        src, p = procs.pop(0)
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
