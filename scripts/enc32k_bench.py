"""Encoder_32K (the ResNet-bottleneck backbone) throughput on the frame-folded clip [B*T, 3, 256, 256]:
forward (eval) and training pass (train-mode forward with Dropout2d + backward) in frames/s on the GPU (HIP events), the CPU oracle
(oracle/encoder32k_ref.py, torch fp32 on the host cores, a bounded sample of frames) beside it; per-entry-point time share from a
second, synchronised pass.  24.0 GFLOP per frame forward (x3 with the backward)."""
import argparse, collections, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transformer_cnn_hybrid_network_for_video_processing_amd as P  # noqa: F401
from transformer_cnn_hybrid_network_for_video_processing_amd import _lib
from transformer_cnn_hybrid_network_for_video_processing_amd import encoder32k as M
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=16); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--cpu", action="store_true"); ap.add_argument("--cpu-frames", type=int, default=2); ap.add_argument("--no-train", action="store_true")
a = ap.parse_args()
torch.manual_seed(0)
m = M.Encoder_32K().cuda().eval()
x = torch.rand(a.frames, 3, 256, 256, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.no_grad():
    m(x); m(x)
    e0.record()
    for _ in range(a.reps): m(x)
    e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / a.reps
GF = 24.0
out = {"workload": f"Encoder_32K forward, frames [{a.frames},3,256,256] -> tokens [{a.frames},8,4096], fp32", "ms": ms,
       "frames_per_s": a.frames / ms * 1e3, "tflops": GF * a.frames / ms}
if not a.no_train:
    m.train()
    def train_pass():
        for p_ in m.parameters(): p_.grad = None
        m(x).square().mean().backward()
    train_pass(); train_pass()
    e0.record()
    for _ in range(a.reps): train_pass()
    e1.record(); e1.synchronize()
    tms = e0.elapsed_time(e1) / a.reps
    out["train_fwd_bwd"] = {"ms": tms, "frames_per_s": a.frames / tms * 1e3, "tflops": 3 * GF * a.frames / tms}
acc = collections.Counter()
orig = _lib._Lib.call
def timed(self, name, *args):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = orig(self, name, *args); torch.cuda.synchronize(); acc[name] += time.perf_counter() - t0; return r
_lib._Lib.call = timed
if a.no_train:
    with torch.no_grad(): m(x)
else:
    train_pass()
_lib._Lib.call = orig
tot = sum(acc.values())
out["share_by_entry_point"] = {k: round(v / tot, 3) for k, v in acc.most_common()}
if a.cpu:
    from oracle import encoder32k_ref as R
    p = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    xc = x[: a.cpu_frames].cpu()
    with torch.no_grad():
        t0 = time.time(); R.forward(p, xc, False); dt = time.time() - t0
    out["cpu_oracle"] = {"ms": dt * 1e3, "frames_per_s": a.cpu_frames / dt, "cores": torch.get_num_threads(), "sample": f"{a.cpu_frames} frames"}
    if not a.no_train:
        for k, v in p.items():
            if v.is_floating_point() and "running" not in k: v.requires_grad_()
        t0 = time.time(); R.forward(p, xc, True).square().mean().backward(); dt = time.time() - t0
        out["cpu_oracle_train_fwd_bwd"] = {"ms": dt * 1e3, "frames_per_s": a.cpu_frames / dt, "cores": torch.get_num_threads(),
                                           "sample": f"{a.cpu_frames} frames"}
print(json.dumps(out))
