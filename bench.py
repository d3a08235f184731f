#!/usr/bin/env python3
"""Benchmark of the hot path: clips/sec of one full harness step.  Default = BASELINE config 2
([B=8 per GPU, T=16, 3, 224, 224], 4-stage CNN 32-64-128-256 + 2-layer transformer d=512 h=8, bf16).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,4,5}] [--dtype {bf16,fp32}]

`--gpus N` with N > 1 and no launcher environment starts the N ranks ITSELF (a child `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` of this same file, before this process touches the GPU), forwards
rank 0's JSON line and exits with the child's status.  Launched by torch.distributed.run directly (RANK/WORLD_SIZE set) it is
one rank of that job.

A step = zero_grad -> forward -> cross-entropy -> backward -> (RCCL gradient all-reduce when N>1) -> AdamW.step, i.e. the
reference harness inner loop (Model.py:55-59 / FCT.py:328-338) with nothing skipped; synthetic clips are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line (see the task contract) including

  "fwd_bwd_only": the same K steps without zero_grad/AdamW (the metric's literal "fwd+bwd"), timed right after the headline
  "roofline":     the dominant HIP kernel, timed live with events on the launch stream, against the MI355X peak
  "cpu_baseline": the CPU oracle (oracle/hybrid_ref.py, kind "port") timed on the host cores on a bounded sample

Configs (BASELINE.json `configs`; config 2 is the one the metric is quoted on, the others are labelled in config.workload):
  2  [8,16,3,224,224]  d=512 h=8 hid=2048      4  [8,64,3,224,224]  d=768 h=8 hid=3072      5  [4,16,3,448,448]  d=512 h=8 hid=2048
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16 (MI355X_MICROARCH.md, chip-level parameters)
MFMA_F32_PEAK_TFLOPS = 157.3
# bf16x3: every product is three bf16 MFMAs (hi*hi + hi*lo + lo*hi), so the mode's own ceiling in useful FLOPs is a third of the bf16 peak
# mixed: bf16 conv stages (where the dominant kernel lives) + bf16x3 temporal part
MODE_PEAK = {"bf16": MFMA_BF16_PEAK_TFLOPS, "mixed": MFMA_BF16_PEAK_TFLOPS, "bf16x3": MFMA_BF16_PEAK_TFLOPS / 3.0, "fp32": MFMA_F32_PEAK_TFLOPS}
HBM_PEAK_GBS = 8000.0               # HBM3E spec peak (6.29 TB/s measured copy)

CFG = dict(cnn_channels=(32, 64, 128, 256), num_layers=2, num_classes=8)
CONFIGS = {
    2: dict(batch=8, frames=16, size=224, d_model=512, num_heads=8, hidden_dim=2048),
    4: dict(batch=8, frames=64, size=224, d_model=768, num_heads=8, hidden_dim=3072),
    5: dict(batch=4, frames=16, size=448, d_model=512, num_heads=8, hidden_dim=2048),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "bf16x3", "mixed"])
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json config number (defaults of the flags below)")
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--d-model", type=int, default=None)
    ap.add_argument("--heads", type=int, default=None)
    ap.add_argument("--hidden", type=int, default=None)
    ap.add_argument("--optimizer", default="hybrid", choices=["hybrid", "torch"],
                    help="AdamW implementation of the step: this repo's one-launch kernel (SURVEY 8f-2) or torch.optim.AdamW(fused=True)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fwd-bwd-only", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the PCIe-inclusive leg (host uint8 clips fed through clips.ClipPipeline)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not re-measure roofline.traffic with rocprofv3 --pmc child runs (use profiles/*_traffic.json)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the short extra legs printed beside the headline (config 4 / 5, fp32 and bf16x3 "
                                                                 "modes, FCT, Encoder_32K, config-1 CPU baseline, the in-run bf16 logits check)")
    ap.add_argument("--eager", action="store_true", help="issue every launch from Python each step instead of replaying the captured hipGraphs "
                                                         "(graph.GraphedTrainStep); same kernels and arithmetic, more host time")
    args = ap.parse_args(argv)
    c = CONFIGS[args.config]
    for flag, key in (("batch", "batch"), ("frames", "frames"), ("size", "size"), ("d_model", "d_model"), ("heads", "num_heads"), ("hidden", "hidden_dim")):
        if getattr(args, flag) is None:
            setattr(args, flag, c[key])
    return args


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as a CHILD job (never exec: this process may be watched by a
    profiler that already initialised the GPU), stream its output through, and return its exit status."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    n_json = 0
    for line in child.stdout:
        if line.startswith("{"):
            n_json += 1
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = child.wait()
    if rc == 0 and n_json != 1:
        sys.stderr.write(f"bench.py: expected exactly one JSON line from rank 0, saw {n_json}\n")
        rc = 1
    return rc


torch = None        # imported by _import_torch(): the self-launching parent (`--gpus N`, no launcher) must stay clear of the GPU runtime


def _import_torch():
    global torch
    if torch is None:
        import torch as _torch
        torch = _torch
    return torch


def conv_kernel_table(args, dt_code, tdt, dev):
    """Time the contraction kernels of one step standalone through the C ABI (HIP events on the launch stream = torch's
    current stream).  Rows with calls/step > 0 and composite=False are single kernels whose name appears in rocprof."""
    _import_torch()
    from transformer_cnn_hybrid_network_for_video_processing_amd._lib import lib
    st = torch.cuda.current_stream().cuda_stream
    es = 2 if tdt == torch.bfloat16 else 4
    N = args.batch * args.frames
    chans = (3,) + CFG["cnn_channels"]
    rows = []

    def timeit(fn, reps=10):
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    H = args.size
    for li in range(4):
        ci, co = chans[li], chans[li + 1]
        flops = 2.0 * 9 * ci * co * H * H * N
        if li == 0:
            # stage 1 is a fused recompute path: time the whole stage (composite rows, several kernels each)
            x = torch.rand(N, ci, H, H, device=dev)
            w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
            gamma, beta = torch.ones(co, device=dev), torch.zeros(co, device=dev)
            rm, rv = torch.zeros(co, device=dev), torch.ones(co, device=dev)
            nbt = torch.zeros((), dtype=torch.int64, device=dev)
            pooled = torch.empty(N, H // 2, H // 2, co, dtype=tdt, device=dev)
            ss, mi = torch.empty(2, co, device=dev), torch.empty(2, co, device=dev)
            wsf = torch.empty(lib.query("hyb_convstage_fwd_workspace", dt_code, 1, 0, co), dtype=torch.uint8, device=dev)
            wsb = torch.empty(lib.query("hyb_convstage_bwd_workspace", dt_code, 1, N, H, H, 0, co), dtype=torch.uint8, device=dev)
            dp = (torch.randn(N, H // 2, H // 2, co, device=dev) * 0.1).to(tdt)
            dw, dg, db = torch.empty_like(w), torch.empty(co, device=dev), torch.empty(co, device=dev)

            def s1f():
                lib.call("hyb_convstage_fwd", dt_code, 1, x.data_ptr(), w.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                         nbt.data_ptr(), 1, 0.1, 1e-5, N, H, H, ci, 0, co, co, None, pooled.data_ptr(), ss.data_ptr(), mi.data_ptr(),
                         None, None, wsf.data_ptr(), wsf.numel(), st)

            def s1b():
                lib.call("hyb_convstage_bwd", dt_code, 1, dp.data_ptr(), x.data_ptr(), None, None, w.data_ptr(), gamma.data_ptr(), ss.data_ptr(), mi.data_ptr(),
                         1, N, H, H, ci, 0, co, co, None, dw.data_ptr(), dg.data_ptr(), db.data_ptr(), None, wsb.data_ptr(), wsb.numel(), st)
            rows.append(dict(kernel="stage1_fwd (conv+stats, conv+bn+relu+pool)", composite=True, flops=2 * flops,
                             bytes=float(2 * N * ci * H * H * 4 + N * (H // 2) ** 2 * co * es), ms=timeit(s1f)))
            rows.append(dict(kernel="stage1_bwd (recompute+reduce, recompute+wgrad)", composite=True, flops=3 * flops,
                             bytes=float(2 * N * ci * H * H * 4 + 2 * N * (H // 2) ** 2 * co * es), ms=timeit(s1b)))
            del x, pooled, dp, wsf, wsb
            H //= 2
            continue
        cip, cop = ci, co
        x = torch.rand(N, H, H, ci, device=dev).to(tdt)
        w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        wp = torch.empty(lib.query("hyb_conv_packed_elems", 0, cip, cop), dtype=tdt, device=dev)
        lib.call("hyb_conv_pack_weight", dt_code, 0, w.data_ptr(), wp.data_ptr(), co, ci, cop, cip, st)
        y = torch.empty(N, H, H, cop, dtype=tdt, device=dev)
        stats = torch.zeros(2, cop, device=dev)
        part = torch.empty(lib.query("hyb_conv_stats_workspace", cop), dtype=torch.uint8, device=dev)
        dy = (torch.randn(N, H, H, cop, device=dev) * 0.1).to(tdt)
        nb = lib.query("hyb_conv3x3_wgrad_workspace", 0, N, H, H, cip, cop)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        wpd = torch.empty(cip * 9 * cop, dtype=tdt, device=dev)
        lib.call("hyb_conv_pack_weight", dt_code, 1, w.data_ptr(), wpd.data_ptr(), co, ci, cop, cip, st)
        dx = torch.empty(N, H, H, cip, dtype=tdt, device=dev)
        io_bytes = float(N * H * H * (ci + co) * es)
        rows.append(dict(kernel=f"conv{li + 1}_fwd", composite=False, flops=flops, bytes=io_bytes, name="conv3x3_v2_kernel",
                         ms=timeit(lambda: lib.call("hyb_conv3x3_fwd", dt_code, 0, x.data_ptr(), wp.data_ptr(), y.data_ptr(), stats.data_ptr(),
                                                    part.data_ptr(), N, H, H, ci, cip, cop, st))))
        rows.append(dict(kernel=f"conv{li + 1}_dgrad", composite=False, flops=flops, bytes=io_bytes, name="conv3x3_v2_kernel",
                         ms=timeit(lambda: lib.call("hyb_conv3x3_fwd", dt_code, 0, dy.data_ptr(), wpd.data_ptr(), dx.data_ptr(), None, None,
                                                    N, H, H, co, cop, cip, st))))
        # dw = NULL: only the contraction kernel (partial slabs), without the fixed-order slab reduce
        rows.append(dict(kernel=f"conv{li + 1}_wgrad", composite=False, flops=flops, bytes=io_bytes,
                         name="wgrad_v2_kernel<false, %d>" % (64 if cip % 64 == 0 else 32),
                         ms=timeit(lambda: lib.call("hyb_conv3x3_wgrad", dt_code, 0, x.data_ptr(), dy.data_ptr(), None, N, H, H, ci, cip, co, cop,
                                                    ws.data_ptr(), nb, st))))
        del x, y, dy, ws, dx
        H //= 2
    return rows


def instep_kernel_table(args, step_fn, nsteps=8):
    """Time the nine conv contraction kernels INSIDE real training steps: the library records HIP events on the launch
    stream immediately around each hooked kernel (hyb_profile_set), one synchronised step at a time, outside the headline
    timed region so the headline is not perturbed."""
    _import_torch()
    from transformer_cnn_hybrid_network_for_video_processing_amd._lib import lib
    N = args.batch * args.frames
    chans = (3,) + CFG["cnn_channels"]
    specs = []
    H = args.size // 2
    bf16 = getattr(args, "dtype", "bf16") in ("bf16", "mixed")        # (mixed: the conv stages are the bf16 mode's)
    es = 2 if bf16 else 4                             # fp32 and bf16x3 store fp32 activations
    for li in range(1, 4):
        ci, co = chans[li], chans[li + 1]
        flops = 2.0 * 9 * ci * co * H * H * N
        io = float(N * H * H * (ci + co) * es)
        # kernel names as rocprofv3 lists them in this mode (fp32 / bf16x3: the float instances of the first-generation templates)
        conv_name = "conv3x3_v2_kernel" if bf16 else "conv3x3_nhwc_kernel<float>"
        specs.append((f"conv{li + 1}_fwd", 1, ci, co, conv_name, flops, io))
        specs.append((f"conv{li + 1}_dgrad", 1, co, ci, conv_name, flops, io))
        # the wgrad kernel is fused with the BN/ReLU/pool backward: reads x, raw conv output y, dpooled; writes the dense gradient + dW
        io_w = float(N * H * H * (ci + 2 * co) * es + N * (H // 2) * (H // 2) * co * es)
        # (the third-generation kernel takes 64-channel blocks on images of 28 k columns, conv_wgrad_v3.h)
        gen3 = ci % 64 == 0 and co % 64 == 0 and H % 28 == 0 and H % 4 == 0 and os.environ.get("HYB_WGRAD_V3", "1") != "0"
        wname = ("wgrad_v3_kernel" if gen3 else "wgrad_v2_kernel<true, %d>" % (64 if ci % 64 == 0 else 32)) if bf16 else \
            "conv3x3_wgrad_kernel<float, %d, true>" % (4 if ci % 64 == 0 else 2)
        specs.append((f"conv{li + 1}_wgrad", 2, ci, co, wname, flops, io_w))
        H //= 2
    evs = []
    for slot, sp in enumerate(specs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()                       # force creation of the underlying hipEvent_t
        evs.append((e0, e1))
    torch.cuda.synchronize()
    for slot, sp in enumerate(specs):
        lib.call("hyb_profile_set", slot, sp[1], sp[2], sp[3], evs[slot][0].cuda_event, evs[slot][1].cuda_event)
    acc = [0.0] * len(specs)
    try:
        for _ in range(nsteps):
            step_fn()
            torch.cuda.synchronize()
            for i, (e0, e1) in enumerate(evs):
                acc[i] += e0.elapsed_time(e1)
    finally:
        lib.call("hyb_profile_clear")
    return [dict(kernel=sp[0], composite=False, name=sp[4], flops=sp[5], bytes=sp[6], ms=acc[i] / nsteps) for i, sp in enumerate(specs)]


def dominant_kernel(rows):
    """The kernel NAME with the largest total time per step (what rocprofv3 --stats ranks first among the contractions)."""
    tot = {}
    for r in rows:
        if r.get("composite"):
            continue
        key = r["name"] if "wgrad" in r["name"] else r["kernel"]      # fwd/dgrad template instances differ per layer
        tot.setdefault(key, []).append(r)
    key = max(tot, key=lambda k: sum(r["ms"] for r in tot[k]))
    grp = tot[key]
    ms = sum(r["ms"] for r in grp) / len(grp)
    return dict(name=key, launches_per_step=len(grp), layers=[r["kernel"] for r in grp], ms=ms, flops=grp[0]["flops"],
                bytes=sum(r["bytes"] for r in grp) / len(grp))


def host_cores():
    """CPU cores this process may actually use (cgroup quota / affinity), not the machine's core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_model_name():
    """The host CPU as `lscpu` names it (the CPU baseline states the CPU model and the cores used)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    try:
        out = subprocess.run(["lscpu"], stdout=subprocess.PIPE, text=True, timeout=10).stdout
        for line in out.splitlines():
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


def cpu_baseline(args, budget_s=25.0):
    """The oracle (a port: stock torch fp32 on the host cores) on a bounded sample of the same workload: the configuration's own batch of
    clips, a warm-up step and up to four timed full steps inside the time budget (config 2: ~2.5 s per step on 16 cores, ~13 s in all)."""
    _import_torch()
    from oracle import hybrid_ref as R
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = R.TransformerCNNHybridRef(cnn_channels=CFG["cnn_channels"], d_model=args.d_model, num_heads=args.heads,
                                  num_layers=CFG["num_layers"], hidden_dim=args.hidden, num_classes=CFG["num_classes"])
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    nb = max(1, min(int(args.batch), 8))
    x, y = R.synthetic_batch(nb, args.frames, args.size, args.size, CFG["num_classes"], seed=0)

    def step():
        opt.zero_grad()
        R.loss_fn(m(x), y).backward()
        opt.step()
    step()
    times = []
    t_budget = time.time() + budget_s
    while len(times) < 4 and (not times or time.time() < t_budget):
        t0 = time.time()
        step()
        times.append(time.time() - t0)
    times.sort()
    med = times[len(times) // 2]
    return dict(value=nb / med, unit="clips/s", cores=torch.get_num_threads(), cpu=cpu_model_name(), kind="port",
                sample=f"oracle/hybrid_ref.py fp32, {nb} clip{'s' if nb > 1 else ''} [{nb},{args.frames},3,{args.size},{args.size}] x {len(times)} full steps "
                       f"(median {med * 1e3:.0f} ms per step) after 1 warm-up")


class EntryPointTimer:
    """HIP events (torch's current stream = the stream the library launches on) around every C-ABI call while active:
    per (entry point, shape key) call count and total GPU milliseconds -- live, inside real passes, no synchronisation per call."""

    KEYS = {"hyb_fct_mha_bwd": slice(14, 18), "hyb_fct_mha_fwd": slice(9, 13), "hyb_conv2d_bwd": slice(7, 17), "hyb_conv2d_fwd": slice(5, 15),
            "hyb_fct_conv_bwd": slice(8, 16), "hyb_fct_conv_fwd": slice(5, 12)}

    def __enter__(self):
        _import_torch()
        from transformer_cnn_hybrid_network_for_video_processing_amd import _lib
        self._lib, self._orig, self.rec = _lib, _lib._Lib.call, []
        timer = self

        def call(lib_self, name, *args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = timer._orig(lib_self, name, *args)
            e1.record()
            key = tuple(a for a in args[self.KEYS[name]]) if name in self.KEYS else ()
            timer.rec.append((name, key, args[4] is not None if name == "hyb_conv2d_bwd" else True, e0, e1))
            return r
        _lib._Lib.call = call
        return self

    def __exit__(self, *exc):
        self._lib._Lib.call = self._orig
        torch.cuda.synchronize()
        tot = {}
        for name, key, flag, e0, e1 in self.rec:
            t = tot.setdefault((name, key, flag), [0, 0.0])
            t[0] += 1
            t[1] += e0.elapsed_time(e1)
        self.totals = tot
        return False


def entry_point_roofline(totals, passes):
    """The heaviest (entry point, shape) of a pass against the fp32 matrix peak (these rows compute in fp32)."""
    by_name = {}
    for (n_, _, _), (_, t_) in totals.items():
        by_name[n_] = by_name.get(n_, 0.0) + t_
    top = max(by_name, key=by_name.get)                                     # the entry point with the largest share of the pass ...
    (name, key, flag), (calls, ms) = max(((k, v) for k, v in totals.items() if k[0] == top), key=lambda kv: kv[1][1])      # ... and its heaviest shape
    ms_call = ms / calls
    flops, what = None, name
    if name in ("hyb_fct_mha_bwd", "hyb_fct_mha_fwd"):
        N, L, C, heads = key
        flops = (10.0 if name.endswith("bwd") else 4.0) * N * L * L * C + (16.0 if name.endswith("bwd") else 8.0) * N * L * C * C
        what = f"{name} N={N} L={L} C={C} heads={heads}: one call = the attention kernels over {L}-token sequences + the in/out projections"
    elif name in ("hyb_conv2d_bwd", "hyb_conv2d_fwd"):
        N, H, W, Ci, Co, k, stride, pad, dil = key[:9]
        Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
        flops = 2.0 * N * Ho * Wo * Co * Ci * k * k * ((2 if flag else 1) if name.endswith("bwd") else 1)
        what = f"{name} N={N} {H}x{W} Ci={Ci} Co={Co} k={k} stride={stride}" + (" (dx + dw)" if name.endswith("bwd") and flag else "")
    total = sum(v[1] for v in totals.values())
    r = {"entry_point": what, "calls_per_pass": calls // passes, "ms_per_call": ms_call, "share_of_pass": ms / total,
         "entry_point_share_of_pass": by_name[top] / total, "bound": "mfma", "peak": MFMA_F32_PEAK_TFLOPS,
         "unit": "TFLOP/s", "traffic": None,
         "how": "HIP events recorded on the launch stream around every C-ABI call inside real training passes; the heaviest (entry point, shape)"}
    if flops is not None:
        r.update(achieved=flops / (ms_call * 1e-3) / 1e12, frac=flops / (ms_call * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, flops_per_call=flops)
    return r


def committed_traffic(kernel, cfgno, dtype):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this round (profiles/r04_traffic.json: rocprofv3 --pmc FETCH_SIZE x 2 +
    WRITE_SIZE, separate passes, per configuration and mode) -- (bytes, source) or (None, None)."""
    if dtype == "mixed":          # the conv stages of `mixed` are the bf16 mode's kernels on the same tensors
        dtype = "bf16"
    for tname in ("r04_traffic.json",):
        tpath = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tpath):
            t = json.load(open(tpath)).get(f"config{cfgno}_{dtype}", {}).get(kernel)
            if t:
                return t["hbm_bytes_per_launch"], (f"profiles/{tname} [config{cfgno}_{dtype}] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                                   f"`bench.py --config {cfgno} --dtype {dtype} --eager`; not re-measured in this run)")
    return None, None


def committed_traffic_f(leg, kernel):
    """HBM bytes of the heaviest launch of `kernel` in one training pass of the fct / enc32k benches, from the committed counter passes
    (profiles/r04_traffic_f.json: rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate passes; scripts/collect_pmc_f_r04.sh) -- (bytes, source)."""
    tpath = os.path.join(ROOT, "profiles", "r04_traffic_f.json")
    if os.path.exists(tpath):
        for k, v in json.load(open(tpath)).get(leg, {}).items():
            if kernel in k:
                return v["hbm_bytes_per_launch_max"], ("profiles/r04_traffic_f.json [%s][%s]: the launch of this kernel name with the most traffic in the pass "
                                                       "(the hooked launch is the heaviest of its name); not re-measured in this run" % (leg, k))
    return None, None


def hooked_kernel_roofline(kernel_id, a, b, flops, name, run_pass, passes=2):
    """One kernel of a pass against the fp32 matrix peak, timed by the library's own HIP events around that kernel alone (hyb_profile_set) inside
    real passes -- the last matching launch of each pass."""
    from transformer_cnn_hybrid_network_for_video_processing_amd._lib import lib
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()
    torch.cuda.synchronize()
    lib.call("hyb_profile_set", 0, kernel_id, a, b, e0.cuda_event, e1.cuda_event)
    ms = 0.0
    try:
        for _ in range(passes):
            run_pass()
            torch.cuda.synchronize()
            ms += e0.elapsed_time(e1)
    finally:
        lib.call("hyb_profile_clear")
    ms /= passes
    if not (ms > 1e-4):
        return None
    ach = flops / (ms * 1e-3) / 1e12
    return {"kernel": name, "bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TFLOPS,
            "traffic": None, "ms": ms, "flops_per_launch": flops,
            "how": f"HIP events recorded by the library around this kernel alone (hyb_profile_set, kernel {kernel_id}, key ({a}, {b})) inside {passes} real passes"}


def model_leg(cfgno, dtype, dev, steps, warmup, want_roofline):
    """A short run of one more configuration / precision mode: the same full step as the headline (graph replay when capture works)."""
    _import_torch()
    import transformer_cnn_hybrid_network_for_video_processing_amd as P
    from transformer_cnn_hybrid_network_for_video_processing_amd import ops
    c = CONFIGS[cfgno]
    ns = argparse.Namespace(batch=c["batch"], frames=c["frames"], size=c["size"], d_model=c["d_model"], heads=c["num_heads"], hidden=c["hidden_dim"], dtype=dtype,
                            config=cfgno)
    torch.manual_seed(0)
    model = P.TransformerCNNHybrid(cnn_channels=CFG["cnn_channels"], d_model=ns.d_model, num_heads=ns.heads, num_layers=CFG["num_layers"],
                                   hidden_dim=ns.hidden, num_classes=CFG["num_classes"], dropout=0.0, compute_dtype=dtype).to(dev).train()
    crit, opt = P.HybridCrossEntropyLoss(), None
    opt = P.HybridAdamW(model.parameters(), lr=1e-3)
    g = torch.Generator(device="cpu").manual_seed(1000)
    x = torch.rand(ns.batch, ns.frames, 3, ns.size, ns.size, generator=g).to(dev)
    y = torch.randint(0, CFG["num_classes"], (ns.batch,), generator=g).to(dev)
    trainer, fallback = None, None
    try:
        trainer = P.GraphedTrainStep(model, crit, opt, x, y)
        step, fwd_bwd = trainer.step, trainer.eager_fwd_bwd
    except Exception as e:                                                  # noqa: BLE001 (reported)
        fallback = f"{type(e).__name__}: {e}"[:200]
        ops.set_step_counter(None); opt.set_step_counter(None)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = crit(model(x), y)
            loss.backward()
            opt.step()
            return loss
        fwd_bwd = step
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"value": ns.batch * steps / dt, "unit": "clips/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup, "dtype": dtype,
           "workload": f"config {cfgno}: clips [{ns.batch},{ns.frames},3,{ns.size},{ns.size}], d={ns.d_model} h={ns.heads} hid={ns.hidden}; full step",
           "graph_fallback": fallback is not None, "final_loss": float(loss.item())}
    if fallback:
        res["graph_fallback_reason"] = fallback
    if want_roofline:
        rows = instep_kernel_table(ns, fwd_bwd, nsteps=4)
        dom = dominant_kernel(rows)
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        traffic, tsrc = committed_traffic(dom["name"], cfgno, dtype)
        res["roofline"] = {"kernel": dom["name"], "layers": dom["layers"], "bound": "mfma", "achieved": ach, "peak": MODE_PEAK[dtype], "unit": "TFLOP/s",
                           "frac": ach / MODE_PEAK[dtype], "traffic": traffic, "traffic_source": tsrc, "ms": dom["ms"], "flops_per_launch": dom["flops"],
                           "algorithmic_bytes_per_launch": dom["bytes"],
                           "how": "HIP events recorded by the library around this kernel inside 4 real steps (hyb_profile_set)"}
        if dtype == "bf16x3":
            res["roofline"]["peak_note"] = "2500 / 3 TFLOP/s: three bf16 MFMAs per product in this mode"
        if dtype == "mixed":
            res["roofline"]["peak_note"] = "the dominant kernel belongs to the bf16 conv stages: priced against the bf16 peak"
        assert 0.0 < res["roofline"]["frac"] < 1.0, res["roofline"]       # (an events-timed-nothing hook once reported 83.6)
    if trainer is not None:
        trainer.close()
    ops.set_step_counter(None)
    del model, opt, x, y, trainer
    gc.collect()
    torch.cuda.empty_cache()
    return res


def fct_leg(dev, reps=5, frames=16, size=224):
    """FCT: the reference's training step (FCT.py:328-338: forward, DiceLoss, backward) on frame-folded clips, fp32."""
    _import_torch()
    import transformer_cnn_hybrid_network_for_video_processing_amd as P
    torch.manual_seed(0)
    m = P.FCT().to(dev).train()
    crit = P.DiceLoss()
    x = torch.rand(frames, 3, size, size, device=dev)
    yt = (torch.rand(frames, 1, size, size, device=dev) > 0.5).float()

    def train_pass():
        for p_ in m.parameters():
            p_.grad = None
        crit(m(x), yt).backward()
    train_pass(); train_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        train_pass()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    with EntryPointTimer() as ept:
        train_pass(); train_pass()
    res = {"value": frames / ms * 1e3, "unit": "frames/s", "ms_per_pass": ms, "dtype": "f32",
           "workload": f"FCT training pass (forward + DiceLoss + backward, train mode), frames [{frames},3,{size},{size}]",
           "roofline": entry_point_roofline(ept.totals, 2)}
    # the heaviest KERNEL inside that entry point, by itself: the dK / dV kernel of the attention backward over the longest sequences
    mha = [k for k in ept.totals if k[0] == "hyb_fct_mha_bwd"]
    if mha:
        (_, (N_, L_, C_, heads_), _), _ = max(((k, v) for k, v in ept.totals.items() if k[0] == "hyb_fct_mha_bwd"), key=lambda kv: kv[1][1])
        kr = hooked_kernel_roofline(5, int(L_), int(heads_), 8.0 * N_ * L_ * L_ * C_, "flash_bwd4_dkv_kernel", train_pass)
        if kr:
            kr["traffic"], kr["traffic_source"] = committed_traffic_f("fct", "flash_bwd4_dkv_kernel<1>")
            kr["what"] = (f"dK / dV kernel of FCT's attention backward, N={N_} L={L_} C={C_} heads={heads_}: recomputes S = QK^T, dP = dO V^T and forms "
                          "dV = P^T dO, dK = dS^T Q -- 8 L^2 C FLOP per image")
            res["kernel_roofline"] = kr
    from oracle import fct_ref as F
    cores = host_cores()
    torch.set_num_threads(cores)
    ref = F.FCT().train()
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    n = 2
    xc, yc = x[:n].cpu(), yt[:n].cpu()
    F.DiceLoss()(ref(xc), yc).backward()
    t0 = time.time()
    F.DiceLoss()(ref(xc), yc).backward()
    dt = time.time() - t0
    res["cpu_baseline"] = {"value": n / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "cpu": cpu_model_name(), "kind": "port",
                           "sample": f"oracle/fct_ref.py, one training pass on {n} frames [{n},3,{size},{size}] after 1 warm-up ({dt * 1e3:.0f} ms)"}
    del m, x, yt
    gc.collect(); torch.cuda.empty_cache()
    return res


def enc32k_leg(dev, reps=5, frames=16):
    """Encoder_32K: training pass (train-mode forward with Dropout2d + backward) on frame-folded 256 x 256 frames, fp32."""
    _import_torch()
    import transformer_cnn_hybrid_network_for_video_processing_amd as P
    torch.manual_seed(0)
    m = P.Encoder_32K().to(dev).train()
    x = torch.rand(frames, 3, 256, 256, device=dev)

    def train_pass():
        for p_ in m.parameters():
            p_.grad = None
        m(x).square().mean().backward()
    train_pass(); train_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        train_pass()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    with EntryPointTimer() as ept:
        train_pass(); train_pass()
    res = {"value": frames / ms * 1e3, "unit": "frames/s", "ms_per_pass": ms, "dtype": "f32",
           "workload": f"Encoder_32K training pass (forward + backward, train mode), frames [{frames},3,256,256] -> tokens [{frames},8,4096]; 72 GFLOP per frame",
           "tflops": 72.0 * frames / ms, "roofline": entry_point_roofline(ept.totals, 2)}
    # the heaviest KERNEL by itself: the implicit-GEMM input gradient of the heaviest stride-1 convolution (gemm_nt_lds_kernel)
    convs = [(k, v) for k, v in ept.totals.items() if k[0] == "hyb_conv2d_bwd" and k[2] and k[1][6] == 1 and k[1][5] > 1]
    if convs:
        (_, key, _), _ = max(convs, key=lambda kv: kv[1][1])
        N_, H_, W_, Ci_, Co_, k_, stride_, pad_, dil_ = key[:9]
        Co8 = (Co_ + 7) // 8 * 8
        Kp = k_ * k_ * Co8                                              # K of the input-gradient GEMM as hyb_conv2d_bwd launches it
        Ho = (H_ + 2 * pad_ - dil_ * (k_ - 1) - 1) // stride_ + 1
        Wo = (W_ + 2 * pad_ - dil_ * (k_ - 1) - 1) // stride_ + 1
        # hyb_conv2d_bwd works in image chunks of <= 512 MB of scratch; the events time the LAST chunk's launch
        per_img = Ho * Wo * ((k_ * k_ * Ci_ + 7) // 8 * 8 + Co8) * 4
        nb = max(1, min(N_, (512 << 20) // per_img))
        last = N_ - nb * ((N_ - 1) // nb)
        kr = hooked_kernel_roofline(4, int(Ci_), int(Kp), 2.0 * last * H_ * W_ * Ci_ * Kp,
                                    "gemm_nt_lds_kernel<%d, %d, true>" % ((128, 2) if Ci_ > 64 else (64, 3)) if Ci_ >= 64 else "gemm_nt_tall_kernel<4, true>", train_pass)
        if kr:
            kr["traffic"], kr["traffic_source"] = committed_traffic_f("enc32k", kr["kernel"])
            kr["algorithmic_bytes_per_launch"] = float(last * H_ * W_ * (Ci_ + Co8) * 4)
            kr["what"] = (f"input gradient of the {k_}x{k_} convolution {Ci_} -> {Co_} on {last} x {H_}x{W_} pixels (the last image chunk of {N_}) as an "
                          f"implicit GEMM: rows = input pixels, columns = {Ci_}, K = {Kp}")
            res["kernel_roofline"] = kr
    from oracle import encoder32k_ref as E
    cores = host_cores()
    torch.set_num_threads(cores)
    n = 2
    p = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    for k, v in p.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_()
    xc = x[:n].cpu()
    E.forward(p, xc, True).square().mean().backward()
    t0 = time.time()
    E.forward(p, xc, True).square().mean().backward()
    dt = time.time() - t0
    res["cpu_baseline"] = {"value": n / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "cpu": cpu_model_name(), "kind": "port",
                           "sample": f"oracle/encoder32k_ref.py, one training pass on {n} frames after 1 warm-up ({dt * 1e3:.0f} ms)"}
    del m, x
    gc.collect(); torch.cuda.empty_cache()
    return res


def config1_cpu_baseline():
    """BASELINE.json config 1: the reference's own CPU-runnable case, one clip [1,8,3,112,112] through the oracle (plumbing, no GPU)."""
    ns = argparse.Namespace(frames=8, size=112, d_model=512, heads=8, hidden=2048, batch=1)
    r = cpu_baseline(ns, budget_s=10.0)
    r["workload"] = "config 1: clips [1,8,3,112,112], d=512 h=8 hid=2048"
    return r


def bf16_logits_check(dev):
    """The bf16 mode's forward error of THIS build in THIS run: 2 clips [2,16,3,224,224] (train-mode BatchNorm, dropout off) against the
    fp32 CPU oracle on the same weights; max|dlogits| / max|logits| (the statistic of tests/test_gpu_fullsize.py)."""
    _import_torch()
    import transformer_cnn_hybrid_network_for_video_processing_amd as P
    from oracle import hybrid_ref as R
    torch.set_num_threads(host_cores())
    torch.manual_seed(0)
    ref = R.TransformerCNNHybridRef(cnn_channels=CFG["cnn_channels"], d_model=512, num_heads=8, num_layers=2, hidden_dim=2048, num_classes=8)
    for a in ref.encoder.attention_layers:
        a.dropoutLayer.p = 0.0
    ref.train()
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    x, _ = R.synthetic_batch(2, 16, 224, 224, seed=0)
    with torch.no_grad():
        lr = ref(x)
    out = {}
    for mode in ("bf16", "mixed", "bf16x3", "fp32"):
        m = P.TransformerCNNHybrid(compute_dtype=mode)
        m.load_state_dict(sd)
        for a in m.encoder.attention_layers:
            a.dropoutLayer.p = 0.0
        m = m.to(dev).train()
        with torch.no_grad():
            lh = m(x.to(dev))
        out[mode] = float(((lh.cpu() - lr).abs().max() / lr.abs().max()).item())
        del m
    out["what"] = "max|dlogits| / max|logits| vs oracle/hybrid_ref.py (fp32, host) on 2 clips [2,16,3,224,224], same weights, measured in this run"
    return out


def live_traffic(kernel_name, timeout_s=150):
    """HBM bytes per launch of `kernel_name`, MEASURED IN THIS RUN: two child runs of this script (3 eager steps each) under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (counters only, separate passes, as MI355X_MICROARCH.md prescribes; FETCH_SIZE
    doubled: the gfx950 correction).  Returns (bytes, how) or (None, reason)."""
    import csv, glob, shutil, tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="hyb_pmc_", dir="/tmp")
        try:
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--eager",
                   "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-fwd-bwd-only", "--no-pipeline", "--no-extra-legs"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} exited {r.returncode}"
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, "no counter_collection.csv"
            v = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                 if row["Counter_Name"] == counter and kernel_name.split("<")[0] in row["Kernel_Name"] and "reduce" not in row["Kernel_Name"]]
            if kernel_name.startswith("wgrad_v2_kernel<"):      # template instances of one name: keep the block width asked for
                want = kernel_name.split(",")[1].strip(" >")
                v = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                     if row["Counter_Name"] == counter and "wgrad_v2_kernel<" in row["Kernel_Name"] and f", {want}," in row["Kernel_Name"]]
            if not v:
                return None, f"kernel {kernel_name} not in the {counter} pass"
            vals[counter] = sum(v) / len(v)
        except subprocess.TimeoutExpired:
            return None, f"rocprofv3 --pmc {counter} timed out"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024, (
        "measured in this run: child runs of this script (3 eager steps) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate "
        "passes); FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, KiB units, average over this kernel's launches")


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))                       # before ANY torch / HIP call in this process
    _import_torch()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the numbers would not be what the command line says")
    single_dev = bool(os.environ.get("HYB_SINGLE_DEVICE"))        # test hook: several ranks share cuda:0 (one-GPU box, gloo backend)
    if not single_dev and torch.cuda.device_count() < world:      # (device_count does not initialise the GPU runtime)
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) visible")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if single_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("HYB_DIST_BACKEND", "nccl")      # "nccl" is RCCL over xGMI on ROCm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    import transformer_cnn_hybrid_network_for_video_processing_amd as P
    from transformer_cnn_hybrid_network_for_video_processing_amd.dp import GradAllReducer
    from transformer_cnn_hybrid_network_for_video_processing_amd import ops

    torch.manual_seed(0)                     # identical weights on every rank (also broadcast below); dropout seeds differ per rank (ops.next_seed)
    model = P.TransformerCNNHybrid(cnn_channels=CFG["cnn_channels"], d_model=args.d_model, num_heads=args.heads,
                                   num_layers=CFG["num_layers"], hidden_dim=args.hidden, num_classes=CFG["num_classes"],
                                   dropout=0.0, compute_dtype=args.dtype).to(dev)
    model.train()
    crit = P.HybridCrossEntropyLoss()
    if args.optimizer == "hybrid":
        opt = P.HybridAdamW(model.parameters(), lr=1e-3)                      # same update as torch.optim.AdamW, one launch
    else:
        try:
            opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)  # stock torch optimizer, fused multi-tensor kernels
        except Exception:
            opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    graphed = not args.eager and args.optimizer == "hybrid"
    reducer = GradAllReducer(model) if world > 1 and not graphed else None

    g = torch.Generator(device="cpu").manual_seed(1000 + rank)          # rank r seeds its own clips
    x = torch.rand(args.batch, args.frames, 3, args.size, args.size, generator=g).to(dev)
    y = torch.randint(0, CFG["num_classes"], (args.batch,), generator=g).to(dev)

    graph_fallback = None
    if graphed:
        # the whole step captured once as three hipGraphs (forward + temporal backward | backbone backward | AdamW); the gradient
        # all-reduce runs between them, outside the graphs, overlapped with the backbone backward (graph.py).  If capture fails on
        # ANY rank (it has only ever run on one-GPU boxes), every rank falls back to the eager step -- same kernels, same arithmetic --
        # and the JSON line says so, rather than losing the measurement.
        trainer = None
        try:
            if os.environ.get("HYB_BENCH_FORCE_GRAPH_FAIL"):                # test hook for the fallback path (tests/test_gpu_dp.py)
                raise RuntimeError("forced by HYB_BENCH_FORCE_GRAPH_FAIL")
            trainer = P.GraphedTrainStep(model, crit, opt, x, y)
        except Exception as e:                                             # noqa: BLE001 (reported in the output line)
            graph_fallback = f"{type(e).__name__}: {e}"[:300]
        if world > 1:
            ok = torch.tensor([0 if graph_fallback else 1], device=dev, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and graph_fallback is None:
                graph_fallback = "graph capture failed on another rank"
        if graph_fallback is not None:
            if trainer is not None:
                trainer.close()
            from transformer_cnn_hybrid_network_for_video_processing_amd import ops as _ops
            _ops.set_step_counter(None)
            opt.set_step_counter(None)
            for p_ in model.parameters():
                p_.grad = None
            trainer, graphed = None, False
            reducer = GradAllReducer(model) if world > 1 else None
    if graphed:
        step, fwd_bwd = trainer.step, trainer.fwd_bwd
    else:
        def step():
            opt.zero_grad(set_to_none=True)
            loss = crit(model(x), y)
            loss.backward()
            if reducer is not None:
                reducer.finalize()
            opt.step()
            return loss

        def fwd_bwd():                              # the metric's literal "fwd+bwd": gradients (all-reduced when N>1), no optimizer
            for p in params:
                p.grad = None
            loss = crit(model(x), y)
            loss.backward()
            if reducer is not None:
                reducer.finalize()
            return loss
    params = list(model.parameters())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, n):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    for _ in range(args.warmup):
        step()
    # Host policy of a training loop, not a shortcut: everything allocated so far (modules, operator registrations, the import graph)
    # is long-lived, so move it to the permanent generation.  Without this CPython's cyclic GC re-traverses it during the loop: measured
    # +0.9 ms host time per step and a 166 ms full-collection pause every ~25 steps (scripts/host_jitter.py), which would make the
    # step host-bound.  Young objects are still collected.
    gc.collect()
    gc.freeze()
    dt, loss = timed(step, args.steps)
    final_loss = float(loss.item())
    fb = None
    if not args.no_fwd_bwd_only:
        fwd_bwd()
        fdt, _ = timed(fwd_bwd, args.steps)
        fb = {"value": args.batch * world * args.steps / fdt, "unit": "clips/s", "ms_per_step": fdt / args.steps * 1e3,
              "step": "fwd+cross_entropy+bwd" + ("+grad_allreduce" if world > 1 else "") + " (no zero_grad, no optimizer); same K steps, timed after the headline"}

    fed = None
    if not args.no_pipeline and world == 1:
        # PCIe-inclusive rate (never `value`): every step takes a NEW host batch -- uint8 frames in pinned memory, async H2D on the copy
        # stream, ToTensor on the device, two batches ahead (clips.ClipPipeline) -- instead of HBM-resident clips
        import itertools
        src = P.SyntheticClipSource(args.batch, args.frames, args.size, CFG["num_classes"], seed=1000 + rank, distinct=3)
        pipe = iter(P.ClipPipeline(itertools.islice(iter(src), args.steps + 3), device=dev, depth=2))

        def fed_step():
            xb, yb = next(pipe)
            if graphed:
                trainer.load(xb, yb)
                return trainer.step()
            opt.zero_grad(set_to_none=True)
            loss = crit(model(xb), yb)
            loss.backward()
            opt.step()
            return loss
        for _ in range(3):
            fed_step()
        pdt, _ = timed(fed_step, args.steps)
        fed = {"value": args.batch * args.steps / pdt, "unit": "clips/s", "ms_per_step": pdt / args.steps * 1e3,
               "what": "same step, but each batch starts as uint8 [B,T,H,W,3] in pinned host memory and crosses PCIe (clips.ClipPipeline, "
                       "depth 2); not the headline value"}

    if rank == 0:
        ms = dt / args.steps * 1e3
        labels = {2: "config 2", 4: "config 4 (long clip)", 5: "config 5 (high-res)"}
        is_cfg = all(getattr(args, k) == v for k, v in (("batch", CONFIGS[args.config]["batch"]), ("frames", CONFIGS[args.config]["frames"]),
                                                        ("size", CONFIGS[args.config]["size"]), ("d_model", CONFIGS[args.config]["d_model"]),
                                                        ("heads", CONFIGS[args.config]["num_heads"]), ("hidden", CONFIGS[args.config]["hidden_dim"])))
        out = {
            "metric": "clips/sec fwd+bwd, [B=8,T=16,3,224,224] d=512, 1/2/4/8 MI355X",
            "value": args.batch * world * args.steps / dt,
            "unit": "clips/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{labels[args.config] if is_cfg else 'custom'}: clips [{args.batch},{args.frames},3,{args.size},{args.size}] per GPU, "
                                   f"CNN 32-64-128-256 + 2-layer transformer d={args.d_model} h={args.heads} hid={args.hidden}, 8 classes",
                       "global_batch": args.batch * world, "frames": args.frames,
                       "step": "zero_grad+fwd+cross_entropy+bwd+grad_allreduce+adamw",
                       "launch": (("1 replayed hipGraph per step" if world == 1 else "3 replayed hipGraphs per step, the gradient all-reduces between them") + " (graph.GraphedTrainStep)") if graphed else "eager (one Python-issued launch per kernel)" + (f"; graph capture fell back: {graph_fallback}" if graph_fallback else ""),
                       "optimizer": "HybridAdamW (hyb_adamw_step, one launch)" if args.optimizer == "hybrid" else "torch.optim.AdamW(fused=True)",
                       "parallelism": f"dp{world}",
                       "train_mode": "BatchNorm batch stats, attention dropout 0.1 (reference semantics)",
                       "logits_parity": "north_star's 1e-3 rel vs the CPU oracle is met by compute_dtype fp32 (exact fp32 MFMA), bf16x3 (split-bf16 "
                                        "products) and mixed (bf16 conv stages + bf16x3 temporal part); bf16 mode is at bf16 rounding level -- all four "
                                        "measured in this run: key logits_check"},
            "final_loss": final_loss,
            "graph_fallback": bool(graph_fallback),
        }
        if fb is not None:
            out["fwd_bwd_only"] = fb
        if fed is not None:
            out["pcie_inclusive"] = fed
        if not args.no_roofline and world == 1:
            rows = instep_kernel_table(args, trainer.eager_fwd_bwd if graphed else step)
            dom = dominant_kernel(rows)
            peak = MODE_PEAK[args.dtype]
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            traffic, tsrc = committed_traffic(dom["name"], args.config, args.dtype) if is_cfg else (None, None)
            for tname in (() if traffic is not None else ("r03_traffic.json", "r02_traffic.json", "r01_traffic.json")):           # PMC bytes per launch (rocprofv3 --pmc passes of this command)
                tpath = os.path.join(ROOT, "profiles", tname)
                if os.path.exists(tpath) and args.dtype == "bf16" and args.config == 2 and is_cfg:
                    t = json.load(open(tpath)).get(dom["name"])
                    if t:
                        traffic, tsrc = t["hbm_bytes_per_launch"], f"profiles/{tname} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not re-measured in this run)"
                        break
            if not args.no_live_traffic and args.dtype == "bf16" and args.config == 2 and is_cfg and not args.no_extra_legs:
                lt, how = live_traffic(dom["name"])
                if lt is not None:
                    traffic, tsrc = lt, how
                else:
                    tsrc = (tsrc or "none") + f" [live PMC measurement unavailable: {how}]"
            out["roofline"] = {"kernel": dom["name"], "layers": dom["layers"], "launches_per_step": dom["launches_per_step"],
                               "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                               "traffic": traffic, "traffic_source": tsrc, "ms": dom["ms"], "flops_per_launch": dom["flops"],
                               "algorithmic_bytes_per_launch": dom["bytes"],
                               "how": "HIP events recorded by the library around this kernel inside 8 real steps (hyb_profile_set)"}
            out["kernel_table"] = [{"kernel": r["kernel"], "ms": round(r["ms"], 4), "TFLOPs": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2),
                                    "GBps": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)} for r in rows]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        # the extra legs belong to the plain command (what the driver runs); the trimmed forms the profiling scripts use skip them
        if not (args.no_extra_legs or args.no_roofline or args.no_cpu_baseline or args.eager) and world == 1 and is_cfg and args.config == 2 and args.dtype == "bf16":
            # free the headline's model and graphs first: config 4 wants the memory
            if graphed:
                trainer.close()
            ops.set_step_counter(None)
            del model, opt, x, y
            gc.collect()
            torch.cuda.empty_cache()
            legs = (("config4", lambda: model_leg(4, "bf16", dev, 10, 3, True)), ("config5", lambda: model_leg(5, "bf16", dev, 10, 3, True)),
                    ("fp32", lambda: model_leg(2, "fp32", dev, 10, 3, False)), ("bf16x3", lambda: model_leg(2, "bf16x3", dev, 10, 3, True)),
                    ("mixed", lambda: model_leg(2, "mixed", dev, 20, 5, True)), ("fct", lambda: fct_leg(dev)), ("enc32k", lambda: enc32k_leg(dev)), ("logits_check", lambda: bf16_logits_check(dev)),
                    ("cpu_baseline_config1", config1_cpu_baseline))
            for key, fn in legs:
                if key == "cpu_baseline_config1" and args.no_cpu_baseline:
                    continue
                try:
                    out[key] = fn()
                except Exception as e:                                      # noqa: BLE001 (a failed extra leg must not lose the headline)
                    out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
            # the headline mode (bf16, BASELINE config 2's dtype) is at bf16 rounding level; north_star's "logits within 1e-3 rel of the CPU
            # reference" is met by the bf16x3 mode: its throughput on the SAME workload, and the errors of both measured in this run
            lc = out.get("logits_check", {})
            ok = [(out[m]["value"], m) for m in ("mixed", "bf16x3", "fp32") if "value" in out.get(m, {}) and lc.get(m, 1.0) <= 1e-3]
            if ok:
                _, best = max(ok)
                out["value_at_north_star_tolerance"] = {
                    "value": out[best]["value"], "unit": "clips/s", "dtype": best, "ms_per_step": out[best]["ms_per_step"], "logits_rel_err": lc[best],
                    "tolerance": 1e-3, "headline_logits_rel_err": lc.get("bf16"),
                    "what": "same config-2 full step in the fastest compute_dtype whose forward logits are within north_star's 1e-3 of the CPU oracle "
                            "(measured in this run, `logits_check`): 'mixed' = bf16 conv stages + split-bf16 (bf16x3) temporal part, 'bf16x3' = "
                            "fp32 storage with split-bf16 MFMA products everywhere; the headline `value` is the bf16 mode, whose logits error is "
                            "`headline_logits_rel_err`"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
