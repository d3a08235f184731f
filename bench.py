#!/usr/bin/env python3
"""Benchmark of the hot path: clips/sec of one full harness step on BASELINE config 2
([B=8 per GPU, T=16, 3, 224, 224], 4-stage CNN 32-64-128-256 + 2-layer transformer d=512 h=8, bf16).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = zero_grad -> forward -> cross-entropy -> backward -> (RCCL gradient all-reduce when N>1) -> AdamW.step, i.e. the
reference harness inner loop (Model.py:55-59 / FCT.py:328-338) with nothing skipped; synthetic clips are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line (see the task contract) including

  "roofline":     the dominant HIP kernel, timed live with events on the launch stream, against the MI355X peak
  "cpu_baseline": the CPU oracle (oracle/hybrid_ref.py, kind "port") timed on the host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16 (MI355X_MICROARCH.md, chip-level parameters)
MFMA_F32_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0               # HBM3E spec peak (6.29 TB/s measured copy)

CFG = dict(B=8, T=16, H=224, W=224, cnn_channels=(32, 64, 128, 256), d_model=512, num_heads=8, num_layers=2,
           hidden_dim=2048, num_classes=8)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=CFG["B"], help="clips per GPU")
    ap.add_argument("--frames", type=int, default=CFG["T"])
    ap.add_argument("--size", type=int, default=CFG["H"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def conv_kernel_table(args, dt_code, tdt, dev):
    """Time every conv GEMM kernel of one step standalone through the C ABI (HIP events on the launch stream)."""
    from transformer_cnn_hybrid_network_for_video_processing_amd._lib import lib
    st = torch.cuda.current_stream().cuda_stream
    es = 2 if tdt == torch.bfloat16 else 4
    N = args.batch * args.frames
    chans = (3,) + CFG["cnn_channels"]
    rows = []
    H = args.size
    for li in range(4):
        ci, co = chans[li], chans[li + 1]
        first = li == 0
        cip, cop = (0 if first else ci), co
        x = torch.rand(N, ci, H, H, device=dev) if first else torch.rand(N, H, H, ci, device=dev).to(tdt)
        w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        wp = torch.empty(lib.query("hyb_conv_packed_elems", int(first), cip, cop), dtype=tdt, device=dev)
        lib.call("hyb_conv_pack_weight", dt_code, 2 if first else 0, w.data_ptr(), wp.data_ptr(), co, ci, cop, cip, st)
        y = torch.empty(N, H, H, cop, dtype=tdt, device=dev)
        stats = torch.zeros(2, cop, device=dev)
        part = torch.empty(lib.query("hyb_conv_stats_workspace", cop), dtype=torch.uint8, device=dev)
        flops = 2.0 * 9 * ci * co * H * H * N

        def t_fwd():
            lib.call("hyb_conv3x3_fwd", dt_code, int(first), x.data_ptr(), wp.data_ptr(), y.data_ptr(), stats.data_ptr(), part.data_ptr(), N, H, H, ci, cip, cop, st)
        rows.append(dict(kernel=f"conv{li + 1}_fwd", flops=flops, bytes=float(N * H * H * (ci * (4 if first else es) + co * es)), fn=t_fwd))
        dy = (torch.randn(N, H, H, cop, device=dev) * 0.1).to(tdt)
        nb = lib.query("hyb_conv3x3_wgrad_workspace", int(first), N, H, H, cip, cop)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        dw = torch.empty_like(w)

        def t_wgrad():
            lib.call("hyb_conv3x3_wgrad", dt_code, int(first), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, H, H, ci, cip, co, cop, ws.data_ptr(), nb, st)
        rows.append(dict(kernel=f"conv{li + 1}_wgrad", flops=flops, bytes=float(N * H * H * (ci * (4 if first else es) + co * es)), fn=t_wgrad))
        if not first:
            wpd = torch.empty(cip * 9 * cop, dtype=tdt, device=dev)
            lib.call("hyb_conv_pack_weight", dt_code, 1, w.data_ptr(), wpd.data_ptr(), co, ci, cop, cip, st)
            dx = torch.empty(N, H, H, cip, dtype=tdt, device=dev)

            def t_dgrad():
                lib.call("hyb_conv3x3_fwd", dt_code, 0, dy.data_ptr(), wpd.data_ptr(), dx.data_ptr(), None, None, N, H, H, co, cop, cip, st)
            rows.append(dict(kernel=f"conv{li + 1}_dgrad", flops=flops, bytes=float(N * H * H * (ci + co) * es), fn=t_dgrad))
        for r in rows:
            if "ms" in r:
                continue
            r["fn"](); r["fn"]()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                r["fn"]()
            e1.record()
            e1.synchronize()
            r["ms"] = e0.elapsed_time(e1) / reps
            del r["fn"]
        del x, y, dy, ws
        H //= 2
    return rows


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


def cpu_baseline(args):
    """The oracle (a port: stock torch fp32 on the host cores) on a bounded sample: 1 clip of the same shape."""
    from oracle import hybrid_ref as R
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = R.TransformerCNNHybridRef(cnn_channels=CFG["cnn_channels"], d_model=CFG["d_model"], num_heads=CFG["num_heads"],
                                  num_layers=CFG["num_layers"], hidden_dim=CFG["hidden_dim"], num_classes=CFG["num_classes"])
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    x, y = R.synthetic_batch(1, args.frames, args.size, args.size, CFG["num_classes"], seed=0)

    def step():
        opt.zero_grad()
        R.loss_fn(m(x), y).backward()
        opt.step()
    step()
    times = []
    t_budget = time.time() + 25.0
    while len(times) < 5 and time.time() < t_budget:
        t0 = time.time()
        step()
        times.append(time.time() - t0)
    times.sort()
    med = times[len(times) // 2]
    return dict(value=1.0 / med, unit="clips/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle/hybrid_ref.py fp32, 1 clip [1,{args.frames},3,{args.size},{args.size}] x {len(times)} full steps "
                       f"(median {med * 1e3:.0f} ms) after 1 warm-up")


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if os.environ.get("HYB_SINGLE_DEVICE"):       # test hook: several ranks share cuda:0 (one-GPU box, gloo backend)
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

    torch.manual_seed(0)                     # identical weights on every rank (also broadcast below)
    model = P.TransformerCNNHybrid(cnn_channels=CFG["cnn_channels"], d_model=CFG["d_model"], num_heads=CFG["num_heads"],
                                   num_layers=CFG["num_layers"], hidden_dim=CFG["hidden_dim"], num_classes=CFG["num_classes"],
                                   dropout=0.0, compute_dtype=args.dtype).to(dev)
    model.train()
    crit = P.HybridCrossEntropyLoss()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    reducer = GradAllReducer(model) if world > 1 else None

    g = torch.Generator(device="cpu").manual_seed(1000 + rank)          # rank r seeds its own clips
    x = torch.rand(args.batch, args.frames, 3, args.size, args.size, generator=g).to(dev)
    y = torch.randint(0, CFG["num_classes"], (args.batch,), generator=g).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = crit(model(x), y)
        loss.backward()
        if reducer is not None:
            reducer.finalize()
        opt.step()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
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
            "config": {"workload": f"config 2: clips [{args.batch},{args.frames},3,{args.size},{args.size}] per GPU, CNN 32-64-128-256 + "
                                   f"2-layer transformer d=512 h=8 hid=2048, 8 classes",
                       "global_batch": args.batch * world, "frames": args.frames,
                       "step": "zero_grad+fwd+cross_entropy+bwd+grad_allreduce+adamw", "parallelism": f"dp{world}",
                       "train_mode": "BatchNorm batch stats, attention dropout 0.1 (reference semantics)"},
            "final_loss": final_loss,
        }
        if not args.no_roofline and world == 1:
            dt_code = ops.dtype_code(args.dtype)
            rows = conv_kernel_table(args, dt_code, ops.torch_dtype(dt_code), dev)
            dom = max(rows, key=lambda r: r["ms"])
            peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            out["roofline"] = {"kernel": dom["kernel"], "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "frac": ach / peak, "traffic": None, "ms": dom["ms"],
                               "hbm_GBps_algorithmic": dom["bytes"] / (dom["ms"] * 1e-3) / 1e9}
            out["kernel_table"] = [{"kernel": r["kernel"], "ms": round(r["ms"], 4), "TFLOPs": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2),
                                    "GBps": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)} for r in rows]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
