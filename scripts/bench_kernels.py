"""Time (or profile) the conv kernels of one config-2 step standalone through the C ABI."""
import argparse, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from transformer_cnn_hybrid_network_for_video_processing_amd import ops
ap = argparse.ArgumentParser(); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--frames", type=int, default=16); ap.add_argument("--size", type=int, default=224)
a = ap.parse_args()
dt = ops.dtype_code(a.dtype)
rows = bench.conv_kernel_table(a, dt, ops.torch_dtype(dt), torch.device("cuda", 0))
for r in rows:
    print("%-50s %8.4f ms  %8.1f TF/s  %8.1f GB/s" % (r["kernel"], r["ms"], r["flops"] / r["ms"] / 1e9, r["bytes"] / r["ms"] / 1e6))
print(bench.dominant_kernel(rows))
