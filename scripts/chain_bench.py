"""Per-launch cost of the token-side entry points in a DEPENDENT chain of a replayed hipGraph (200 nodes, ping-pong buffers), i.e. what one
more small launch costs inside the encoder when nothing else is in the way: hyb_linear_fwd (M=128, 512->512 and 512->2048, bf16: convert +
split-K GEMM = 2 launches per call), hyb_ln_residual_fwd, hyb_attention_fwd.  Compare with the ~5-7 us these kernels show inside the real
step (profiles/r02_b_kernel_stats_c2.csv) and the 2.0 us of a minimal kernel (profiles/r02_switch_microbench.json)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_cnn_hybrid_network_for_video_processing_amd._lib import lib, HYB_BF16

dev = "cuda"
M, D, HID, H = 128, 512, 2048, 8
bf = torch.bfloat16
st = torch.cuda.Stream()
res = {}


def chain(name, fn, nodes=200, launches_per_call=1):
    with torch.cuda.stream(st):
        fn(0); fn(1)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for i in range(nodes):
                fn(i)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
    res[name] = {"us_per_call": round(best * 1e3 / nodes, 3), "launches_per_call": launches_per_call}


x = [torch.randn(M, D, device=dev).to(bf) * 0.1 for _ in range(2)]
hbuf = [torch.zeros(M, HID, device=dev, dtype=bf) for _ in range(2)]
W1 = torch.randn(D, D, device=dev) * 0.03
W2 = torch.randn(HID, D, device=dev) * 0.03
b1 = torch.zeros(D, device=dev); b2 = torch.zeros(HID, device=dev)
s_ = lambda: torch.cuda.current_stream().cuda_stream
chain("linear_fwd 128x512->512 bf16", lambda i: lib.call("hyb_linear_fwd", HYB_BF16, x[i & 1].data_ptr(), D, W1.data_ptr(), b1.data_ptr(), x[(i + 1) & 1].data_ptr(), M, D, D, 0, s_()), launches_per_call=2)
chain("linear_fwd 128x512->2048 bf16", lambda i: lib.call("hyb_linear_fwd", HYB_BF16, x[0].data_ptr(), D, W2.data_ptr(), b2.data_ptr(), hbuf[i & 1].data_ptr(), M, HID, D, 1, s_()), launches_per_call=2)
gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
stats = torch.zeros(M, 2, device=dev)
chain("ln_residual_fwd 128x512 bf16", lambda i: lib.call("hyb_ln_residual_fwd", HYB_BF16, x[i & 1].data_ptr(), x[i & 1].data_ptr(), gamma.data_ptr(), beta.data_ptr(), x[(i + 1) & 1].data_ptr(), stats.data_ptr(), M, D, 1e-5, 0.70710678, 0.0, 0, s_()))
qkv = [torch.randn(M, D, device=dev).to(bf) * 0.1 for _ in range(3)]
ast = torch.zeros(8 * H * 16 * 2, device=dev)
chain("attention_fwd B=8 S=16 D=512 H=8 bf16", lambda i: lib.call("hyb_attention_fwd", HYB_BF16, x[i & 1].data_ptr(), qkv[1].data_ptr(), qkv[2].data_ptr(), None, x[(i + 1) & 1].data_ptr(), ast.data_ptr(), 8, 16, D, H, 0.0, 0, s_()))
print(json.dumps(res))
