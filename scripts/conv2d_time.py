"""Forward / backward time of hybrid::conv2d on every convolution geometry of Encoder_32K at 16 frames of 256 x 256, beside the
shape's HBM and fp32-matrix floors (6 TB/s, 157 TF/s).  usage: python scripts/conv2d_time.py [frames]"""
import sys, json
import torch
import transformer_cnn_hybrid_network_for_video_processing_amd  # noqa: F401
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
CASES = [("stem 7x7 s2 3->64", 7, 2, 3, 3, 64, 256), ("1x1 64->64", 1, 1, 0, 64, 64, 128), ("3x3 64->64", 3, 1, 1, 64, 64, 128), ("1x1 64->256", 1, 1, 0, 64, 256, 128),
         ("1x1 256->64", 1, 1, 0, 256, 64, 128), ("1x1 256->128", 1, 1, 0, 256, 128, 128), ("3x3 s2 128->128", 3, 2, 1, 128, 128, 128),
         ("1x1 128->512", 1, 1, 0, 128, 512, 64), ("1x1 s2 256->512", 1, 2, 0, 256, 512, 128), ("1x1 512->128", 1, 1, 0, 512, 128, 64),
         ("3x3 128->128", 3, 1, 1, 128, 128, 64), ("3x3 512->128", 3, 1, 1, 512, 128, 64), ("3x3 128->64", 3, 1, 1, 128, 64, 64)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timeit(fn, reps=5):
    fn(); fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
rows = []
for name, k, s, p, ci, co, hw in CASES:
    x = torch.randn(N, hw, hw, ci, device="cuda").requires_grad_(ci != 3)
    w = (torch.randn(co, ci, k, k, device="cuda") * 0.05).requires_grad_()
    ho = (hw + 2 * p - k) // s + 1
    fwd = timeit(lambda: torch.ops.hybrid.conv2d(x.detach(), w.detach(), None, s, p, 1, 0))
    dy = torch.randn(N, ho, ho, co, device="cuda")
    e = torch.empty(0, device="cuda")
    bwd = timeit(lambda: torch.ops.hybrid.conv2d_bwd(dy, x.detach(), w.detach(), e, False, ci != 3, s, p, 1, 0))
    gf = 2.0 * N * ho * ho * co * ci * k * k / 1e9
    mb = 4.0 * N * (hw * hw * ci + ho * ho * co) / 1e6
    rows.append({"conv": name, "fwd_us": round(fwd, 1), "bwd_us": round(bwd, 1), "gflop": round(gf, 2), "fwd_tflops": round(gf / fwd * 1e3, 1),
                 "hbm_floor_us": round(mb / 6e6 * 1e6, 1), "mfma_floor_us": round(gf / 157e3 * 1e6, 1)})
    print(json.dumps(rows[-1]), flush=True)
