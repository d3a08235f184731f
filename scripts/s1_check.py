"""Debug: stage-1 forward at full size against torch ops on the GPU (fp32), for HYB_S1_WAVE=0/1."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_cnn_hybrid_network_for_video_processing_amd import ops
from transformer_cnn_hybrid_network_for_video_processing_amd._lib import lib
dt = ops.dtype_code(sys.argv[1] if len(sys.argv) > 1 else "bf16"); tdt = ops.torch_dtype(dt)
N, H, W, ci, co = int(os.environ.get("N", 128)), int(os.environ.get("H", 224)), int(os.environ.get("W", 224)), 3, 32
dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = torch.rand(N, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.2
gamma, beta = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev) * 0.1
rm, rv, nbt = torch.zeros(co, device=dev), torch.ones(co, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
pooled = torch.empty(N, H // 2, W // 2, co, dtype=tdt, device=dev)
ss, mi = torch.empty(2, co, device=dev), torch.empty(2, co, device=dev)
wsf = torch.empty(lib.query("hyb_convstage_fwd_workspace", dt, 1, 0, co), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
lib.call("hyb_convstage_fwd", dt, 1, x.data_ptr(), w.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(),
         1, 0.1, 1e-5, N, H, W, ci, 0, co, co, None, pooled.data_ptr(), ss.data_ptr(), mi.data_ptr(), None, None, wsf.data_ptr(), wsf.numel(), st)
torch.cuda.synchronize()
xw, ww = (x.bfloat16().float(), w.bfloat16().float()) if dt == 1 else (x, w)
y = F.conv2d(xw, ww, padding=1)
mean = y.mean((0, 2, 3)); var = y.var((0, 2, 3), unbiased=False)
ref = F.max_pool2d(F.relu(F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5)), 2).permute(0, 2, 3, 1)
got = pooled.float()
err = (got - ref).abs()
print("wave", os.environ.get("HYB_S1_WAVE", "1"), "mean err", (mi[0] - mean).abs().max().item(), "invstd err", (mi[1] - (var + 1e-5).rsqrt()).abs().max().item(),
      "pooled max err", err.max().item(), "bad elems", int((err > 0.05).sum()), "of", err.numel())
if (err > 0.05).any():
    idx = (err > 0.05).nonzero()
    print("first bad", idx[:5].tolist(), "last bad", idx[-5:].tolist())
    bad_n = idx[:, 0].unique(); print("bad images", bad_n[:20].tolist(), len(bad_n))
    print("bad rows (oy)", idx[:, 1].unique()[:40].tolist()); print("bad cols (ox)", idx[:, 2].unique()[:40].tolist())
