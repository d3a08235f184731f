"""Stage-1 kernels alone at the config-2 shape (frames [128,3,224,224] -> pooled [128,112,112,32]) through the C ABI: forward (pack +
statistics pass + finalize + apply/pool pass) and backward, HIP-event timed.  HYB_S1_ABLATE / HYB_S1_WAVE select variants."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_cnn_hybrid_network_for_video_processing_amd import ops
from transformer_cnn_hybrid_network_for_video_processing_amd._lib import lib
dt = ops.dtype_code(sys.argv[1] if len(sys.argv) > 1 else "bf16"); tdt = ops.torch_dtype(dt)
N, H, ci, co = 128, 224, 3, 32
dev = torch.device("cuda", 0)
x = torch.rand(N, ci, H, H, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.2
gamma, beta = torch.ones(co, device=dev), torch.zeros(co, device=dev)
rm, rv, nbt = torch.zeros(co, device=dev), torch.ones(co, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
pooled = torch.empty(N, H // 2, H // 2, co, dtype=tdt, device=dev); dp = torch.randn_like(pooled)
ss, mi = torch.empty(2, co, device=dev), torch.empty(2, co, device=dev)
wsf = torch.empty(lib.query("hyb_convstage_fwd_workspace", dt, 1, 0, co), dtype=torch.uint8, device=dev)
wsb = torch.empty(lib.query("hyb_convstage_bwd_workspace", dt, 1, N, H, H, 0, co), dtype=torch.uint8, device=dev)
dw, dg, db = torch.empty_like(w), torch.empty(co, device=dev), torch.empty(co, device=dev)
st = torch.cuda.current_stream().cuda_stream
def fwd():
    lib.call("hyb_convstage_fwd", dt, 1, x.data_ptr(), w.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(),
             1, 0.1, 1e-5, N, H, H, ci, 0, co, co, None, pooled.data_ptr(), ss.data_ptr(), mi.data_ptr(), None, None, wsf.data_ptr(), wsf.numel(), st)
def bwd():
    lib.call("hyb_convstage_bwd", dt, 1, dp.data_ptr(), x.data_ptr(), None, None, w.data_ptr(), gamma.data_ptr(), ss.data_ptr(), mi.data_ptr(),
             1, N, H, H, ci, 0, co, co, None, dw.data_ptr(), dg.data_ptr(), db.data_ptr(), None, wsb.data_ptr(), wsb.numel(), st)
def timeit(f, reps=20):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print(json.dumps({"ablate": os.environ.get("HYB_S1_ABLATE", "0"), "wave": os.environ.get("HYB_S1_WAVE", "1"), "fwd_us": round(timeit(fwd), 1),
                  "bwd_us": round(timeit(bwd), 1), "checksum": float(pooled.float().sum())}))
