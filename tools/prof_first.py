"""time the first layer's kernels (1 -> 16 channels at 4 x 256^3): forward (first_fwd_kernel) and weight gradient"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keymorph_amd import _lib, backbone_ops as B
from keymorph_amd.backbone_ops import _p, _stream, _f32, workspace, check
lib = _lib.load()
dev = "cuda"
N, D, Cout = 4, 256, 16
x = torch.rand(N, D, D, D, device=dev)
w = torch.randn(Cout, 1, 3, 3, 3, device=dev) * 0.2
sc, sh = torch.ones(N, device=dev), torch.zeros(N, device=dev)
y = torch.empty(N, D, D, D, Cout, device=dev)
st = torch.zeros(N, Cout, 2, dtype=torch.float64, device=dev)
ws = workspace(int(lib.kmh_conv3d_first_layer_fwd_ws_bytes(N, D, D, D, Cout)), x.device, "first")
f = lambda: check(lib.kmh_conv3d_first_layer_fwd(_p(x), _p(sc), _p(sh), _p(w), _p(y), N, D, D, D, Cout, _p(ws), _p(st), _stream()), "fwd")
for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    f()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"first layer forward: {ms:.3f} ms  ({y.numel() * 4 / ms / 1e6:.0f} GB/s of output)")
