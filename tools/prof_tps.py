"""time the TPS grid evaluation (2 x 256^3 voxels x 512 keypoints: the headline step's launches), forward and backward"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("KMH_LIB"):
    from keymorph_amd import _lib as _l
    _l.LIBPATH = os.environ["KMH_LIB"]
from keymorph_amd import ops
dev = "cuda"
N, T, D = 2, 512, 256
g = torch.Generator(device=dev).manual_seed(5)
ctrl = (torch.rand(N, T, 3, device=dev, generator=g) * 1.6 - 0.8)
theta = torch.randn(N, T + 4, 3, device=dev, generator=g) * 0.01
theta.requires_grad_(True); ctrl.requires_grad_(True)
cot = torch.randn(N, D, D, D, 3, device=dev, generator=g)
def ev(): return torch.cuda.Event(enable_timing=True)
tf = tb = 0.0
for it in range(6):
    e0, e1, e2 = ev(), ev(), ev()
    e0.record()
    grid = ops.tps_grid(theta, ctrl, (D, D, D))
    e1.record()
    grid.backward(cot)
    e2.record(); torch.cuda.synchronize()
    if it >= 2:
        tf += e0.elapsed_time(e1) / 4; tb += e1.elapsed_time(e2) / 4
print(f"tps grid forward {tf:.3f} ms, backward {tb:.3f} ms; dtheta checksum {float(theta.grad.double().abs().sum()):.9e} dctrl {float(ctrl.grad.double().abs().sum()):.9e}")
