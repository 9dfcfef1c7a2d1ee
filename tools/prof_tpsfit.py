"""time the TPS fit (two 516 x 516 systems: the headline step's launch), forward (assemble + LU + solve) and backward"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("KMH_LIB"):
    from keymorph_amd import _lib as _l
    _l.LIBPATH = os.environ["KMH_LIB"]
from keymorph_amd import ops
dev = "cuda"
N, T = 2, 512
g = torch.Generator(device=dev).manual_seed(5)
ctrl = (torch.rand(N, T, 3, device=dev, generator=g) * 1.6 - 0.8).requires_grad_(True)
tgt = (ctrl.detach() + 0.05 * torch.randn(N, T, 3, device=dev, generator=g)).requires_grad_(True)
lm = torch.zeros(N, device=dev)
cot = torch.randn(N, T + 4, 3, device=dev, generator=g)
def ev(): return torch.cuda.Event(enable_timing=True)
tf = tb = 0.0
for it in range(12):
    e0, e1, e2 = ev(), ev(), ev()
    e0.record()
    th = ops.tps_fit(ctrl, tgt, lm)
    e1.record()
    th.backward(cot)
    e2.record(); torch.cuda.synchronize()
    if it >= 2:
        tf += e0.elapsed_time(e1) / 10; tb += e1.elapsed_time(e2) / 10
print(f"tps fit forward {tf:.3f} ms, backward {tb:.3f} ms; theta checksum {float(th.double().abs().sum()):.12e}")
