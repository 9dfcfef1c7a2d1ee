"""Times the fused head (final 1x1x1 conv + ReLU + center of mass) forward and backward at one shape.
usage: python tools/prof_head.py [D=256] [Cin=16] [K=512] [N=2]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("KMH_LIB"):      # A/B runs against another build of the library (tools/build_old_lib.sh)
    from keymorph_amd import _lib
    _lib.LIBPATH = os.environ["KMH_LIB"]
from keymorph_amd import backbone_ops as bo

D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Cin = int(sys.argv[2]) if len(sys.argv) > 2 else 16
K = int(sys.argv[3]) if len(sys.argv) > 3 else 512
N = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
feat = torch.randn((N, D, D, D, Cin), generator=g).relu_().to(dev).requires_grad_(True)
w = (torch.randn((K, Cin), generator=g) * 0.2).to(dev).requires_grad_(True)
b = (torch.randn((K,), generator=g) * 0.1).to(dev).requires_grad_(True)
dpts = torch.randn((N, K, 3), generator=g).to(dev)


def run():
    pts, power = bo._HeadCoM.apply(feat, w, b, True)
    return pts


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    tf = timed(run)
pts = run()


def bwd():
    torch.autograd.grad(pts, (feat, w, b), dpts, retain_graph=True)


tb = timed(bwd)
print(f"head D={D} Cin={Cin} K={K} N={N} mode={bo.CONV_MODE}: fwd {tf:.3f} ms  bwd {tb:.3f} ms")
