"""time the fused keypoint head (64 -> 512 at 4 x 128^3: the headline step's call) forward and backward, with the
stored-sign-mask backward and with the recomputing one; KMH_TRACE=1 wraps nothing -- run under rocprofv3 for kernels"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
if __import__("os").environ.get("KMH_LIB"):
    from keymorph_amd import _lib as _l
    _l.LIBPATH = __import__("os").environ["KMH_LIB"]
from keymorph_amd import backbone_ops as B
B.set_conv_mode("f16x3")
dev = "cuda"
N, D, Cin, Cout = 4, 128, 64, 512
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn(N, D, D, D, Cin, device=dev, generator=g).abs()
w = (torch.randn(Cout, Cin, 1, 1, 1, device=dev, generator=g) / 8).requires_grad_(True)
b = (0.3 * torch.randn(Cout, device=dev, generator=g)).requires_grad_(True)
cot = torch.randn(N, Cout, 3, device=dev, generator=g)


def ev():
    return torch.cuda.Event(enable_timing=True)


for use_mask in (True, False, True, False):
    B.HEAD_MASK = use_mask
    xs = x.clone().requires_grad_(True)
    tf = tb = 0.0
    for it in range(7):
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        p = B.head_com(xs, w, b, feat_from_relu=True)
        e1.record()
        (p * cot).sum().backward()
        e2.record()
        torch.cuda.synchronize()
        if it >= 2:
            tf += e0.elapsed_time(e1) / 5
            tb += e1.elapsed_time(e2) / 5
    print(f"mask={use_mask}: forward {tf:.3f} ms, backward {tb:.3f} ms (incl. the small launches around the kernels)")
