"""time / trace the pooling-epilogue convolution (16 -> 32 at 256^3, N = 2) next to the plain one + separate pooling"""
import os, sys, torch
sys.path.insert(0, '.')
if os.environ.get('KMH_LIB'):
    from keymorph_amd import _lib as _l
    _l.LIBPATH = os.environ['KMH_LIB']
from keymorph_amd import backbone_ops as B
B.set_conv_mode("f16x3")
dev = "cuda"
N, D, Cin, Cout = 2, 256, 16, 32
x = torch.randn(N, D, D, D, Cin, device=dev).abs()
gamma, beta = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.05
with torch.no_grad():
    for pool in (True, False):
        f = (lambda: B.single_conv_gcr(x, gamma, beta, w, 8, dy_premasked=True, pool=True)) if pool else \
            (lambda: B.maxpool2(B.single_conv_gcr(x, gamma, beta, w, 8, dy_premasked=True)))
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            f()
        e1.record()
        torch.cuda.synchronize()
        print(f"pool_in_epilogue={pool}: {e0.elapsed_time(e1) / 5:.3f} ms (incl. the statistics / coefficient launches)")
