"""The 32 -> 16 data gradient behind the 256^3 pooling (N = 2): pooling backward + convolution, three operand formats.
  blocked fp32 scatter -> conv3_fwd_g_kernel<1,true> (round 4) | KEYMORPH_FWD_S=3: conv3_fwd_s_kernel<1,true>
  pre-split records    -> conv3_fwd_s_kernel<1,true,true> (round 5)
usage: prof_split.py [D]   (KMH_G_TRACE=1: cycle stamps on stderr)"""
import os, sys, torch
sys.path.insert(0, '.')
from keymorph_amd import _lib, backbone_ops as B
from keymorph_amd.ops import _p, _stream, check
lib = _lib.load()
B.set_conv_mode("f16x3")
dev = "cuda"
N, D, Cin, Cout = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 32, 16
V = D ** 3
dy = torch.randn(N, D // 2, D // 2, D // 2, Cin, device=dev)
arg = torch.randint(0, 8, dy.shape, dtype=torch.uint8, device=dev)
sc = B.absmax_scale(dy)
w = torch.randn(Cin, Cout, 3, 3, 3, device=dev) * 0.05
pk = B.pack_weight(w, True)
blocked = torch.empty(N, Cin // 8, D, D, D, 8, device=dev)
rec = torch.empty(N, Cin // 8, V + 1, 8, device=dev)
def t(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
pb = lambda: check(lib.kmh_maxpool3d_bwd(None, _p(arg), _p(dy), None, 0, _p(blocked), N, D, D, D, Cin, 1, _stream()), "b")
ps = lambda: check(lib.kmh_maxpool3d_bwd_split(_p(arg), _p(dy), _p(sc), _p(rec), N, D, D, D, Cin, _stream()), "s")
cb = lambda: B.conv3_raw(blocked, None, None, pk, None, N, D, D, D, Cin, Cout, False, False, ascale=sc, in_blocked=True)
cs = lambda: B.conv3_raw(rec, None, None, pk, None, N, D, D, D, Cin, Cout, False, False, ascale=sc, in_blocked=2)
print(f"pool backward: blocked fp32 {t(pb):.3f} ms, pre-split {t(ps):.3f} ms")
yb, ys = cb(), cs()
print("bit-identical:", bool(torch.equal(yb, ys)), " max|y|", float(ys.abs().max()))
print(f"data gradient 32 -> 16 at {N} x {D}^3: blocked fp32 {t(cb):.3f} ms, pre-split {t(cs):.3f} ms")
