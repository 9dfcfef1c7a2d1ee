"""Run one conv layer's forward / dgrad / wgrad a few times (for rocprofv3 --pmc runs).
usage: prof_layer.py D Cin Cout [mode] [nomask] ; KMH_LIB=<path> loads another build of the library (A/B runs);
KMH_TIME=1 also prints the event-timed average of each of the three launches."""
import os, sys, torch
sys.path.insert(0, '.')
if os.environ.get("KMH_LIB"):
    from keymorph_amd import _lib
    _lib.LIBPATH = os.environ["KMH_LIB"]
from keymorph_amd import backbone_ops as B
N, D, Cin, Cout = 2, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "bf16x6"
B.set_conv_mode(mode)
dev = "cuda"
x = torch.randn(N, D, D, D, Cin, device=dev)
w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.05
dy = torch.randn(N, D, D, D, Cout, device=dev)
y = torch.randn(N, D, D, D, Cout, device=dev)
sc = torch.ones(N, Cin, device=dev); sh = torch.zeros(N, Cin, device=dev)
asc = B.absmax_scale(x) if B._needs_range_scales() else None      # sc = 1, sh = 0: the normalised tensor is x itself
dsc = B.absmax_scale(dy) if B._needs_range_scales() else None
masked = not (len(sys.argv) > 5 and sys.argv[5] == "nomask")
blocked = bool(os.environ.get("KMH_BLOCKED"))      # data-gradient input in the channel-blocked layout (N, C/8, D, H, W, 8)
if blocked:
    dyb = dy.view(N, D, D, D, Cout // 8, 8).permute(0, 4, 1, 2, 3, 5).contiguous()
    ref = B.conv3_raw(dy, None, None, B.pack_weight(w, True), None, N, D, D, D, Cout, Cin, False, False)
    got = B.conv3_raw(dyb, None, None, B.pack_weight(w, True), None, N, D, D, D, Cout, Cin, False, False, in_blocked=True)
    print("blocked vs NDHWC data gradient: max |diff| =", float((got - ref).abs().max()))
    rw = B.conv3_wgrad(x, sc, sh, dy, N, D, D, D, Cin, Cout, False, xscale=asc)
    gw = B.conv3_wgrad(x, sc, sh, dyb, N, D, D, D, Cin, Cout, False, xscale=asc, dz_blocked=True)
    print("blocked vs NDHWC weight gradient: max |diff| =", float((gw - rw).abs().max()))
wf, wt = B.pack_weight(w, False), B.pack_weight(w, True)
xblocked = bool(os.environ.get("KMH_XBLOCKED"))    # FORWARD input in the channel-blocked layout (N, C/8, D, H, W, 8): what would it buy?
xb = x.view(N, D, D, D, Cin // 8, 8).permute(0, 4, 1, 2, 3, 5).contiguous() if xblocked else None
if xblocked:
    r0 = B.conv3_raw(x, sc, sh, wf, None, N, D, D, D, Cin, Cout, False, True, ascale=asc)
    r1 = B.conv3_raw(xb, sc, sh, wf, None, N, D, D, D, Cin, Cout, False, True, ascale=asc, in_blocked=True)
    print("blocked vs NDHWC forward: max |diff| =", float((r1 - r0).abs().max()))
steps = [
    ("fwd", lambda: B.conv3_raw(xb if xblocked else x, sc, sh, wf, None, N, D, D, D, Cin, Cout, False, True, ascale=asc, in_blocked=xblocked)),
    ("dgrad", lambda: B.conv3_raw(dyb if blocked else dy, None, None, wt, None, N, D, D, D, Cout, Cin, False, False, mask=y if masked else None, in_blocked=blocked)),
    ("wgrad", lambda: B.conv3_wgrad(x, sc, sh, dyb if blocked else dy, N, D, D, D, Cin, Cout, False, dzmask=y if masked else None, xscale=asc, dscale=dsc, dz_blocked=blocked)),
]
for it in range(3):
    for _, f in steps:
        f()
torch.cuda.synchronize()
if os.environ.get("KMH_TIME"):
    for name, f in steps:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / 10:.3f} ms")
print("done")
