"""time kmh_up2_boxsum at the two decoder levels of the headline step and check the LDS-tiled kernel against the plain one
(KEYMORPH_BOXSUM_PLAIN=1 selects the plain kernel for the whole process)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keymorph_amd import _lib
from keymorph_amd.backbone_ops import _p, _stream, check
lib = _lib.load()
dev = "cuda"
torch.manual_seed(7)
for (Dl, Cout) in ((64, 64), (32, 128), (5, 24)):
    N = 4 if Dl > 8 else 2
    shp = (N, 2 * Dl, 2 * Dl + (2 if Dl == 5 else 0), 2 * Dl + (6 if Dl == 5 else 0), Cout)
    dz = torch.randn(*shp, device=dev)
    Dl_, Hl_, Wl_ = shp[1] // 2, shp[2] // 2, shp[3] // 2
    G = torch.full((N, Dl_ * Hl_ * Wl_, 27, Cout), float("nan"), device=dev)
    f = lambda: check(lib.kmh_up2_boxsum(_p(dz), _p(G), N, Dl_, Hl_, Wl_, Cout, 0, _stream()), "boxsum")
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    print(f"boxsum Dl={Dl_}x{Hl_}x{Wl_} Cout={Cout}: {e0.elapsed_time(e1) / 5:.3f} ms  checksum {float(G.double().sum()):.10e} "
          f"abs {float(G.double().abs().sum()):.10e} finite {bool(torch.isfinite(G).all())}")
