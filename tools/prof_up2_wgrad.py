"""Time the upsampled channels' weight gradient both ways at the two decoder levels of the bench step (N = 4):
box sums + matrix product (kmh_up2_boxsum + kmh_up2_wgrad_gemm) against the product that forms them itself (kmh_up2_wgrad_fold)."""
import sys, torch
sys.path.insert(0, '.')
from keymorph_amd import _lib
from keymorph_amd import backbone_ops as B
from keymorph_amd.backbone_ops import _p, _stream, check
lib = _lib.load()
B.set_conv_mode("f16x3")
dev = "cuda"
for (N, ld, Cl, Cout) in ((4, (64, 64, 64), 128, 64), (4, (32, 32, 32), 256, 128)):
    Vl = ld[0] * ld[1] * ld[2]
    dz = torch.randn(N, 2 * ld[0], 2 * ld[1], 2 * ld[2], Cout, device=dev)
    xl = torch.randn(N, *ld, Cl, device=dev)
    sc = torch.ones(N, Cl, device=dev); sh = torch.zeros(N, Cl, device=dev)
    asc, dsc = B.absmax_scale(xl), B.absmax_scale(dz)
    bsc = dsc * torch.tensor([0.125, 8.0], device=dev)
    boxes = torch.empty(N, Vl, 27 * Cout, device=dev)
    old = torch.empty(N, Cl, 27 * Cout, device=dev); new = torch.empty_like(old)
    ws = torch.empty(int(lib.kmh_up2_wgrad_gemm_ws_bytes(N, Vl, Cl, 27 * Cout)) // 4 + 1, device=dev)
    ws2 = torch.empty(int(lib.kmh_up2_wgrad_fold_ws_bytes(N, ld[0], ld[1], ld[2], Cl, Cout)) // 4 + 1, device=dev)
    def a():
        check(lib.kmh_up2_boxsum(_p(dz), _p(boxes), N, ld[0], ld[1], ld[2], Cout, 0, _stream()), "boxsum")
        check(lib.kmh_up2_wgrad_gemm(_p(xl), _p(boxes), _p(old), N, Vl, Cl, 27 * Cout, 2, _p(asc), _p(bsc), _p(sc), _p(sh), _p(ws), _stream()), "gemm")
    dzb = dz.reshape(N, -1, Cout // 8, 8).permute(0, 2, 1, 3).contiguous()
    def b():
        check(lib.kmh_up2_wgrad_fold(_p(xl), _p(dz), _p(new), N, ld[0], ld[1], ld[2], Cl, Cout, 2, _p(asc), _p(dsc), _p(sc), _p(sh), 0, _p(ws2), _stream()), "fold")
    def c():
        check(lib.kmh_up2_wgrad_fold(_p(xl), _p(dzb), _p(new), N, ld[0], ld[1], ld[2], Cl, Cout, 2, _p(asc), _p(dsc), _p(sc), _p(sh), 1, _p(ws2), _stream()), "fold")
    def d():
        check(lib.kmh_up2_boxsum(_p(dzb), _p(boxes), N, ld[0], ld[1], ld[2], Cout, 1, _stream()), "boxsum")
        check(lib.kmh_up2_wgrad_gemm(_p(xl), _p(boxes), _p(old), N, Vl, Cl, 27 * Cout, 2, _p(asc), _p(bsc), _p(sc), _p(sh), _p(ws), _stream()), "gemm")
    for name, f in (("boxsum + gemm", a), ("boxsum + gemm, dz channel-blocked (the step's case)", d), ("fold", b), ("fold, dz channel-blocked (the step's case)", c)):
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        print(f"low {ld} Cl {Cl} Cout {Cout}: {name}: {e0.elapsed_time(e1) / 5:.3f} ms")
    print("  max |diff| / max:", float((new - old).abs().max() / old.abs().max()))
