"""time the up2 kernels at the bench shapes (N = 4): decoder levels 128^3 (64 + 128 -> 64) and 64^3 (128 + 256 -> 128)"""
import os, sys, torch
sys.path.insert(0, '.')
if os.environ.get('KMH_LIB'):
    from keymorph_amd import _lib as _l
    _l.LIBPATH = os.environ['KMH_LIB']
from keymorph_amd import backbone_ops as B
B.set_conv_mode("f16x3")
dev = "cuda"
for (D, Cs, Cl, Cout) in ((128, 64, 128, 64), (64, 128, 256, 128)):
    N = 4
    dz = torch.randn(N, D, D, D, Cout, device=dev)
    w = torch.randn(Cout, Cs + Cl, 3, 3, 3, device=dev) * 0.02
    dsc = B.absmax_scale(dz)
    for _ in range(2): B.conv3_up2_dgrad(dz, w, Cs, Cl, dsc)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): B.conv3_up2_dgrad(dz, w, Cs, Cl, dsc)
    e1.record(); torch.cuda.synchronize()
    print(f"up2_dgrad D={D} Cl={Cl} Cout={Cout}: {e0.elapsed_time(e1)/5:.3f} ms")

from keymorph_amd import _lib
lib = _lib.load()
for (D, Cs, Cl, Cout) in ((128, 64, 128, 64), (64, 128, 256, 128)):
    N = 4
    Vl = (D // 2) ** 3
    A = torch.randn(N, Vl, Cl, device=dev); Bx = torch.randn(N, Vl, 27 * Cout, device=dev)
    C = torch.empty(N, Cl, 27 * Cout, device=dev)
    ws = torch.empty(int(lib.kmh_up2_wgrad_gemm_ws_bytes(N, Vl, Cl, 27 * Cout)), dtype=torch.uint8, device=dev)
    sa, sb = B.absmax_scale(A), B.absmax_scale(Bx)
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: lib.kmh_up2_wgrad_gemm(A.data_ptr(), Bx.data_ptr(), C.data_ptr(), N, Vl, Cl, 27 * Cout, 2, sa.data_ptr(), sb.data_ptr(), None, None, ws.data_ptr(), st)
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    ref = torch.bmm(A[:1].transpose(1, 2).double(), Bx[:1].double())
    err = float((C[:1].double() - ref).abs().max() / ref.abs().max())
    print(f"up2_wgrad_gemm Vl={Vl} Cl={Cl} J={27*Cout}: {ms:.3f} ms  {2.0*N*Vl*Cl*27*Cout/ms/1e9:.0f} TF-eq  rel err {err:.1e}")
