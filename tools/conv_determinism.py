"""Run the 27-tap forward kernel repeatedly on the same inputs and compare outputs and epilogue statistics bit for bit
(usage: python tools/conv_determinism.py D Cin Cout [N] [reps])"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keymorph_amd import backbone_ops as B
D, Cin, Cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 2
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 12
B.set_conv_mode("f16x3")
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(N, D, D, D, Cin, device="cuda", generator=g)
w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda", generator=g) * 0.05
sc = 1 + 0.2 * torch.randn(N, Cin, device="cuda", generator=g); sh = 0.2 * torch.randn(N, Cin, device="cuda", generator=g)
asc = B.absmax_scale(x * 2)
pk = B.pack_weight(w, False)
mode = os.environ.get("MODE", "fwd")     # fwd | dgrad (premasked gradient, no norm) | dgrad_blocked | addend
if mode.startswith("dgrad"):
    pk = B.pack_weight(w.permute(1, 0, 2, 3, 4).contiguous(), True)      # data gradient Cin -> Cout of the transposed filter
    sc = sh = None
    asc = B.absmax_scale(x)
xin = x.view(N, D, D, D, Cin // 8, 8).permute(0, 4, 1, 2, 3, 5).contiguous() if mode == "dgrad_blocked" else x
add = torch.randn(N, D, D, D, Cout, device="cuda", generator=g) if mode == "addend" else None
ref = None
bad = 0
for it in range(reps):
    st = torch.full((N, Cout, 2), float("nan"), dtype=torch.float64, device="cuda")
    y = B.conv3_raw(xin, sc, sh, pk, None, N, D, D, D, Cin, Cout, False, not mode.startswith('dgrad'), ascale=asc, stats_out=st,
                    in_blocked=mode == 'dgrad_blocked', addend=add)
    torch.cuda.synchronize()
    if ref is None:
        ref = (y.clone(), st.clone())
    else:
        ey, es = torch.equal(y, ref[0]), torch.equal(st, ref[1])
        if not (ey and es):
            bad += 1
            d = (y != ref[0])
            idx = d.nonzero()
            lo, hi = idx.min(0).values.tolist(), idx.max(0).values.tolist()
            print(f"rep {it}: outputs equal {ey} ({int(d.sum())} elements differ, max |diff| {float((y - ref[0]).abs().max()):.3e}; "
                  f"(n,z,y,x,c) from {lo} to {hi}), stats equal {es}")
print(f"{D}^3 {Cin}->{Cout} N={N}: {bad} of {reps - 1} repetitions differ from the first")
