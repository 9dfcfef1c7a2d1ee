"""Determinism stress of kmh_tps_fit_fwd: the same systems solved repeatedly must give bit-identical theta.  Interleaves
other work (a conv-sized allocation + fill) so that the workspace and caches hold different residue between solves."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keymorph_amd import ops
dev = "cuda"
for T, lam in ((200, 1.0), (512, 0.0), (64, 0.1), (700, 1.0)):
    g = torch.Generator().manual_seed(12 + T)
    ctrl = (torch.rand(2, T, 3, generator=g) * 1.6 - 0.8).to(dev)
    tgt = ctrl + 0.05 * torch.randn(2, T, 3, generator=g).to(dev)
    lm = torch.full((2,), lam, device=dev)
    ref = ops.tps_fit(ctrl, tgt, lm).clone()
    bad = 0
    worst = 0.0
    for it in range(300):
        if it % 3 == 0:
            junk = torch.full((1 << 22,), float("nan"), device=dev); del junk       # poison freed memory
        out = ops.tps_fit(ctrl, tgt, lm)
        if not torch.equal(out, ref):
            bad += 1
            worst = max(worst, float((out - ref).abs().max() / ref.abs().max()))
    print(f"T={T} lambda={lam}: {bad}/300 solves differ from the first (worst relative difference {worst:.2e})")
# batches: the cluster kernel's workgroup count follows N (G = 8 while N * G <= 128, then 4, 2, 1)
for N in (1, 8, 16, 40, 130):
    T = 128
    g = torch.Generator().manual_seed(N)
    ctrl = (torch.rand(N, T, 3, generator=g) * 1.6 - 0.8).to(dev)
    tgt = ctrl + 0.05 * torch.randn(N, T, 3, generator=g).to(dev)
    lm = torch.full((N,), 0.5, device=dev)
    ref = ops.tps_fit(ctrl, tgt, lm).clone()
    one = torch.cat([ops.tps_fit(ctrl[i:i + 1], tgt[i:i + 1], lm[i:i + 1]) for i in range(N)])
    bad = sum(int(not torch.equal(ops.tps_fit(ctrl, tgt, lm), ref)) for _ in range(50))
    print(f"N={N}: batch == one by one: {bool(torch.equal(ref, one))}; {bad}/50 repeats differ")
