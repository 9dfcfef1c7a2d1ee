"""A handful of launches of the bilinear warp kernels at 256^3 (for rocprofv3 --pmc passes: few dispatches, no side legs).
usage: python tools/prof_sampler_min.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keymorph_amd import _lib, synthetic
from keymorph_amd.transformations import AffineTransform
lib = _lib.load()
S, dev = 256, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(1, 1, S, S, S, device=dev, generator=g)
f = torch.rand(1, 1, S, S, S, device=dev, generator=g)
grid = AffineTransform(matrix=synthetic.random_affine_matrix(3, dev), dim=3).get_flow_field((1, 1, S, S, S)).contiguous()
out, loss, dg = torch.empty_like(x), torch.empty(1, device=dev), torch.empty_like(grid)
ws = torch.empty(int(lib.kmh_reduce_ws_bytes()), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
for _ in range(3):
    lib.kmh_grid_sample3d_fwd(p(x), p(grid), p(out), 1, 1, S, S, S, S, S, S, 0, st)
    lib.kmh_warp_mse_fwd_grad(p(x), p(grid), p(f), p(out), p(loss), p(dg), 1, 1, S, S, S, S, S, S, p(ws), st)
torch.cuda.synchronize()
print("done", float(loss))
