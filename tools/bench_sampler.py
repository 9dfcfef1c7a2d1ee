"""Time the trilinear sampler forward (+fused MSE) and grid-gradient at 256^3 (run once with KMH_SAMPLER_OLD=1)."""
import os, sys, math, torch
sys.path.insert(0, '.')
from keymorph_amd import _lib
if os.environ.get("KMH_LIB"):
    _lib.LIBPATH = os.environ["KMH_LIB"]
lib = _lib.load()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(1, 1, S, S, S, device=dev, generator=g)
f = torch.rand(1, 1, S, S, S, device=dev, generator=g)
lin = torch.linspace(-1, 1, S, device=dev)
zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
c, s_ = math.cos(0.2), math.sin(0.2)
grid = torch.stack([1.05 * (c * xx - s_ * yy) + 0.03, 0.95 * (s_ * xx + c * yy) - 0.02, 1.1 * zz + 0.05 * xx], -1)[None].contiguous()
grid = grid + 0.01 * torch.sin(7 * grid.flip(-1))
if os.environ.get("GRID") == "affine3":      # bench.py's stand-alone leg: rotations about all three axes, shear, scale
    from keymorph_amd import synthetic
    from keymorph_amd.transformations import AffineTransform
    grid = AffineTransform(matrix=synthetic.random_affine_matrix(3, dev), dim=3).get_flow_field((1, 1, S, S, S)).contiguous()
out = torch.empty_like(x); loss = torch.empty(1, device=dev); dg = torch.empty_like(grid)
ws = torch.empty(int(lib.kmh_reduce_ws_bytes()), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
QUICK = bool(os.environ.get("KMH_SAMPLER_QUICK"))      # counter passes: a few dispatches per kernel are enough
def timeit(fn, n=30, reps=5):
    if QUICK: n, reps = 3, 1
    for _ in range(1 if QUICK else 3): fn()
    best = 1e9
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best
V = S ** 3
t = timeit(lambda: lib.kmh_warp_mse_fwd(p(x), p(grid), p(f), p(out), p(loss), 1, 1, S, S, S, S, S, S, p(ws), st))
print(f"warp_mse_fwd   {t*1e3:8.1f} us  {V*24/t/1e6:8.1f} GB/s (24 B/voxel)   loss {float(loss):.6f}")
t = timeit(lambda: lib.kmh_grid_sample3d_fwd(p(x), p(grid), p(out), 1, 1, S, S, S, S, S, S, 0, st))
print(f"sample_fwd     {t*1e3:8.1f} us  {V*20/t/1e6:8.1f} GB/s (20 B/voxel)   sum {float(out.double().sum()):.4f}")
t = timeit(lambda: lib.kmh_grid_sample3d_bwd_grid(p(x), p(grid), p(f), p(dg), 1, 1, S, S, S, S, S, S, st))
print(f"sample_bwd_grid{t*1e3:8.1f} us  {V*32/t/1e6:8.1f} GB/s (32 B/voxel)   sum {float(dg.double().abs().sum()):.4f}")
for C in (1, 14):
    xs = torch.rand(1, C, S, S, S, device=dev, generator=g); outs = torch.empty_like(xs)
    t = timeit(lambda: lib.kmh_grid_sample3d_fwd(p(xs), p(grid), p(outs), 1, C, S, S, S, S, S, S, 0, st), n=10)
    print(f"sample_fwd C={C:2d} {t*1e3:8.1f} us  {V*(12+8*C)/t/1e6:8.1f} GB/s ({12+8*C} B/voxel)  [KMH_SAMPLER_XCD={os.environ.get('KMH_SAMPLER_XCD','1')}]")
t = timeit(lambda: lib.kmh_warp_mse_fwd_grad(p(x), p(grid), p(f), p(out), p(loss), p(dg), 1, 1, S, S, S, S, S, S, p(ws), st))
print(f"warp_mse_fwd_grad {t*1e3:8.1f} us  {V*36/t/1e6:8.1f} GB/s (36 B/voxel)")
