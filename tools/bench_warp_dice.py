"""Time the fused warp + Dice kernels (kmh_warp_dice_sums / kmh_warp_dice_bwd_grid) against the unfused route at
N x C x S^3 (default 2 x 14 x 256^3, the bench's Dice leg), with the A/B switches KMH_WD_ILP_A / KMH_WD_ILP_B /
KMH_WD_BLOCKS read at the first call of each launcher (so: one process per setting)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keymorph_amd import _lib, loss_ops, utils
lib = _lib.load()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
C = int(sys.argv[2]) if len(sys.argv) > 2 else 14
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
lab = lambda: torch.randint(0, C, (N, 1, S, S, S), device=dev, generator=g)
x = torch.zeros(N, C, S, S, S, device=dev).scatter_(1, lab(), 1.0)
f = torch.zeros(N, C, S, S, S, device=dev).scatter_(1, lab(), 1.0)
lin = torch.linspace(-1, 1, S, device=dev)
zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
c, s_ = math.cos(0.2), math.sin(0.2)
grid = torch.stack([1.05 * (c * xx - s_ * yy) + 0.03, 0.95 * (s_ * xx + c * yy) - 0.02, 1.1 * zz + 0.05 * xx], -1)[None]
grid = (grid + 0.01 * torch.sin(7 * grid.flip(-1))).repeat(N, 1, 1, 1, 1).contiguous()
del zz, yy, xx
sums = torch.empty(N * C, 3, device=dev); dg = torch.empty_like(grid)
ca = torch.randn(N * C, device=dev); cb = torch.randn(N * C, device=dev)
ws = torch.empty(int(lib.kmh_reduce_ws_bytes()), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()


def timeit(fn, n=10, reps=3):
    for _ in range(2): fn()
    best = 1e9
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best


V = N * S ** 3
tag = " ".join(f"{k}={os.environ[k]}" for k in ("KMH_WD_ILP_A", "KMH_WD_ILP_B", "KMH_WD_BLOCKS") if k in os.environ)
labx = torch.empty(N, S ** 3, dtype=torch.uint8, device=dev); labf = torch.empty_like(labx)
gate = torch.ones(1, dtype=torch.int32, device=dev)
t = timeit(lambda: (lib.kmh_onehot_to_labels(p(x), N, C, S ** 3, p(labx), p(gate), st),
                    lib.kmh_onehot_to_labels(p(f), N, C, S ** 3, p(labf), p(gate), st)))
print(f"[{tag}] onehot_to_labels x 2   {t:7.3f} ms  {2*V*4*C/t/1e9:7.2f} TB/s ({8*C} B/voxel read)  gate {int(gate)}")
for name, lx, lf, gt in (("labels", labx, labf, gate), ("dense ", None, None, None)):
    pl = (lambda t_: 0 if t_ is None else t_.data_ptr())
    t = timeit(lambda: lib.kmh_warp_dice_sums(p(x), p(grid), p(f), p(sums), N, C, S, S, S, S, S, S, pl(lx), pl(lf), pl(gt), p(ws), st))
    print(f"[{tag}] {name} warp_dice_sums     {t:7.3f} ms  {V*(12+8*C)/t/1e9:7.2f} TB/s of the dense {12+8*C} B/voxel  checksum {float(sums.double().sum()):.3f}")
    t = timeit(lambda: lib.kmh_warp_dice_bwd_grid(p(x), p(grid), p(f), p(ca), p(cb), p(dg), N, C, S, S, S, S, S, S, pl(lx), pl(lf), pl(gt), st))
    print(f"[{tag}] {name} warp_dice_bwd_grid {t:7.3f} ms  {V*(24+8*C)/t/1e9:7.2f} TB/s of the dense {24+8*C} B/voxel  checksum {float(dg.double().abs().sum()):.3f}")
if not tag:
    out = torch.empty_like(x)
    t = timeit(lambda: lib.kmh_grid_sample3d_fwd(p(x), p(grid), p(out), N, C, S, S, S, S, S, S, 0, st))
    print(f"unfused: sample_fwd C={C} {t:7.3f} ms  {V*(12+8*C)/t/1e9:7.2f} TB/s")
    t = timeit(lambda: lib.kmh_dice_sums(p(out), p(f), N * C, S ** 3, p(sums), p(ws), st))
    print(f"unfused: dice_sums      {t:7.3f} ms  {V*8*C/t/1e9:7.2f} TB/s")
    t = timeit(lambda: lib.kmh_grid_sample3d_bwd_grid(p(x), p(grid), p(out), p(dg), N, C, S, S, S, S, S, S, st))
    print(f"unfused: bwd_grid       {t:7.3f} ms  {V*(24+8*C)/t/1e9:7.2f} TB/s")
    src = torch.empty(V * (12 + 8 * C) // 8, device=dev); dst = torch.empty_like(src)
    t = timeit(lambda: dst.copy_(src))
    print(f"copy_ of the same bytes {t:7.3f} ms  {V*(12+8*C)/t/1e9:7.2f} TB/s")
