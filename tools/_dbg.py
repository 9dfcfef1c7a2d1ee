import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.util import golden, T
from tests.test_parity_r2_gpu import _aligner, tunet
from oracle import keymorph_oracle as O
DEV="cuda"
g, e = golden("realworld_small.npz"), golden("e2e_tiny.npz")
pf, pm = T(e["affine::points_f"]).to(DEV), T(e["affine::points_m"]).to(DEV)
s = torch.tensor([32.,32.,32.]).to(DEV)
kw = dict(dim=3, align_in_real_world_coords=True, aff_f=T(g["aff_f"]).to(DEV), aff_m=T(g["aff_m"]).to(DEV), shape_f=s, shape_m=s)
for tt in ("rigid","affine","tps_10"):
    al = _aligner(tt, pm, pf, None, kw)
    grid = al.get_flow_field((1,1,32,32,32))
    print(tt, "ours on the reference's keypoints vs golden:", float((grid.cpu()-T(g[f"km::{tt}::grid"])).abs().max()))
    r = O.register_real_world(pf.cpu().double(), pm.cpu().double(), tt, (32,32,32), T(g["aff_f"]).double(), T(g["aff_m"]).double(), s.cpu().double(), s.cpu().double())
    print("    ours vs fp64 oracle on the same keypoints:", float((grid.cpu().double()-r["grid"]).abs().max()), " golden vs that:", float((T(g[f"km::{tt}::grid"]).double()-r["grid"]).abs().max()))
    if tt != "tps_10":
        print("    inverse matrix (mm):", al.inverse_transform_matrix.cpu().numpy().round(6).tolist())
        from keymorph_amd.utils import convert_points_norm2real
        rf = convert_points_norm2real(pf.cpu().double(), T(g["aff_f"]).double(), s.cpu().double())
        rm = convert_points_norm2real(pm.cpu().double(), T(g["aff_m"]).double(), s.cpu().double())
        fit = O.rigid_fit if tt=="rigid" else O.affine_fit
        print("    fp64 fit:", O.square(fit(rf, rm)).numpy().round(6).tolist())
