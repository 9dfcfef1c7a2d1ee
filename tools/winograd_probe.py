#!/usr/bin/env python3
"""Bounded Winograd experiment: ONE layer, 64 -> 64 at 128^3 forward, F(2x2x2, 3x3x3) -- 64 instead of 216
products per 8 outputs -- with the transforms in fp32 BEFORE the fp16 hi / lo split, against conv3_fwd_s_kernel<2> (f16x3) and
the fp64 convolution.  Experiment code, not product: the three phases run as torch / hipBLASLt calls, which is the OPTIMISTIC
bound for any implementation that is not fused into one kernel (a library batched GEMM at its own speed, transforms as plain
streaming passes):
  V = B^T d B   (N tiles x 64 positions x Cin, fp32 -> hi, lo fp16)          8 x the activation tensor
  M_p = V_p U_p (64 batched [tiles x 64] x [64 x 64] GEMMs, 3 products hi*hi + hi*lo + lo*hi, fp32 accumulate)
  Y = A^T M A   (fp32)
Why only unfused: a fused kernel must hold, per group of T tiles, 64 positions x T x 64 couts of fp32 partial results before
the output transform = 16 KB x T; the smallest MFMA tile (T = 16) needs 256 KB -- all 256 accumulator registers of all four
waves of a CU for ONE 16-tile group, leaving no register-level reuse of either operand (every MFMA then needs a fresh A and a
fresh B fragment: 4 KB of LDS reads per 48 MFMA cycles and wave = 340 B / cycle / CU against the LDS's 128); see DESIGN
section 8 (round 6).
usage: python tools/winograd_probe.py [D=128] [N=2]   ->  profiles/r6*_winograd_64x64_128.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keymorph_amd import backbone_ops as B  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
C = 64
dev = "cuda"
torch.manual_seed(0)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32, device=dev)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64, device=dev)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32, device=dev)


def ev(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


def range_scale(t):
    m = float(t.abs().max())
    import math
    return 2.0 ** (15 - math.frexp(m)[1]) if m > 0 else 1.0


def split(t, s):
    r = t * s
    hi = r.half()
    lo = (r - hi.float()).half()
    return hi, lo


x = torch.relu(torch.randn(N, D, D, D, C, device=dev) * 1.2 + 0.1)
w = (torch.rand(C, C, 3, 3, 3, device=dev) * 2 - 1) / (27 * C) ** 0.5

# ---- the product path: conv3_fwd_s_kernel<2> through the C ABI (GroupNorm scale 1 / shift 0, no ReLU on either side) ----
B.set_conv_mode("f16x3")
sc, sh = torch.ones(N, C, device=dev), torch.zeros(N, C, device=dev)
asc = B.absmax_scale(x)
wf = B.pack_weight(w, False)
t_direct, y_direct = ev(lambda: B.conv3_raw(x, sc, sh, wf, None, N, D, D, D, C, C, False, False, ascale=asc), reps=10)

# ---- Winograd, unfused ----
T = D // 2
U = w.double().permute(2, 3, 4, 1, 0)                                # (3,3,3,Cin,Cout)
U = torch.einsum("ia,jb,kc,abcxy->ijkxy", G, G, G, U).float().reshape(64, C, C).contiguous()      # filter transform: once per step
su = range_scale(U)
Uh, Ul = split(U, su)


def input_transform():
    xp = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1, 1, 1))        # (N, D+2, D+2, D+2, C)
    d = xp.unfold(1, 4, 2).unfold(2, 4, 2).unfold(3, 4, 2)           # (N, T, T, T, C, 4, 4, 4) view
    V = torch.einsum("ia,jb,kc,ntuvxabc->ijkntuvx", BT, BT, BT, d)   # (4,4,4,N,T,T,T,C)
    return V.reshape(64, N * T * T * T, C)


def split_v(V, sv):
    return split(V, sv)


def gemms(Vh, Vl):
    M = torch.bmm(Vh, Uh).float()
    M += torch.bmm(Vh, Ul).float()
    M += torch.bmm(Vl, Uh).float()
    return M


def gemms_f32acc(Vh, Vl):
    # fp32 results from fp16 operands: the accumulate the MFMA does; torch exposes it through out_dtype where available
    try:
        M = torch.bmm(Vh, Uh, out_dtype=torch.float32)
        M += torch.bmm(Vh, Ul, out_dtype=torch.float32)
        M += torch.bmm(Vl, Uh, out_dtype=torch.float32)
        return M
    except TypeError:
        return None


def output_transform(M, s):
    M = M.reshape(4, 4, 4, N, T, T, T, C)
    Y = torch.einsum("ia,jb,kc,abcntuvy->ntiujvky", AT, AT, AT, M)   # (N, T,2, T,2, T,2, C)
    return Y.reshape(N, D, D, D, C) * (1.0 / s)


t_in, V = ev(input_transform, reps=3, warm=1)
sv = range_scale(V)
t_sp, (Vh, Vl) = ev(lambda: split_v(V, sv), reps=3, warm=1)
del V
t_g16, M16 = ev(lambda: gemms(Vh, Vl), reps=3, warm=1)
M32 = gemms_f32acc(Vh, Vl)
t_g32 = None
if M32 is not None:
    t_g32, M32 = ev(lambda: gemms_f32acc(Vh, Vl), reps=3, warm=1)
Muse = M32 if M32 is not None else M16
t_out, y_w = ev(lambda: output_transform(Muse, sv * su), reps=3, warm=1)

# ---- the floor of an unfused implementation: its mandatory HBM traffic at this chip's measured copy rate ----
big = torch.empty(int(2e9) // 4, device=dev)
t_copy, _ = ev(lambda: big[: big.numel() // 2].copy_(big[big.numel() // 2:]), reps=5)
copy_gbs = 2 * (big.numel() // 2) * 4 / t_copy / 1e6
del big
vox = N * D ** 3
bytes_v = vox * C * 4 * 8          # V as fp16 hi + lo: 4 bytes per value, 8 x the activation (64 positions per 8 voxels)
bytes_m = vox * C * 4 * 8          # M in fp32
floor_ms = (vox * C * 4 + 2 * bytes_v + 2 * bytes_m + vox * C * 4) / (copy_gbs * 1e6)

# ---- accuracy against the fp64 convolution on a sub-volume (the full fp64 conv at 128^3 is minutes of library time) ----
S = 24
xs = x[:1, :S + 2, :S + 2, :S + 2].permute(0, 4, 1, 2, 3).double()
ref = torch.nn.functional.conv3d(xs, w.double())[0].permute(1, 2, 3, 0)      # outputs at voxels 1..S of the full volume
mx = float(ref.abs().max())
e_dir = float((y_direct[0, 1:S + 1, 1:S + 1, 1:S + 1].double() - ref).abs().max()) / mx
e_win = float((y_w[0, 1:S + 1, 1:S + 1, 1:S + 1].double() - ref).abs().max()) / mx
flops = 2.0 * 27 * C * C * vox
print(f"64 -> 64, N = {N}, {D}^3 forward; measured copy rate {copy_gbs:.0f} GB/s")
print(f"direct  conv3_fwd_s_kernel<2> (f16x3)          {t_direct:8.3f} ms   {flops / t_direct / 1e9:7.1f} TFLOP/s   max err / max |y| {e_dir:.2e}")
print("Winograd F(2x2x2,3x3x3), unfused (torch / hipBLASLt):")
print(f"   input transform (fp32, strided einsum)       {t_in:8.3f} ms")
print(f"   range scale + hi / lo split of V             {t_sp:8.3f} ms")
print(f"   3 x 64 batched GEMMs, fp16 out + fp32 sum    {t_g16:8.3f} ms   (a LOWER bound: fp16 results lose the accuracy)")
if t_g32 is not None:
    print(f"   3 x 64 batched GEMMs, fp32 out               {t_g32:8.3f} ms")
print(f"   output transform (fp32)                      {t_out:8.3f} ms")
tot = t_in + t_sp + (t_g32 if t_g32 is not None else t_g16) + t_out
print(f"   total                                        {tot:8.3f} ms   = {t_direct / tot:.2f} x the direct kernel's speed   max err / max |y| {e_win:.2e}"
      f"{'' if M32 is not None else '  (fp16 GEMM outputs)'}")
print(f"   HBM floor of ANY unfused version (x once, V and M written + read, y once = {(vox * C * 8 + 2 * bytes_v + 2 * bytes_m) / 1e9:.1f} GB "
      f"at the copy rate)  {floor_ms:8.3f} ms   = {t_direct / floor_ms:.2f} x the direct kernel's speed")
print(f"adopt rule: error <= 2e-6 of the tensor maximum AND >= 1.4 x faster  ->  "
      f"{'ADOPT' if (e_win <= 2e-6 and t_direct / tot >= 1.4) else 'not adopted'} "
      f"(error {'passes' if e_win <= 2e-6 else 'fails'}; speed {t_direct / tot:.2f} x, floor {t_direct / floor_ms:.2f} x)")
