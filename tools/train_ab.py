"""Two trainings from the same initialisation and data, one per arithmetic (default f16x3 vs native fp32 MFMA): loss
curves side by side.  64^3 volumes, 512 keypoints, 2 pairs, Adam.  Chaotic divergence of two fp32-class trajectories is
expected to grow slowly; a systematic gradient error would separate the curves within tens of steps.
Usage (GPU box): python tools/train_ab.py [STEPS=200] [TYPE=tps_1] [LR=1e-3] [SIZE=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model, train_step
from keymorph_amd import parallel, synthetic, backbone_ops as B

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tt = sys.argv[2] if len(sys.argv) > 2 else "tps_1"
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
size = int(sys.argv[4]) if len(sys.argv) > 4 else 64
dev = torch.device("cuda", 0)
pairs = [synthetic.make_pair(size, i, dev) for i in range(2)]
img_f = torch.cat([p[0] for p in pairs]).contiguous()
img_m = torch.cat([p[1] for p in pairs]).contiguous()
curves = {}
for mode in ("f32", "f16x3"):
    B.set_conv_mode(mode)
    model = build_model(512, dev)                 # seeds itself (torch.manual_seed(23))
    flat = parallel.FlatParams(model.parameters())
    opt = parallel.FusedAdam(flat, lr=lr)
    curves[mode] = [float(train_step(model, flat, opt, img_f, img_m, tt).item()) for _ in range(steps)]
    print(mode, "finite parameters:", bool(torch.isfinite(flat.flat).all()))
B.set_conv_mode("f16x3")
print("step   loss(f32)   loss(f16x3)   rel diff")
for s in list(range(0, steps, max(1, steps // 20))) + [steps - 1]:
    a, b = curves["f32"][s], curves["f16x3"][s]
    print(f"{s:4d}  {a:.6f}   {b:.6f}   {abs(a - b) / max(abs(a), 1e-12):.1e}")
tail = slice(steps - steps // 10, steps)
ma, mb = sum(curves["f32"][tail]) / (steps // 10), sum(curves["f16x3"][tail]) / (steps // 10)
print(f"mean loss over the last {steps // 10} steps: f32 {ma:.6f}  f16x3 {mb:.6f}")
