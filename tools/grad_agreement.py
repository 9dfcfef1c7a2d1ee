"""Along a real training trajectory (bench model at SIZE^3, 512 keypoints, 2 pairs, Adam lr 1e-3, default f16x3 arithmetic)
compare, every few steps and at the SAME parameters and batch, the gradient of the default arithmetic with the native
fp32-MFMA gradient: relative L2 over all parameters and the worst parameter tensor.  (Fresh-initialisation agreement says
nothing about dead keypoint channels, saturated ReLUs or gradient spikes -- those appear after a few steps.)
Usage (GPU box): python tools/grad_agreement.py [SIZE=64] [STEPS=60] [TYPE=tps_0] [LR=1e-3] [TRAIN_MODE=f16x3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model, train_step
from keymorph_amd import ops, parallel, synthetic, backbone_ops as B

size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tt = sys.argv[3] if len(sys.argv) > 3 else "tps_0"
lr = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-3
train_mode = sys.argv[5] if len(sys.argv) > 5 else "f16x3"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = build_model(int(os.environ.get("KEYPOINTS", "512")), dev)     # bench.build_model takes the keypoint count
flat = parallel.FlatParams(model.parameters())
opt = parallel.FusedAdam(flat, lr=lr)
pairs = [synthetic.make_pair(size, i, dev) for i in range(2)]
img_f = torch.cat([p[0] for p in pairs]).contiguous()
img_m = torch.cat([p[1] for p in pairs]).contiguous()
names = [n for n, p in model.named_parameters() if p.requires_grad]


def grads(mode):
    B.set_conv_mode(mode)
    flat.zero_grad()
    res = model(img_f, img_m, transform_type=tt, return_aligned_points=False)[tt]
    loss, _ = ops.warp_mse(img_m, res["grid"], img_f)
    loss.backward()
    return float(loss.detach()), flat.grad.clone()


worst_all = 0.0
for step in range(steps + 1):
    if step % 10 == 0:
        l32, g32 = grads("f32")
        l16, g16 = grads("f16x3")
        rel = float((g16 - g32).norm() / g32.norm())
        o, worst, wname = 0, 0.0, ""
        for n, p in zip(names, flat.params):
            k = p.numel()
            d = float((g16[o:o + k] - g32[o:o + k]).norm() / g32[o:o + k].norm().clamp_min(1e-30))
            if d > worst:
                worst, wname = d, n
            o += k
        worst_all = max(worst_all, rel)
        print(f"step {step:3d}  loss f32 {l32:.6f} f16x3 {l16:.6f}  grad rel-L2 {rel:.2e}  worst tensor {worst:.2e} ({wname})")
    B.set_conv_mode(train_mode)
    train_step(model, flat, opt, img_f, img_m, tt)
    if step % 10 != 9 and not bool(torch.isfinite(flat.flat).all()):
        print("non-finite parameters after step", step, "in mode", train_mode)
        break
print("max rel-L2 along the trajectory: %.2e" % worst_all)
