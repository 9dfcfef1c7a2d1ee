"""Train the bench model for 300 steps at 64^3 / 64 keypoints (TPS 0.1) with a real learning rate and print loss,
max |grad|, max |w| and finiteness every 25 steps, plus how many gradient range scales were carried with the
gradients vs measured.  Usage (GPU box): python tools/stability_check.py [lr]"""
import sys, torch
sys.path.insert(0, "/root/repo")
from bench import build_model, train_step
from keymorph_amd import parallel, synthetic, backbone_ops as B
dev = torch.device("cuda", 0)
for mode in ("f16x3",):
    B.set_conv_mode(mode)
    torch.manual_seed(0)
    model = build_model(64, dev)          # 64 KEYPOINTS (bench.build_model takes the keypoint count), 64^3 volumes below
    flat = parallel.FlatParams(model.parameters())
    opt = parallel.FusedAdam(flat, lr=float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3)
    pairs = [synthetic.make_pair(64, i, dev) for i in range(2)]
    img_f = torch.cat([p[0] for p in pairs]).contiguous(); img_m = torch.cat([p[1] for p in pairs]).contiguous()
    losses = []
    for step in range(300):
        loss = train_step(model, flat, opt, img_f, img_m, "tps_0.1")
        if step % 25 == 0 or step == 299:
            l = float(loss.item()); g = float(flat.grad.abs().max().item()); w = float(flat.flat.abs().max().item())
            print(mode, step, "loss %.6f  max|grad| %.3e  max|w| %.3f  finite %s" % (l, g, w, bool(torch.isfinite(flat.flat).all().item())))
    print("carried / measured grad scales:", B.GRAD_SCALE_STATS)
