"""Which approximation moves the first block's GroupNorm gradients away from the fp64 truth?  The backbone's backward at 128^3 /
512 keypoints (tests/oracle_at_size.py::oracle_backbone_fp64) under: the default arithmetic, KEYMORPH_NO_STATS_FOLD=1 (GroupNorm's
sum dxn * xhat from a pass over dxn and x instead of the weight-gradient contraction), bf16x6, the exact fp32 MFMA."""
import os, sys, torch
sys.path.insert(0, '.')
from keymorph_amd import backbone_ops as B
from tests.oracle_at_size import hip_model, oracle_backbone_fp64
S, K = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 512
ref = oracle_backbone_fp64(S, K)
g64, g32 = ref["grads_fp64"], ref["grads_fp32"]
names = [k for k in g64 if k.startswith("encoders.0.")] + ["encoders.1.basic_module.SingleConv1.groupnorm.bias", "final_conv.weight"]
def run(tag, mode="f16x3", env=None):
    for k, v in (env or {}).items(): os.environ[k] = v
    B.set_conv_mode(mode)
    try:
        km = hip_model(ref["sd"], K, "cuda")
        pts = km.get_keypoints(ref["x"].cuda())
        torch.autograd.backward([pts], [ref["cot"].cuda()])
        num = den = 0.0
        row = {}
        for k, p in km.backbone.named_parameters():
            t = g64[k].double(); a = p.grad.detach().cpu().double()
            num += float((a - t).pow(2).sum()); den += float(t.pow(2).sum())
            row[k] = float((a - t).norm() / (t.norm() + 1e-300))
        print(f"{tag:34s} whole {(num / den) ** 0.5:.2e}  " + "  ".join(f"{row[k]:.1e}" for k in names))
    finally:
        for k in (env or {}): os.environ.pop(k, None)
        B.set_conv_mode("f16x3")
print("columns:", [k.replace("basic_module.", "").replace("encoders.", "e") for k in names])
num = den = 0.0
for k in g64:
    num += float((g32[k].double() - g64[k]).pow(2).sum()); den += float(g64[k].pow(2).sum())
print(f"{'oracle fp32':34s} whole {(num / den) ** 0.5:.2e}  " + "  ".join(f"{float((g32[k].double() - g64[k]).norm() / g64[k].norm()):.1e}" for k in names))
run("hip f16x3 (default)")
run("hip f16x3 NO_STATS_FOLD", env={"KEYMORPH_NO_STATS_FOLD": "1"})
run("hip f16x3 NO_LAZY_FIRST", env={"KEYMORPH_NO_LAZY_FIRST": "1"})
run("hip f16x3 NO_CONV_POOL", env={"KEYMORPH_NO_CONV_POOL": "1"})
run("hip bf16x6", mode="bf16x6")
run("hip f32 mfma", mode="f32")
