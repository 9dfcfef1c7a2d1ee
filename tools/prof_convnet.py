"""Per-entry-point time of one training step with the ConvNet(instance norm) backbone at the bench's sizes (2 pairs x 256^3,
512 keypoints, tps_0): where the ConvNet leg's milliseconds are.  Also torch-level time outside the library (gaps)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keymorph_amd import _lib, backbone_ops, parallel, synthetic
from keymorph_amd.model import KeyMorph
from keymorph_amd.net import ConvNet
import bench
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
norm = sys.argv[3] if len(sys.argv) > 3 else "instance"
dev = torch.device("cuda", 0)
backbone_ops.set_conv_mode("f16x3")
torch.manual_seed(23)
cm = KeyMorph(ConvNet(3, 1, K, norm), K, 3, max_train_keypoints=None).to(dev).train()
flat = parallel.FlatParams(cm.parameters()); opt = parallel.FusedAdam(flat, lr=3e-6)
pairs = [synthetic.make_pair(S, i, dev) for i in range(2)]
f = torch.cat([p[0] for p in pairs]).contiguous(); m = torch.cat([p[1] for p in pairs]).contiguous()
for _ in range(2):
    bench.train_step(cm, flat, opt, f, m, "tps_0")
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3):
    bench.train_step(cm, flat, opt, f, m, "tps_0")
torch.cuda.synchronize(); print(f"step {(time.perf_counter() - t) / 3 * 1e3:.1f} ms")
_lib.profiler.reset(); _lib.profiler.enabled = True
bench.train_step(cm, flat, opt, f, m, "tps_0")
prof = _lib.profiler.summary()
tot = sum(v["ms"] for v in prof.values())
print(f"library entries: {tot:.1f} ms")
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k:34s} {v['calls']:3d} calls {v['ms']:8.2f} ms")
for name, a, b, meta in _lib.profiler.records:
    if meta and "shape" in meta:
        ms = a.elapsed_time(b)
        print(f"# {name:22s} {str(meta['shape']):40s} {ms:8.3f} ms {meta['flops'] / ms / 1e9:7.1f} TF")
