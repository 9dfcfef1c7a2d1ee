"""time the channel-blocked pooling backward of the 256^3 level (4 x 256^3 x 32 channels): KEYMORPH_POOL_BWD_PLANES=1 selects the
kernel that writes 32-byte pieces per plane"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keymorph_amd import _lib
from keymorph_amd.backbone_ops import _p, _stream, check
lib = _lib.load()
dev = "cuda"
torch.manual_seed(3)
N, D, C = 4, 256, 32
arg = torch.randint(0, 8, (N, D // 2, D // 2, D // 2, C), dtype=torch.uint8, device=dev)
dy = torch.randn(N, D // 2, D // 2, D // 2, C, device=dev)
dx = torch.empty(N, C // 8, D, D, D, 8, device=dev)
f = lambda: check(lib.kmh_maxpool3d_bwd(None, _p(arg), _p(dy), None, 0, _p(dx), N, D, D, D, C, 1, _stream()), "bwd")
for _ in range(2): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"blocked pooling backward: {ms:.3f} ms ({dx.numel() * 4 / ms / 1e6:.0f} GB/s written), checksum {float(dx.double().abs().sum()):.10e}")
