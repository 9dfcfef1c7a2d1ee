"""Which ATen ops run inside one headline training step, and what they cost on the GPU (torch.profiler, grouped by op and
input shapes): the housekeeping around the C-ABI launches.  usage: python tools/prof_step_ops.py [size] [keypoints]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from keymorph_amd import parallel, synthetic
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
model = bench.build_model(K, dev)
flat = parallel.FlatParams(model.parameters())
opt = parallel.FusedAdam(flat, lr=3e-6)
pairs = [synthetic.make_pair(size, s, dev) for s in (0, 1)]
img_f, img_m = torch.cat([p[0] for p in pairs]), torch.cat([p[1] for p in pairs])
for _ in range(2):
    bench.train_step(model, flat, opt, img_f, img_m, "tps_0")
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    for _ in range(2):
        bench.train_step(model, flat, opt, img_f, img_m, "tps_0")
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print(f"aten ops with GPU time over 2 steps: {tot / 1e3:.2f} ms self device time in {sum(e.count for e in rows)} calls")
for e in rows[:45]:
    print(f"{e.self_device_time_total / 2e3:8.3f} ms/step  {e.count // 2:4d}/step  {e.key:28s} {str(e.input_shapes)[:110]}")
# host-side runtime calls that can stall the launch queue (per step)
rt = [e for e in prof.key_averages() if e.key.startswith(("hip", "cuda")) and not e.key.startswith(("hipLaunchKernel", "hipExtLaunch", "hipModuleLaunch"))]
rt.sort(key=lambda e: -e.count)
print("runtime calls per step (besides kernel launches):")
for e in rt[:15]:
    print(f"  {e.count / 2:7.1f}/step  {e.cpu_time_total / 2e3:8.3f} ms/step host  {e.key}")
# ---- round 6: WHO launches the elementwise adds (35 `CUDAFunctor_add` launches per step, 0.93 ms, 6.6 GB) ----
# one more step with Python stacks; every aten::add / add_ / sum / mul / copy_ with GPU time is attributed to the chain of its
# CPU parents (the autograd node that ran it: "...: XBackward" = inside that Function's backward, bare "evaluate_function" with an
# AccumulateGrad / no Python frame = the engine summing two gradients of a tensor that was used twice) and its Python frame.
if os.environ.get("KMH_WHO", "1") != "0":
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof2:
        bench.train_step(model, flat, opt, img_f, img_m, "tps_0")
        torch.cuda.synchronize()
    import collections
    who = collections.defaultdict(lambda: [0, 0.0])
    for e in prof2.events():
        if not e.name.startswith(("aten::add", "aten::sum", "aten::mul", "aten::copy_", "aten::cat", "aten::clone", "aten::div", "aten::sub")):
            continue
        if e.self_device_time_total <= 0:
            continue
        chain, p = [], e.cpu_parent
        while p is not None:
            if not p.name.startswith("aten::"):
                chain.append(p.name[:70])
            p = p.cpu_parent
        frame = next((s for s in (e.stack or []) if "keymorph_amd" in s or "bench.py" in s), "")
        key = (e.name, str(e.input_shapes)[:70], " <- ".join(chain[:2]), frame.strip()[-90:])
        who[key][0] += 1
        who[key][1] += e.self_device_time_total
    print("who launches the elementwise / reduction ATen kernels (one step):")
    for key, (n, us) in sorted(who.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{us / 1e3:8.3f} ms {n:3d}x  {key[0]:14s} {key[1]:70s}\n              parents: {key[2]}\n              frame:   {key[3]}")
