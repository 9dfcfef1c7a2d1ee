#!/usr/bin/env python3
"""Headline benchmark: volume-pairs/sec, forward+backward(+Adam), 256^3, 512 keypoints, TPS lambda=0.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched via torch.distributed.run)

One "step" = one training step of scripts/train.py:102-176 restated on this package: KeyMorph.forward
(TruncatedUNet3D on [fixed; moving] -> center of mass -> TPS fit -> dense grid) -> align_img -> MSE ->
backward -> gradient all-reduce (RCCL, N > 1) -> Adam, on synthetic pairs that are resident in HBM before
the timed region starts.  Weak scaling: every rank owns --pairs-per-gpu pairs (default 2 = BASELINE.json
configs[2] "TPS lambda=0, bs=2 on 1xMI355X" at N=1 and configs[3] "bs=16 across 8 GPUs" at N=8).
Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the dominant
kernel (the 3x3x3 conv, fp32 results from split-bf16 MFMA) and `cpu_baseline` (the oracle on the host cores, bounded sample).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_FP32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 (v_mfma_f32_32x32x16_bf16)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--keypoints", type=int, default=512)
    ap.add_argument("--transform", default="tps_0")
    ap.add_argument("--pairs-per-gpu", type=int, default=2,
                    help="pairs per rank per step; 2 = BASELINE configs[2] (bs=2 on one GPU) and configs[3] (16 pairs / 8 GPUs)")
    ap.add_argument("--conv", default=os.environ.get("KEYMORPH_HIP_CONV", "f16x3"), choices=["f32", "f16x3", "bf16x6"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="host threads of the cpu_baseline leg (0 = every hardware thread).  Measured on the MI355X box "
                         "(2 x EPYC 9575F, 256 hardware threads), s/pair at cfg1: 8 thr 5.65, 16 thr 4.45, 32 thr 4.27, "
                         "64 thr 6.46, 128 thr 12.05, 256 thr 73.7 -- ATen / oneDNN oversubscribe, so the fastest "
                         "setting is the default and the count actually used is reported as `cores`")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="bound of the cpu_baseline sample")
    ap.add_argument("--also-f32", type=int, default=1,
                    help="N > 0: also time N steps with the exact fp32-MFMA convolutions (v_mfma_f32_32x32x2_f32) and "
                         "report them as f32_mfma_* next to the f16x3 headline (N = 1 rank only)")
    ap.add_argument("--dice", type=int, default=1,
                    help="N > 0: also time N steps of the Dice branch (scripts/train.py:146-164: a 14-class one-hot "
                         "segmentation warped with the same grid + DiceLoss as the loss) -> dice_pairs_per_s")
    return ap.parse_args()


def build_model(K, device):
    from keymorph_amd.model import KeyMorph
    from keymorph_amd.unet3d.model import TruncatedUNet3D
    torch.manual_seed(23)  # scripts/run.py:217
    net = TruncatedUNet3D(1, K, 1, final_sigmoid=False, f_maps=32, layer_order="gcr", num_groups=8, num_levels=4,
                          is_segmentation=False, conv_padding=1)
    return KeyMorph(net, K, 3, max_train_keypoints=None).to(device).train()


def train_step(model, flat, opt, img_f, img_m, tt, seg_f=None, seg_m=None):
    """scripts/train.py:102-176.  seg_* given: the Dice branch (loss_fn == "dice"): the one-hot moving segmentation is
    warped with the same grid (bilinear, so that it is differentiable) and soft Dice is the loss."""
    from keymorph_amd import loss_ops, ops, utils
    flat.zero_grad()
    res = model(img_f, img_m, transform_type=tt, return_aligned_points=False)[tt]
    if seg_f is None:
        loss, _img_a = ops.warp_mse(img_m, res["grid"], img_f)   # align_img + MSELoss, one pass
    else:
        loss = loss_ops.DiceLoss()(utils.align_img(res["grid"], seg_m), seg_f)
    loss.backward()
    scale = flat.allreduce_grads()
    opt.step(scale)
    return loss


def synthetic_segmentation(img, classes=14):
    """(N,1,D,H,W) intensity in [0,1] -> (N,classes,D,H,W) one-hot float of equal-width intensity bands (SURVEY 8d:
    "14-class label map from thresholded Gaussians"), through the HIP one-hot kernel."""
    from keymorph_amd import utils
    lab = torch.clamp((img * classes).long(), 0, classes - 1)
    lab[:, :, 0, 0, :classes] = torch.arange(classes, device=img.device)      # every class present in every sample
    return utils.one_hot(lab).float()


def pmc_traffic(prefix):
    """HBM bytes per launch of the kernels whose name starts with `prefix`, from the newest committed
    rocprofv3 --pmc summary under profiles/ (FETCH_SIZE and WRITE_SIZE are collected in separate passes of
    this same command and corrected as MI355X_MICROARCH.md prescribes: 2 x FETCH_SIZE + WRITE_SIZE; see
    tools/pmc_traffic.py).  Counters cannot be read from inside the timed run, so this is the profiled value
    of the same workload, or None when no summary is present."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_hbm_traffic.json")))
    files = [f for f in files if "before" not in f]
    if not files:
        return None, None
    # a traffic summary older than the newest kernel-trace summary describes another kernel mix: refuse it
    traces = sorted(glob.glob(os.path.join(os.path.dirname(files[-1]), "*_kernel_trace_stats.md")))
    tag = lambda f: os.path.basename(f).split("_")[0]      # noqa: E731   "r2c" < "r2d" < "r10a" is not needed: < 10 rounds
    if traces and tag(files[-1]) < tag(traces[-1]):
        return None, f"stale ({os.path.basename(files[-1])} is older than {os.path.basename(traces[-1])})"
    d = json.load(open(files[-1]))
    n = sum(v["launches"] for k, v in d.items() if k.startswith(prefix))
    if not n:
        return None, None
    mb = sum(v["launches"] * v["hbm_MB_per_launch_corrected"] for k, v in d.items() if k.startswith(prefix)) / n
    return mb * 1e6, os.path.basename(files[-1])


def roofline(mode, conv_tf):
    """Dominant kernel = the 3x3x3 conv (forward + data-gradient launches).  `achieved` is ALGORITHMIC
    TFLOP/s (2*27*Cin*Cout flops per output voxel).  In the split modes every algorithmic flop costs 3 (f16x3:
    fp16 hi/lo, products hh + hl + lh) or 6 (bf16x6) 16-bit-MFMA flops, so the roofline for fp32-accurate results
    on the matrix cores is 2500/3 = 833.3 (resp. 2500/6 = 416.7) TFLOP/s; `mfma_util` is the fraction of the raw
    dense fp16/bf16 peak the executed MFMAs reach."""
    if mode == "f32":
        return {"bound": "mfma", "kernel": "conv3_fwd_kernel (v_mfma_f32_32x32x2_f32)", "achieved": conv_tf,
                "peak": MFMA_FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": conv_tf / MFMA_FP32_PEAK_TFLOPS,
                "traffic": None}
    mult = 6 if mode == "bf16x6" else 3
    insn = "v_mfma_f32_32x32x16_bf16" if mode == "bf16x6" else "v_mfma_f32_32x32x16_f16"
    peak = MFMA_BF16_PEAK_TFLOPS / mult
    traffic, src = pmc_traffic("conv3_fwd_")      # conv3_fwd_g_kernel (LDS-DMA staging) + conv3_fwd_bf_kernel
    return {"bound": "mfma",
            "kernel": f"conv3_fwd_g_kernel / conv3_fwd_bf_kernel behind kmh_conv3d_fwd_bf (fp32 results from {mult} x "
                      f"{insn} per product block, fp32 accumulate)",
            "achieved": conv_tf, "peak": peak, "unit": "TFLOP/s", "frac": conv_tf / peak,
            "mfma_util": conv_tf * mult / MFMA_BF16_PEAK_TFLOPS,
            "vs_fp32_mfma_peak": conv_tf / MFMA_FP32_PEAK_TFLOPS, "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE)", "traffic_source": src}


def cpu_baseline(threads, seconds):
    """BASELINE.json configs[0] measured, not extrapolated: the oracle (CPU restatement, same ATen ops as the
    reference; pinned against the reference's own outputs for exactly this pair, tests/test_cfg1_gpu.py) on the host
    cores -- the example_data_half pair at 128^3 (intensity = label / 13; tests/golden/
    cfg1_example_half_128.npz), 128 keypoints, affine aligner, TruncatedUNet3D(f_maps 32), fwd + bwd, MSE.
    One warm-up + >= 2 timed pairs, bounded by `seconds`."""
    import numpy as np
    import platform
    from oracle import keymorph_oracle as O
    from tests.util import unet_shapes, seeded_state_dict
    ncores = os.cpu_count()
    nthreads = ncores if threads <= 0 else min(threads, ncores)
    torch.set_num_threads(nthreads)
    fx = os.path.join(ROOT, "tests", "golden", "cfg1_example_half_128.npz")
    if os.path.exists(fx):
        g = np.load(fx)
        f = torch.from_numpy(g["label_0"].astype(np.float32) / 13.0)[None, None]
        m = torch.from_numpy(g["label_1"].astype(np.float32) / 13.0)[None, None]
        data = "example_data_half label maps / 13, 256^3 -> 128^3 nearest (fixture)"
    else:      # same recipe on a synthetic 14-label map
        gen = torch.Generator().manual_seed(0)
        f = torch.randint(0, 14, (1, 1, 128, 128, 128), generator=gen).float() / 13.0
        m = torch.randint(0, 14, (1, 1, 128, 128, 128), generator=gen).float() / 13.0
        data = "synthetic 14-label maps / 13 (fixture missing)"
    sd = {k: v.requires_grad_(True) for k, v in seeded_state_dict(unet_shapes(128, 32, trunc=1), 23).items()}

    def step():
        for v in sd.values():
            v.grad = None
        r = O.keymorph_forward(lambda x: O.unet3d_forward(sd, x, 4, 1, 8), f, m, "affine")
        O.mse_loss(f, O.align_img(r["grid"], m)).backward()

    step()
    t0 = time.time()
    n = 0
    while n < 2 or (time.time() - t0 < seconds and n < 16):
        step()
        n += 1
    dt = (time.time() - t0) / n
    cpu = platform.processor() or ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    print("# cpu_baseline: %d of %d host cores, %s\n# %s" % (nthreads, ncores, cpu, torch.__config__.show().replace("\n", "\n# ")),
          file=sys.stderr)
    return {"seconds_per_pair": dt, "pairs": n, "threads": nthreads, "host_cores": ncores, "cpu_model": cpu, "data": data}


def main():
    a = parse()
    from keymorph_amd import _lib, backbone_ops, parallel, synthetic
    backbone_ops.set_conv_mode(a.conv)
    rank, local, world = parallel.init_distributed()
    assert world == a.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {a.gpus}"
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)
    _lib.load()
    model = build_model(a.keypoints, dev)
    flat = parallel.FlatParams(model.parameters())
    flat.broadcast(0)
    opt = parallel.FusedAdam(flat, lr=3e-6)
    pairs = [synthetic.make_pair(a.size, 100 * rank + i, dev) for i in range(a.pairs_per_gpu)]
    img_f = torch.cat([p[0] for p in pairs]).contiguous()
    img_m = torch.cat([p[1] for p in pairs]).contiguous()
    tt = a.transform

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        loss = train_step(model, flat, opt, img_f, img_m, tt)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = train_step(model, flat, opt, img_f, img_m, tt)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss.item())
    # a step timed on non-finite parameters would be a measurement of nothing (lambda = 0 TPS is singular once two
    # keypoints coincide): refuse to report it
    if not (math.isfinite(loss_val) and bool(torch.isfinite(flat.flat).all()) and bool(torch.isfinite(flat.grad).all())):
        raise SystemExit(f"bench.py: non-finite loss / parameters / gradients after the timed steps (loss {loss_val})")

    def timed(nsteps, **kw):
        sync()
        t = time.perf_counter()
        for _ in range(nsteps):
            lo = train_step(model, flat, opt, img_f, img_m, tt, **kw)
        sync()
        t = time.perf_counter() - t
        if world > 1:
            tt_ = torch.tensor([t], device=dev)
            torch.distributed.all_reduce(tt_, op=torch.distributed.ReduceOp.MAX)
            t = float(tt_.item())
        return t / nsteps, float(lo.item())

    extra = {}
    if a.dice > 0:          # the Dice branch of the same step: (pairs, 14, size^3) one-hot segmentations, warp + soft Dice
        seg_f, seg_m = synthetic_segmentation(img_f), synthetic_segmentation(img_m)
        timed(1, seg_f=seg_f, seg_m=seg_m)
        dt_d, dice_loss = timed(a.dice, seg_f=seg_f, seg_m=seg_m)
        extra.update({"dice_pairs_per_s": a.pairs_per_gpu * world / dt_d, "dice_ms_per_step": 1000 * dt_d,
                      "dice_loss": dice_loss,
                      "dice_config": f"loss_fn=dice: ({a.pairs_per_gpu},14,{a.size}^3) one-hot segmentations, align_img "
                                     f"(bilinear) + DiceLoss fwd+bwd in place of warp+MSE, {a.dice} timed step(s)"})
        del seg_f, seg_m
    if a.also_f32 > 0 and a.conv != "f32":      # the same step on the exact fp32 MFMA, in the same driver run
        backbone_ops.set_conv_mode("f32")
        timed(1)
        dt_f, _ = timed(a.also_f32)
        backbone_ops.set_conv_mode(a.conv)
        extra.update({"f32_mfma_ms_per_step": 1000 * dt_f, "f32_mfma_pairs_per_s": a.pairs_per_gpu * world / dt_f,
                      "f32_mfma_note": f"same step with KEYMORPH_HIP_CONV=f32 (v_mfma_f32_32x32x2_f32, no operand "
                                       f"splitting), {a.also_f32} timed step(s)"})
    rccl_ranks = 1
    if world > 1:           # an actual collective over the process group the gradients use
        probe = torch.ones(1, device=dev)
        torch.distributed.all_reduce(probe)
        rccl_ranks = int(probe.item())

    # one extra (untimed) step with HIP events around every library launch -> per-kernel roofline
    _lib.profiler.reset()
    _lib.profiler.enabled = True
    train_step(model, flat, opt, img_f, img_m, tt)
    prof = _lib.profiler.summary()
    if os.environ.get("KMH_BENCH_DETAIL") and rank == 0:
        for name, ea, eb, meta in _lib.profiler.records:
            if meta and "shape" in meta:
                ms = ea.elapsed_time(eb)
                print(f"# {name:18s} {str(meta['shape']):34s} {ms:8.3f} ms {meta['flops'] / ms / 1e9:7.1f} TF", file=sys.stderr)
    # algorithmic bytes of the dominant kernel's launches: input + output tensor of each, once (fp32)
    conv_name = "kmh_conv3d_fwd" if a.conv == "f32" else "kmh_conv3d_fwd_bf"
    conv_alg_bytes = [4.0 * m["shape"][0] * m["shape"][1] * m["shape"][2] * m["shape"][3] * (m["shape"][4] + m["shape"][5])
                      for name, _, _, m in _lib.profiler.records if name == conv_name and m and "shape" in m]
    _lib.profiler.enabled = False
    peak_mem = torch.cuda.max_memory_allocated() / 2 ** 30

    if rank == 0:
        conv = prof.get("kmh_conv3d_fwd" if a.conv == "f32" else "kmh_conv3d_fwd_bf", {"ms": 0.0, "flops": 0.0, "calls": 1})
        wg = prof.get("kmh_conv3d_wgrad" if a.conv == "f32" else "kmh_conv3d_wgrad_bf", {"ms": 0.0, "flops": 0.0, "calls": 1})
        conv_tf = conv["flops"] / max(conv["ms"], 1e-9) / 1e9
        total_ms = sum(v["ms"] for v in prof.values())
        gs = prof.get("kmh_warp_mse_fwd_grad", prof.get("kmh_warp_mse_fwd", prof.get("kmh_grid_sample3d_fwd", {"ms": 0, "bytes": 0})))
        gsb = prof.get("kmh_grid_sample3d_bwd_grid", {"ms": 0, "bytes": 0})
        out = {
            "metric": "volume-pairs/sec (fwd+bwd) at 256^3, 512 kp, TPS",
            "value": a.pairs_per_gpu * world * a.steps / dt,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1000 * dt / a.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",   # fp32 tensors, fp32 accumulate; products from split 16-bit MFMA operands (config.arithmetic)
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[2]/[3]: {a.size}^3 synthetic pair(s), {a.keypoints} keypoints, {tt}, "
                            f"bs={a.pairs_per_gpu} pair(s)/GPU, TruncatedUNet3D(f_maps=32, L4, trunc 1, gcr), "
                            f"MSE loss, fwd+bwd+Adam",
                "parallelism": f"dp{world} (pairs sharded, flat-bucket RCCL all-reduce of 16 MB grads)",
                "global_pairs": a.pairs_per_gpu * world,
                "rccl_ranks": rccl_ranks,
                "backend": torch.distributed.get_backend() if world > 1 else None,
                "pair_seeds_rank0": [100 * rank + i for i in range(a.pairs_per_gpu)],
                "pair_seed_rule": "rank r owns pairs 100 r + i, i < pairs-per-gpu (synthetic.make_pair seeds)",
                "arithmetic": {"f16x3": "conv: fp32 operands range-scaled by 2^k and split into fp16 hi+lo, 3 MFMA products, "
                                        "fp32 accumulate (5e-7 vs fp64, like fp32 MFMA); the fused 1x1x1 head uses the same scheme",
                               "bf16x6": "fp32 operands split into bf16 hi+mid+lo, 6 MFMA products, fp32 accumulate",
                               "f32": "v_mfma_f32_32x32x2_f32"}[a.conv],
            },
            "roofline": roofline(a.conv, conv_tf) | {
                "launches": conv["calls"],
                "avg_launch_ms": conv["ms"] / max(conv["calls"], 1),
                "share_of_step_kernel_time": conv["ms"] / max(total_ms, 1e-9),
                "algorithmic_bytes_per_launch": (sum(conv_alg_bytes) / len(conv_alg_bytes)) if conv_alg_bytes else None,
            },
            "kernels_ms": {k: round(v["ms"], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
            "wgrad_tflops": wg["flops"] / max(wg["ms"], 1e-9) / 1e9,
            # warp + MSE + d(loss)/d(grid) are ONE launch (kmh_warp_mse_fwd_grad): 36 algorithmic bytes per voxel
            # (grid 12 + volume 4 + fixed 4 + warped 4 + dgrid 12) where the three-launch route moved 68
            "grid_sample": {"kernel": "sample_fwd_lc_kernel<0,true,true> (warp + MSE + grid gradient, one pass)",
                            "ms": gs["ms"], "algorithmic_GB": gs.get("bytes", 0) / 1e9,
                            "achieved_GBps": gs.get("bytes", 0) / max(gs["ms"], 1e-9) / 1e6,
                            "frac_of_hbm_peak": gs.get("bytes", 0) / max(gs["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS,
                            "separate_bwd_ms": gsb["ms"]},
            "loss": loss_val,
            "peak_mem_gib": peak_mem,
        }
        out.update(extra)
        if not a.no_cpu_baseline and world == 1:      # reported at N = 1 only (rank 0's host cores)
            c = cpu_baseline(a.cpu_threads, a.cpu_seconds)
            vox_ratio = (a.size / 128) ** 3
            out["cpu_baseline"] = {
                "value": 1.0 / c["seconds_per_pair"],
                "unit": "pairs/s",
                "cores": c["threads"],
                "host_cores_available": c["host_cores"],
                "cpu_model": c["cpu_model"],
                "kind": "port",
                "config": "BASELINE configs[0]: 128^3 pair, 128 keypoints, affine, TruncatedUNet3D(f_maps 32), fwd+bwd, MSE",
                "sample": f"oracle (torch CPU restatement) on {c['data']}: {c['pairs']} timed pairs at "
                          f"{c['seconds_per_pair']:.2f} s/pair after one warm-up; measured, not scaled; {c['threads']} "
                          f"threads = the fastest of a sweep on this host type (256 threads: 73.7 s/pair)",
                "extrapolated_256_pairs_per_s": 1.0 / (c["seconds_per_pair"] * vox_ratio),
                "extrapolation": f"value / {vox_ratio:.0f} (voxel ratio to {a.size}^3; optimistic for the CPU: 512-keypoint "
                                 f"TPS costs more than the affine fit, SURVEY section 6)",
            }
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
