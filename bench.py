#!/usr/bin/env python3
"""Headline benchmark: volume-pairs/sec, forward+backward(+Adam), 256^3, 512 keypoints, TPS lambda=0.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process IS
one of the N ranks; started plainly (`python bench.py --gpus N`, WORLD_SIZE unset) it starts the N ranks itself (self_launch)
and refuses -- exit code 2 -- when fewer than N devices are visible.  Either way the line says n_gpus = rccl_ranks = N.

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
    ap.add_argument("--also-f32", type=int, default=3,
                    help="N > 0: also time N steps with the exact fp32-MFMA convolutions (v_mfma_f32_32x32x2_f32) and "
                         "report them as f32_mfma_* next to the f16x3 headline (N = 1 rank only)")
    ap.add_argument("--first-block-exact", type=int, default=3,
                    help="N > 0: also time N steps with the first encoder block's 32 -> 16 data gradient on bf16x6 (24 bits at "
                         "every magnitude; backbone_ops.set_first_block_dgrad) -> first_block_exact_ms_per_step: the price of "
                         "the option DESIGN section 4 describes; never the headline")
    ap.add_argument("--amp", type=int, default=3,
                    help="N > 0: also time N steps with KeyMorph(use_amp=True) -- the one-product fp16 backbone "
                         "(keymorph/model.py:176-191) -> amp_pairs_per_s, amp_roofline against 2.5 PFLOP/s; never the headline")
    ap.add_argument("--dice", type=int, default=3,
                    help="N > 0: also time N steps of the Dice branch (scripts/train.py:146-164: a 14-class one-hot "
                         "segmentation warped with the same grid + DiceLoss as the loss) -> dice_pairs_per_s")
    ap.add_argument("--eval-steps", type=int, default=3,
                    help="N > 0: also time N evaluation passes (scripts/pairwise_register_eval.py:116-171: model.eval(), "
                         "no_grad, a list of transform types, aligned points, align_img per type) -> eval_pairs_per_s")
    ap.add_argument("--groupwise", type=int, default=8,
                    help="S > 0: also time KeyMorph.groupwise_register over S synthetic subjects at --size (BASELINE "
                         "configs[4]: 8 subjects, 512 keypoints, TPS, num_iters 5) -> groupwise_subjects_per_s")
    ap.add_argument("--convnet", type=int, default=3,
                    help="N > 0: also time N training steps with the ConvNet(instance norm) backbone (keymorph/net.py:7-36)")
    ap.add_argument("--sampler", type=int, default=1, help="report stand-alone align_img GB/s for C = 1 and C = 14")
    ap.add_argument("--cpu-256", type=int, default=1,
                    help="1: also time the oracle ONCE at the metric's shape (size^3, 512 keypoints, affine, fwd+bwd; "
                         "~1 min, ~50 GB of host RAM; skipped when less than 96 GB is available)")
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_devices():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def self_launch(a):
    """`python bench.py --gpus N` with no launcher (the way the driver calls --gpus 1): start the N ranks here, one per visible
    device, rendezvous on 127.0.0.1 -- what scripts/run.py:390 (nn.DataParallel) is replaced by.  Rank 0 inherits stdout (the
    ONE JSON line); the other ranks' stdout is folded into stderr.  Any rank failing ends the job non-zero.
    Test hooks (1-GPU / CPU boxes only): KEYMORPH_SHARE_GPU=1 + KEYMORPH_DIST_BACKEND=gloo put every rank on device 0;
    KEYMORPH_BENCH_LAUNCH_CHECK=1 stops every rank after the process group's first collective (no HIP work: runs on CPU)."""
    import subprocess
    n = a.gpus
    ndev = visible_devices()
    hooks = os.environ.get("KEYMORPH_SHARE_GPU") == "1" or os.environ.get("KEYMORPH_BENCH_LAUNCH_CHECK") == "1"
    if ndev < n and not hooks:
        print(f"bench.py: --gpus {n} needs {n} visible devices, this box shows {ndev}; refusing to print a line that "
              f"would say n_gpus {ndev or 1}", file=sys.stderr)
        sys.exit(2)
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KEYMORPH_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            c = p.poll()
            if c is None:
                continue
            live.remove(p)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
                print(f"bench.py: rank {procs.index(p)} exited with {c}; stopping the other ranks", file=sys.stderr)
                for q in live:            # exactly the PIDs started above
                    q.terminate()
    sys.exit(rc)


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
        if os.environ.get("KEYMORPH_BENCH_DICE_UNFUSED"):      # A/B: the three-launch route with the warped tensor stored
            loss = loss_ops.DiceLoss()(utils.align_img(res["grid"], seg_m), seg_f)
        else:
            loss = loss_ops.warp_dice_loss(res["grid"], seg_m, seg_f)        # align_img + DiceLoss, nothing materialised
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
    """HBM bytes per launch of the kernels whose name starts with `prefix`, from the committed rocprofv3 --pmc summary
    that profiles/LATEST declares to be HEAD's (FETCH_SIZE and WRITE_SIZE are collected in separate passes of
    this same command and corrected as MI355X_MICROARCH.md prescribes: 2 x FETCH_SIZE + WRITE_SIZE; see
    tools/pmc_traffic.py).  Counters cannot be read from inside the timed run, so this is the profiled value
    of the same workload, or None when no summary is present."""
    # WHICH summary: profiles/LATEST names the tag of the profile set collected at (or last before) HEAD -- written by the
    # person who copies a set from gpurun_out/ into profiles/ (tools/profile_round.sh prints the line).  Not the file name
    # (tags are labels, "r5zz" sorts after "r5last" and is older) and not the mtime (a checkout gives every file the same).
    pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    try:
        tag = open(os.path.join(pdir, "LATEST")).read().split()[0]
    except (OSError, IndexError):
        return None, "profiles/LATEST missing: no profile set is declared to be HEAD's"
    f = os.path.join(pdir, f"{tag}_pmc_hbm_traffic.json")
    if not os.path.exists(f):
        return None, f"profiles/LATEST names {tag}, which has no {tag}_pmc_hbm_traffic.json"
    files = [f]
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
            "kernel": f"conv3_fwd_s_kernel / conv3_fwd_g_kernel / conv3_fwd_bf_kernel behind kmh_conv3d_fwd_bf (fp32 results from {mult} x "
                      f"{insn} per product block, fp32 accumulate)",
            "achieved": conv_tf, "peak": peak, "unit": "TFLOP/s", "frac": conv_tf / peak,
            "mfma_util": conv_tf * mult / MFMA_BF16_PEAK_TFLOPS,
            "vs_fp32_mfma_peak": conv_tf / MFMA_FP32_PEAK_TFLOPS, "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE)", "traffic_source": src,
            "clock_note": "peak is the guide's 2.5 PFLOP/s at 2.4 GHz; these launches sustain 1.79-1.87 GHz under the power "
                          "cap (GRBM_GUI_ACTIVE / duration, profiles/r4k_effective_clock.txt), i.e. frac x 1.29 of the peak "
                          "at the sustained clock"}


def cpu_baseline(threads, seconds):
    """BASELINE.json configs[0] measured, not extrapolated: the oracle (CPU restatement, same ATen ops as the
    reference; pinned against the reference's own outputs for exactly this pair, tests/test_cfg1_gpu.py) on the host
    cores -- the example_data_half pair at 128^3 (intensity = label / 13; tests/golden/
    cfg1_example_half_128.npz), 128 keypoints, affine aligner, TruncatedUNet3D(f_maps 32), fwd + bwd, MSE.
    One warm-up + >= 2 timed pairs, bounded by `seconds`.  (The full-size oracle run is cpu_baseline_at_size below.)"""
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
    big = None
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
    return {"seconds_per_pair": dt, "pairs": n, "threads": nthreads, "host_cores": ncores, "cpu_model": cpu, "data": data,
            "big": big}


def cpu_baseline_at_size(size, keypoints, threads):
    """BASELINE configs[1] shape on the host: the same synthetic pair recipe as the GPU legs (blob volume and an
    affine-warped copy, generated on the CPU by the oracle), 512 keypoints, affine aligner, fwd + bwd, MSE; ONE pair, cold
    (a pair is two minutes), in a CHILD process with its own thread pool (tests/oracle_at_size.py).  The result is KEPT:
    the HIP path then runs on the same pair and weights -> `parity_at_size`."""
    avail = 0.0
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                avail = float(line.split()[1]) / 2 ** 20
    except OSError:
        pass
    if avail < 96:
        return {"skipped": f"only {avail:.0f} GiB of host RAM available (needs ~50, wants 96)"}
    from tests.oracle_at_size import oracle_pair
    ncores = os.cpu_count()
    nthreads = ncores if threads <= 0 else min(threads, ncores)
    ref = oracle_pair(size, keypoints, threads=nthreads, tt="affine", seed=100, sd_seed=23)
    return {"seconds_per_pair": ref["seconds"], "forward_seconds": ref["forward_seconds"], "size": size,
            "keypoints": keypoints, "mem_available_gib": avail, "ref": ref}


def gpu_sensors(device_index):
    """(power in W, shader clock in MHz) of this rank's GPU from the amdgpu hwmon files -- power1_average (uW) and freq1_input
    (Hz) under /sys/class/drm/card*/device/hwmon/hwmon*/ -- matched to the HIP device by PCI bus id; (None, None) when the
    files are not there.  No rocm-smi, no extra process: two small file reads right after the timed steps."""
    import glob
    try:
        pr = torch.cuda.get_device_properties(device_index)
        want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    except Exception:      # noqa: BLE001
        want = None
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    hit = None
    for c in cards:
        try:
            if want and os.path.basename(os.path.realpath(c)).startswith(want):
                hit = c
                break
        except OSError:
            pass
    if hit is None and len(cards) > device_index and want is None:
        hit = cards[device_index]
    if hit is None:
        return None, None

    def rd(pat):
        for f in glob.glob(os.path.join(hit, "hwmon", "hwmon*", pat)):
            try:
                return float(open(f).read().strip())
            except (OSError, ValueError):
                pass
        return None
    pw, fq = rd("power1_average"), rd("freq1_input")
    if pw is None:
        pw = rd("power1_input")
    return (pw / 1e6 if pw is not None else None), (fq / 1e6 if fq is not None else None)


def rccl_transport_summary(path):
    """What the RCCL debug log of rank 0's communicator set-up says about the wires: counts of the `via <transport>` suffixes
    of its channel lines (P2P/IPC = xGMI or PCIe peer access, SHM = host memory, NET/... = sockets / IB) plus the lines that
    name xGMI / the chosen algorithm.  The first 8-GPU run then says by itself which transport carried the all-reduce."""
    import collections
    import re
    via, notes = collections.Counter(), []
    try:
        for line in open(path, errors="replace"):
            m = re.search(r" via (\S+)", line)
            if m:
                via[m.group(1)] += 1
            if re.search(r"xgmi|XGMI|Connected all rings|Connected all trees|NCCL_ALGO|NCCL_PROTO|Using network|comm 0x.* rank 0 nranks", line):
                if len(notes) < 12:
                    notes.append(line.strip()[-160:])
    except OSError as e:
        return {"error": f"no RCCL debug log: {e}"}
    return {"via": dict(via), "lines": notes}


XGMI_LINK_GBPS = 153.0          # MI355X_MICROARCH.md: 7 point-to-point xGMI links per GPU, ~153 GB/s each


def allreduce_expectation(nbytes, world, measured_ms):
    """What the step's one exchange should cost over xGMI, next to what it did: a ring all-reduce moves 2 (n-1)/n of the bucket
    through every rank's slowest link (xGMI is point-to-point: one ring is bound by ONE ~153 GB/s link; RCCL's several rings /
    trees over the 7 links can approach 7x that), plus ~2 (n-1) hop latencies of a few microseconds.  A measured time far above
    `ring_one_link_ms` means the exchange is NOT on xGMI (see rccl_transport) or the bucket is being split."""
    if world <= 1:
        return None
    moved = 2.0 * (world - 1) / world * nbytes
    one = 1e3 * moved / (XGMI_LINK_GBPS * 1e9)
    return {"ring_one_link_ms": one, "all_links_ms": one / min(7, max(world - 1, 1)), "bytes_per_rank_on_the_wire": moved,
            "link_gb_s": XGMI_LINK_GBPS, "measured_over_one_link": (measured_ms / one) if (measured_ms and one) else None}


def check_rccl_transport(summary, backend):
    """RCCL must carry the gradient bucket over peer-to-peer device memory (xGMI).  Channels `via SHM` (host memory bounce) or
    `via NET/...` (sockets / IB) on a single node mean a misconfigured box (IPC disabled, HSA_ENABLE_IPC_MODE_LEGACY set wrong,
    P2P off): the scaling numbers of such a run are not MI355X numbers, so the run fails instead of reporting them.
    KEYMORPH_BENCH_ALLOW_HOST_TRANSPORT=1 reports them anyway (flagged)."""
    if backend != "nccl" or not isinstance(summary, dict):
        return None
    bad = {k: v for k, v in (summary.get("via") or {}).items() if k.upper().startswith(("SHM", "NET"))}
    if bad and not os.environ.get("KEYMORPH_BENCH_ALLOW_HOST_TRANSPORT"):
        raise SystemExit(f"bench.py: RCCL connected channels through host memory / the network ({bad}; all: {summary.get('via')}) "
                         "instead of peer-to-peer xGMI -- refusing to report multi-GPU numbers from this box "
                         "(KEYMORPH_BENCH_ALLOW_HOST_TRANSPORT=1 overrides)")
    return {"host_or_network_channels": bad} if bad else {"host_or_network_channels": {}}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)          # does not return
    launch_check = os.environ.get("KEYMORPH_BENCH_LAUNCH_CHECK") == "1"
    from keymorph_amd import parallel
    if int(os.environ.get("WORLD_SIZE", "1")) != a.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={os.environ.get('WORLD_SIZE')} but --gpus {a.gpus}")
    if a.gpus > 1 and not launch_check and os.environ.get("KEYMORPH_SHARE_GPU") != "1" and visible_devices() < a.gpus:
        print(f"bench.py: --gpus {a.gpus} needs {a.gpus} visible devices, this box shows {visible_devices()}", file=sys.stderr)
        sys.exit(2)
    rccl_log = None
    if a.gpus > 1 and int(os.environ.get("RANK", "0")) == 0 and not launch_check and "NCCL_DEBUG" not in os.environ:
        # rank 0's communicator set-up says which transport carries the gradients: keep its debug log (INIT / GRAPH only: a few
        # dozen lines, written once when the first collective builds the communicator)
        import tempfile
        rccl_log = os.path.join(tempfile.gettempdir(), f"keymorph_rccl_rank0_{os.getpid()}.log")
        os.environ["NCCL_DEBUG"] = "INFO"
        os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,GRAPH,ENV"
        os.environ["NCCL_DEBUG_FILE"] = rccl_log
    rank, local, world = parallel.init_distributed()
    if launch_check:
        # launcher test hook: the process group the gradients would use, one collective, no HIP work
        probe = torch.ones(1, device="cuda" if torch.cuda.is_available() else "cpu")
        if world > 1:
            torch.distributed.all_reduce(probe)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "rccl_ranks": int(probe.item()),
                              "backend": torch.distributed.get_backend() if world > 1 else None,
                              "self_launched": os.environ.get("KEYMORPH_BENCH_SELF_LAUNCHED") == "1"}))
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    from keymorph_amd import _lib, backbone_ops, synthetic
    backbone_ops.set_conv_mode(a.conv)
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
    power_w, sclk_mhz = gpu_sensors(dev.index if dev.index is not None else 0)      # right after the timed steps
    sensors = torch.tensor([power_w if power_w is not None else float("nan"), sclk_mhz if sclk_mhz is not None else float("nan")],
                           dtype=torch.float64, device=dev)
    rank_sensors = [sensors.tolist()]
    rank_ms = [1000 * dt / a.steps]
    allreduce_ms = 0.0
    if world > 1:
        every_s = [torch.empty_like(sensors) for _ in range(world)]
        torch.distributed.all_gather(every_s, sensors)
        rank_sensors = [x.tolist() for x in every_s]
        t = torch.tensor([dt], device=dev)
        every = [torch.empty_like(t) for _ in range(world)]
        torch.distributed.all_gather(every, t)
        rank_ms = [1000 * float(x.item()) / a.steps for x in every]       # each rank's own clock around the same K steps
        dt = max(float(x.item()) for x in every)                          # MAX over ranks
        # the step's only exchange, timed alone: the flat gradient bucket summed over the ranks (RCCL over xGMI)
        g = flat.grad.clone()
        torch.distributed.all_reduce(g)
        sync()
        ta = time.perf_counter()
        for _ in range(5):
            torch.distributed.all_reduce(g)
        sync()
        allreduce_ms = 1000 * (time.perf_counter() - ta) / 5
        del g
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
                                     f"(bilinear) + DiceLoss fwd+bwd as the fused loss_ops.warp_dice_loss (the warped "
                                     f"segmentation is never stored; both one-hot tensors are re-checked on the device "
                                     f"EVERY step and then read as one byte per voxel) in place of warp+MSE, {a.dice} timed step(s)"})
        del seg_f, seg_m
    if a.also_f32 > 0 and a.conv != "f32":      # the same step on the exact fp32 MFMA, in the same driver run
        backbone_ops.set_conv_mode("f32")
        timed(1)
        dt_f, _ = timed(a.also_f32)
        backbone_ops.set_conv_mode(a.conv)
        extra.update({"f32_mfma_ms_per_step": 1000 * dt_f, "f32_mfma_pairs_per_s": a.pairs_per_gpu * world / dt_f,
                      "f32_mfma_note": f"same step with KEYMORPH_HIP_CONV=f32 (v_mfma_f32_32x32x2_f32, no operand "
                                       f"splitting), {a.also_f32} timed step(s)"})
    if a.first_block_exact > 0 and a.conv == "f16x3":
        backbone_ops.set_first_block_dgrad("bf16x6")
        n0 = backbone_ops.FIRST_BLOCK_STATS["exact_dgrads"]
        timed(1)
        dt_x, loss_x = timed(a.first_block_exact)
        backbone_ops.set_first_block_dgrad("")
        extra.update({"first_block_exact_ms_per_step": 1000 * dt_x, "first_block_exact_pairs_per_s": a.pairs_per_gpu * world / dt_x,
                      "first_block_exact_loss": loss_x,
                      "first_block_exact_launches_per_step": (backbone_ops.FIRST_BLOCK_STATS["exact_dgrads"] - n0) // (a.first_block_exact + 1),
                      "first_block_exact_note": "same step with KEYMORPH_FIRST_BLOCK_DGRAD=bf16x6: the 32 -> 16 data gradient of "
                                                "the first encoder block (256^3) with three bf16 terms / six products on the "
                                                "generic kernel instead of the pre-split fp16 hi/lo operand; what it buys: "
                                                "tests/test_fullsize_gpu.py::test_first_block_dgrad_selector_vs_fp64_oracle_128"})
    if a.amp > 0 and a.conv == "f16x3":      # use_amp=True: the one-product fp16 backbone, in the same driver run
        model.use_amp = True
        timed(1)
        dt_a, loss_a = timed(a.amp)
        _lib.profiler.reset()
        _lib.profiler.enabled = True
        train_step(model, flat, opt, img_f, img_m, tt)
        pa = _lib.profiler.summary()
        _lib.profiler.enabled = False
        _lib.profiler.reset()
        model.use_amp = False
        ca = {"ms": 0.0, "flops": 0.0}
        for nm in ("kmh_conv3d_fwd_bf", "kmh_conv3d_fwd_bf_pool"):
            for k in ca:
                ca[k] += pa.get(nm, {}).get(k, 0)
        amp_tf = ca["flops"] / max(ca["ms"], 1e-9) / 1e9
        extra.update({"amp_pairs_per_s": a.pairs_per_gpu * world / dt_a, "amp_ms_per_step": 1000 * dt_a, "amp_loss": loss_a,
                      "amp_roofline": {"bound": "mfma", "achieved": amp_tf, "peak": 2500.0, "unit": "TFLOP/s",
                                       "frac": amp_tf / 2500.0,
                                       "what": "27-tap forward / data-gradient family under use_amp: algorithmic flops / HIP-event "
                                               "time against the dense fp16 MFMA peak (one product per block)"},
                      "amp_wgrad_tflops": pa.get("kmh_conv3d_wgrad_bf", {}).get("flops", 0) /
                                          max(pa.get("kmh_conv3d_wgrad_bf", {}).get("ms", 0), 1e-9) / 1e9,
                      "amp_note": f"KeyMorph(use_amp=True) (keymorph/model.py:176-191 autocasts the extractor to fp16): the 27-tap "
                                  f"forward / data-gradient kernels, the weight gradient and the fused decoder operator multiply "
                                  f"fp16 hi terms only (fp32 accumulation and tensors), and so do the fused head's forward and "
                                  f"its two masked backward kernels; first layer, aligner, warp and loss unchanged; {a.amp} timed step(s); NOT the headline (a reduced-precision configuration)"})

    def agree(ok):
        """True iff every rank says ok (one MIN all-reduce; ranks must reach this together)"""
        if world == 1:
            return bool(ok)
        f = torch.tensor([1.0 if ok else 0.0], device=dev)
        torch.distributed.all_reduce(f, op=torch.distributed.ReduceOp.MIN)
        return bool(f.item() > 0.5)

    def max_over_ranks(t):
        if world == 1:
            return t
        tt_ = torch.tensor([t], device=dev)
        torch.distributed.all_reduce(tt_, op=torch.distributed.ReduceOp.MAX)
        return float(tt_.item())

    def side_leg(name, fn, need_gib=0.0):
        """The side figures must never cost the headline line: a failure is reported under `<name>_error`.  `fn` does this
        rank's work and returns (seconds, finish) -- `finish(seconds_max_over_ranks)` files the figures.  Under N > 1 the
        ranks first agree that every one of them has `need_gib` of free HBM (a leg with a collective inside -- groupwise's
        all-gather -- is skipped everywhere rather than entered by some ranks), and agree on success BEFORE the timing
        reduction, so that one rank's failure cannot leave the others waiting in a collective."""
        free = (torch.cuda.mem_get_info()[0] + torch.cuda.memory_reserved() - torch.cuda.memory_allocated()) / 2 ** 30
        if not agree(free >= need_gib):
            extra[name + "_error"] = f"skipped: a rank has less than {need_gib:.0f} GiB of free HBM (this rank: {free:.0f})"
            return
        t, finish, err = 0.0, None, None
        try:
            t, finish = fn()
        except Exception as e:      # noqa: BLE001
            err = f"{type(e).__name__}: {e}"[:300]
        if agree(err is None):
            finish(max_over_ranks(t))
        else:
            extra[name + "_error"] = err or "another rank failed"
        torch.cuda.empty_cache()

    def eval_leg():
        # scripts/pairwise_register_eval.py:116-171 / scripts/register.py:264-275: model.eval(), no_grad, one backbone
        # pass, every transform type of the list fitted and evaluated, aligned points, the moving image warped per type
        from keymorph_amd import utils
        types = ["rigid", "affine", "tps_10", "tps_0"]
        model.eval()

        def one():
            with torch.no_grad():
                res = model(img_f, img_m, transform_type=types, return_aligned_points=True)
                return [utils.align_img(res[t]["grid"], img_m) for t in types]
        try:
            one()
            sync()
            t = time.perf_counter()
            for _ in range(a.eval_steps):
                one()
            sync()
            t = (time.perf_counter() - t) / a.eval_steps
        finally:
            model.train()
        return t, lambda t: extra.update({
            "eval_pairs_per_s": a.pairs_per_gpu * world / t, "eval_ms_per_pass": 1000 * t,
            "eval_config": f"model.eval(), no_grad, bs={a.pairs_per_gpu} pair(s)/GPU, transform types {types} from "
                           f"ONE keypoint extraction, aligned points + align_img per type, {a.eval_steps} timed pass(es)"})

    def groupwise_leg():
        # BASELINE configs[4]: KeyMorph.groupwise_register (keymorph/model.py:295-530) over S subjects at full size,
        # TPS, num_iters 5; under N > 1 the subjects are sharded over the ranks (one all-gather of the keypoints)
        from keymorph_amd.transformations import AffineTransform
        from keymorph_amd.utils import align_img
        S, iters, gtype = a.groupwise, 5, "tps_0" if tt.startswith("tps") else tt
        base = synthetic.blob_volume(a.size, 7, dev)
        with torch.no_grad():
            stack = torch.cat([align_img(AffineTransform(matrix=synthetic.random_affine_matrix(40 + i, dev, scale=0.1,
                                                                                               shift=0.1, rot=0.2, shear=0.05),
                                                         dim=3).get_flow_field(base.shape), base) for i in range(S)])
        model.eval()

        def one():
            with torch.no_grad():
                return model.groupwise_register(stack, transform_type=[gtype], device=dev, num_iters=iters,
                                                save_results_to_disk=False)
        reps = 3
        try:
            r = one()
            sync()
            t = time.perf_counter()
            for _ in range(reps):
                r = one()
            sync()
            t = (time.perf_counter() - t) / reps
        finally:
            model.train()
        ok = bool(torch.isfinite(r[gtype]["grouppoints_a"]).all())
        return t, lambda t: extra.update({
            "groupwise_subjects_per_s": S / t, "groupwise_ms": 1000 * t, "groupwise_finite": ok,
            "groupwise_config": f"BASELINE configs[4]: {S} subjects x {a.size}^3, {a.keypoints} keypoints, {gtype}, "
                                f"num_iters {iters}, eval / no_grad, grids kept in HBM; subjects sharded over "
                                f"{world} rank(s), one all-gather of the keypoints; {reps} timed runs"})

    def sampler_leg():
        # stand-alone align_img (keymorph/utils.py:14-21) at the metric's volume size: forward 20 B/voxel for C = 1
        # (grid 12 + volume 4 + out 4), 12 + 8 C in general; C = 14 is the one-hot segmentation of the Dice branch
        from keymorph_amd import utils
        from keymorph_amd.transformations import AffineTransform
        grid = AffineTransform(matrix=synthetic.random_affine_matrix(3, dev), dim=3).get_flow_field(img_f[:1].shape)
        out = {}
        lib = _lib.load()
        S_ = a.size
        for C in (1, 14):
            vol = torch.rand(1, C, S_, S_, S_, device=dev)
            res = torch.empty_like(vol)
            st = torch.cuda.current_stream().cuda_stream

            def kernel_only():          # the C-ABI entry on preallocated tensors: the KERNEL's rate
                lib.kmh_grid_sample3d_fwd(vol.data_ptr(), grid.data_ptr(), res.data_ptr(), 1, C, S_, S_, S_, S_, S_, S_, 0, st)

            def through_api():          # utils.align_img: + allocation, autograd bookkeeping and ctypes per call (~0.1 ms of
                utils.align_img(grid, vol)      # host time, which a 0.09 ms kernel cannot hide)
            ms = {}
            for name, fn in (("kernel", kernel_only), ("api", through_api)):
                with torch.no_grad():
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                ms[name] = e0.elapsed_time(e1) / 20
            gb = S_ ** 3 * (12 + 8 * C) / 1e9
            out[f"C{C}"] = {"ms": ms["kernel"], "ms_through_align_img": ms["api"], "algorithmic_GB": gb,
                            "GBps": gb / ms["kernel"] * 1e3, "frac_of_hbm_peak": gb / ms["kernel"] * 1e3 / HBM_PEAK_GBS}
            del vol, res
        # calibration: a plain device copy moving the same number of bytes as the C = 1 warp (half read, half written)
        nb = a.size ** 3 * 20 // 2
        src = torch.empty(nb // 4, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        dst.copy_(src)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        out["copy_same_bytes"] = {"ms": ms, "GB": 2 * nb / 1e9, "GBps": 2 * nb / ms / 1e6,
                                  "note": "torch copy_ of the C = 1 warp's algorithmic byte count: what a streaming kernel "
                                          "reaches on this chip at this size"}
        # the target is stated against the 8 TB/s pin-rate peak; what a plain stream reaches on this chip is the measured copy
        cp = out["copy_same_bytes"]["GBps"]
        for C in (1, 14):
            out[f"C{C}"]["frac_of_measured_copy_rate"] = out[f"C{C}"]["GBps"] / cp
        return 0.0, lambda _t: extra.update({"align_img_standalone": out})

    def convnet_leg():
        # the reference's other backbone (keymorph/net.py:7-36, instance norm): same step, same sizes
        from keymorph_amd.model import KeyMorph
        from keymorph_amd.net import ConvNet
        torch.manual_seed(23)
        cm = KeyMorph(ConvNet(3, 1, a.keypoints, "instance"), a.keypoints, 3, max_train_keypoints=None).to(dev).train()
        cflat = parallel.FlatParams(cm.parameters())
        copt = parallel.FusedAdam(cflat, lr=3e-6)
        train_step(cm, cflat, copt, img_f, img_m, tt)
        sync()
        t = time.perf_counter()
        for _ in range(a.convnet):
            lo = train_step(cm, cflat, copt, img_f, img_m, tt)
        sync()
        t = (time.perf_counter() - t) / a.convnet
        lo = float(lo.item())
        return t, lambda t: extra.update({
            "convnet_pairs_per_s": a.pairs_per_gpu * world / t, "convnet_ms_per_step": 1000 * t, "convnet_loss": lo,
            "convnet_config": f"ConvNet(instance norm, 9 blocks, {a.keypoints} keypoints) backbone, same step and "
                              f"sizes as the headline, {a.convnet} timed step(s)"})

    vol_gib = a.size ** 3 * 4 / 2 ** 30
    if a.eval_steps > 0:
        side_leg("eval", eval_leg, need_gib=a.pairs_per_gpu * 40 * vol_gib)
    if a.groupwise > 0:       # the all-gather inside: entered by every rank or by none
        side_leg("groupwise", groupwise_leg, need_gib=math.ceil(a.groupwise / world) * 60 * vol_gib)
    if a.sampler > 0 and world == 1:
        side_leg("align_img_standalone", sampler_leg)
    if a.convnet > 0 and world == 1:
        side_leg("convnet", convnet_leg)
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
    # the dominant kernel family: every launch of conv3_fwd_g_kernel / conv3_fwd_bf_kernel, i.e. kmh_conv3d_fwd_bf and its
    # pooling-epilogue variant kmh_conv3d_fwd_bf_pool (same kernel, the output written pooled: 1/8 of the output bytes)
    conv_names = ("kmh_conv3d_fwd",) if a.conv == "f32" else ("kmh_conv3d_fwd_bf", "kmh_conv3d_fwd_bf_pool")
    conv_alg_bytes = [4.0 * m["shape"][0] * m["shape"][1] * m["shape"][2] * m["shape"][3] *
                      (m["shape"][4] + m["shape"][5] * (0.125 if name.endswith("_pool") else 1.0))
                      for name, _, _, m in _lib.profiler.records if name in conv_names and m and "shape" in m]
    _lib.profiler.enabled = False
    peak_mem = torch.cuda.max_memory_allocated() / 2 ** 30

    if rank == 0:
        conv = {"ms": 0.0, "flops": 0.0, "calls": 0}
        for nm in conv_names:
            for k in conv:
                conv[k] += prof.get(nm, {}).get(k, 0)
        conv["calls"] = max(conv["calls"], 1)
        wg = prof.get("kmh_conv3d_wgrad" if a.conv == "f32" else "kmh_conv3d_wgrad_bf", {"ms": 0.0, "flops": 0.0, "calls": 1})
        conv_tf = conv["flops"] / max(conv["ms"], 1e-9) / 1e9
        total_ms = sum(v["ms"] for v in prof.values())
        gs = prof.get("kmh_warp_mse_fwd_grad", prof.get("kmh_warp_mse_fwd", prof.get("kmh_grid_sample3d_fwd", {"ms": 0, "bytes": 0})))
        gsb = prof.get("kmh_grid_sample3d_bwd_grid", {"ms": 0, "bytes": 0})
        backend = torch.distributed.get_backend() if world > 1 else None
        transport = (rccl_transport_summary(rccl_log) if (rccl_log and world > 1 and backend == "nccl") else
                     ({"backend": backend} if world > 1 else None))
        transport_check = check_rccl_transport(transport, backend)      # exits loudly on SHM / NET channels
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
                "launcher": ("bench.py self_launch" if os.environ.get("KEYMORPH_BENCH_SELF_LAUNCHED") == "1" else
                             "torch.distributed.run / external") if world > 1 else None,
                "rank_ms_per_step_min_max": [min(rank_ms), max(rank_ms)],
                # every rank's GPU right after the timed steps (amdgpu hwmon: power1_average, freq1_input): ranks under
                # different power caps / clocks show up here before they show up as a scaling loss
                "rank_power_w_min_max": ([min(p for p, _ in rank_sensors), max(p for p, _ in rank_sensors)]
                                         if all(p == p for p, _ in rank_sensors) else None),
                "rank_sclk_mhz_min_max": ([min(c for _, c in rank_sensors), max(c for _, c in rank_sensors)]
                                          if all(c == c for _, c in rank_sensors) else None),
                "rccl_transport": transport,
                "rccl_transport_check": transport_check,
                "allreduce_ms_per_step": allreduce_ms,
                "allreduce_bytes": 4 * flat.numel,
                "allreduce_expected": allreduce_expectation(4 * flat.numel, world, allreduce_ms),
                "pair_seeds_rank0": [100 * rank + i for i in range(a.pairs_per_gpu)],
                "pair_seed_rule": "rank r owns pairs 100 r + i, i < pairs-per-gpu (synthetic.make_pair seeds)",
                "parity": "keypoints / grid / warped volume / MSE / Dice within 1e-4 of the reference arithmetic wherever the "
                          "problem is well conditioned (tests/: affine, rigid, tps_lambda >= 0.1; 120x120x90 ragged and 128^3 "
                          "whole-path runs at 1e-6..4e-6).  THIS configuration (512 keypoints, lambda = 0, random-init clumped "
                          "keypoints) has cond(A) ~ 1e6: the reference's own fp32 path is 3.8e-4 from the fp64 truth on the "
                          "grid there, and parity is |ours - truth| <= 1.25 |reference - truth| "
                          "(tests/test_ops_gpu.py::test_tps_k512_lambda0_vs_truth), not a plain 1e-4 statement.  GRADIENTS (outside "
                          "north_star's 1e-4 outputs): whole parameter-gradient vector 3.3e-4 from the oracle's fp64 autograd at "
                          "128^3 / 512 kp where the reference's fp32 is 3.8e-4, BUT the first encoder block's cancelling sums are 3-5 x "
                          "FURTHER from fp64 than the reference's fp32 (first GroupNorm weight / bias 4.6e-3 / 4.9e-3 against 1.4e-3 / "
                          "1.0e-3); running that block's data gradient on bf16x6 does not change it (first_block_exact_* keys, "
                          "DESIGN section 4)",
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
            # order matters: the full-size oracle (child process) and the HIP comparison first, the in-process 128^3 sample
            # (which changes this process's intra-op thread count) last
            big = cpu_baseline_at_size(a.size, a.keypoints, a.cpu_threads) if a.cpu_256 else {}
            par_out = None
            if "ref" in big:            # oracle result at the metric's size vs the HIP path on the same pair and weights
                from tests.oracle_at_size import compare_with_hip
                try:
                    ref_big = big.pop("ref")
                    par = compare_with_hip(ref_big, dev)
                    per, per_bb = par.pop("gradient_per_tensor"), par.pop("backbone_gradient_per_tensor")
                    if a.conv != "f32":
                        # the same comparison with the exact fp32-MFMA kernels: what is arithmetic (f16x3 split operands) and
                        # what is implementation, separately in the line
                        pf = compare_with_hip(ref_big, dev, mode="f32")
                        for k in ("keypoints", "matrix", "grid", "warped", "mse", "gradient_rel_l2", "backbone_gradient_rel_l2",
                                  "backbone_gradient_worst_tensor", "backbone_gradient_worst_rel_l2", "tail_rel_l2_hip"):
                            if k in pf:
                                par["f32_mode_" + k] = pf[k]
                    del ref_big
                    par["backbone_gradient_tensors_above_1e-3"] = {k: round(v, 6) for k, v in per_bb.items() if v > 1e-3}
                    par["gradient_notes"] = ("gradient_rel_l2: loss.backward() end to end; backbone_gradient_rel_l2: the "
                                             "backbone's backward alone, HIP and oracle on the SAME d(loss)/d(keypoints); "
                                             "tail_rel_l2_*: d(loss)/d(keypoints) of each fp32 path vs the oracle's fp64 "
                                             "tail (the ill-conditioned part: tests/oracle_at_size.py)")
                    par["what"] = ("HIP path (default arithmetic) vs the CPU oracle's forward and AUTOGRAD backward on the same "
                                   f"{par['size']}^3 pair, seeded weights (23): max-abs differences; gradients relative L2")
                    par_out = par
                except Exception as e:      # noqa: BLE001
                    par_out = {"error": f"{type(e).__name__}: {e}"[:300]}
            c = cpu_baseline(a.cpu_threads, a.cpu_seconds)
            vox_ratio = (a.size / 128) ** 3
            if par_out is not None:
                out["parity_at_size"] = par_out
            measured = "seconds_per_pair" in big
            # `value` is in the headline's units AT the headline's volume size and keypoint count: the measured
            # size^3 / 512-keypoint pair when the host could run it, else the 128^3 sample scaled by the voxel ratio
            value = 1.0 / big["seconds_per_pair"] if measured else 1.0 / (c["seconds_per_pair"] * vox_ratio)
            out["cpu_baseline"] = {
                "value": value,
                "unit": "pairs/s",
                "cores": c["threads"],
                "host_cores_available": c["host_cores"],
                "cpu_model": c["cpu_model"],
                "kind": "port",
                "config": {"aligner": "affine (headline: tps_0)", "size": big.get("size", 128) if measured else 128,
                           "keypoints": big.get("keypoints", 128) if measured else 128, "pairs_timed": 1 if measured else c["pairs"],
                           "warm": not measured},
                "comparable_to_headline": "same volume size and keypoint count, AFFINE aligner (512-keypoint TPS in "
                                          "training mode cannot run on the reference CPU path: 103 GB temporaries, "
                                          "BASELINE.md section 2; its chunked TPS costs more, so this flatters the CPU)",
                "sample": (f"oracle (torch CPU restatement) at BASELINE configs[1] shape: ONE {big.get('size')}^3 synthetic pair, "
                           f"{big.get('keypoints')} keypoints, affine, fwd+bwd, MSE: {big.get('seconds_per_pair', 0):.1f} s "
                           f"(forward {big.get('forward_seconds', 0):.1f} s), measured, not scaled; {c['threads']} threads"
                           if measured else
                           f"oracle at BASELINE configs[0] (128^3, 128 keypoints, affine) scaled by the voxel ratio "
                           f"{vox_ratio:.0f}: {big.get('skipped', 'the full-size sample was not requested')}"),
                "measured_cfg0_pairs_per_s": 1.0 / c["seconds_per_pair"],
                "measured_cfg0_sample": f"BASELINE configs[0]: 128^3 pair ({c['data']}), 128 keypoints, affine, "
                                        f"TruncatedUNet3D(f_maps 32), fwd+bwd, MSE: {c['pairs']} timed pairs at "
                                        f"{c['seconds_per_pair']:.2f} s/pair after one warm-up; {c['threads']} threads = the "
                                        f"fastest of a sweep on this host type (256 threads: 73.7 s/pair)",
                "cfg0_scaled_to_headline_size_pairs_per_s": 1.0 / (c["seconds_per_pair"] * vox_ratio),
            }
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
