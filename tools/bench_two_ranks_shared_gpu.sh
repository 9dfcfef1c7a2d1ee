#!/bin/bash
# bench.py --gpus 2 end to end on a ONE-GPU box: two ranks share device 0 over gloo (RCCL refuses two ranks on one
# device; test hooks KEYMORPH_DIST_BACKEND / KEYMORPH_SHARE_GPU in parallel.init_distributed).  Exercises rendezvous,
# broadcast of the flat parameter buffer, the per-step all-reduce of the flat gradient bucket, barrier + max-over-ranks
# timing and the rank-0 JSON line; throughput is meaningless (both ranks time-share one GPU).
export KEYMORPH_DIST_BACKEND=gloo KEYMORPH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --steps 2 --warmup 1 --pairs-per-gpu 1 --dice 0 --also-f32 0 "$@"
