#!/bin/bash
# Dry run of the driver's multi-GPU command on a ONE-GPU box: the same torch.distributed.run launch line with two
# ranks, both on GPU 0 (GM_BENCH_ONE_GPU=1: rank r of 2 launches ITS share of the task chunks there), the collective over gloo with the
# counts staged through the host (GM_BENCH_BACKEND=gloo).  What it exercises that a one-rank run cannot: rank shares that differ, the sum
# of the per-rank counts, max-over-ranks timing, per_gpu_kernel_ms of length 2, the CPU-baseline carry-over and the rank-0-only
# rocprofv3 passes with the other rank parked at the closing barrier.
# usage: scripts/scale_dryrun.sh [ranks (2)] [bench.py arguments ...]      prints the JSON line of rank 0
set -e
cd "$(dirname "$0")/.."
N=${1:-2}; shift || true
export GM_BENCH_ONE_GPU=1 GM_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "${MASTER_PORT:-29541}" \
     bench.py --gpus "$N" "$@"
