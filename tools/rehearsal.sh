#!/bin/bash
# GPU (one device): full-size rehearsals of `bench.py --gpus N` --
#  - N ranks on device 0 (SF_BENCH_SINGLE_DEVICE=1): the whole N-rank flow -- transport ladder (RCCL refuses two ranks on
#    one device, the DMA pushes take over), decomposition check before and after, the native schedule, per-GPU roofline;
#  - one rank as the inner rank of three whose halos come back to it over the library's RCCL rung (SF_BENCH_SELF_LOOP=1).
# usage: bash tools/rehearsal.sh <tag> [ranks, default 2]
set -o pipefail
tag=${1:-r05}; n=${2:-2}
export HSA_ENABLE_IPC_MODE_LEGACY=0
SF_BENCH_SINGLE_DEVICE=1 timeout -k 10 500 python bench.py --gpus $n --steps 3 --warmup 1 > gpurun_out/${tag}_bench_${n}ranks_one_gpu.json 2>gpurun_out/${tag}_bench_${n}ranks.err; echo "$n ranks rc=$?"; tail -c 2500 gpurun_out/${tag}_bench_${n}ranks_one_gpu.json; echo
SF_BENCH_SELF_LOOP=1 timeout -k 10 400 python bench.py --gpus 1 --steps 3 --warmup 1 > gpurun_out/${tag}_bench_self_loop_rccl.json 2>gpurun_out/${tag}_bench_self.err; echo "self loop rc=$?"; tail -c 2500 gpurun_out/${tag}_bench_self_loop_rccl.json; echo
