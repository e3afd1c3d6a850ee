#!/bin/bash
# Round 4: full-size rehearsals of `bench.py --gpus N` on the one GPU of the box, on the round's final code:
#  - two ranks on device 0 (SF_BENCH_SINGLE_DEVICE=1): the whole N = 2 flow -- ladder (RCCL refuses two ranks on one
#    device, the DMA pushes take over), check before and after, native schedule, per-GPU roofline;
#  - one rank as the inner rank of three whose halos come back to it over the library's RCCL rung (SF_BENCH_SELF_LOOP=1).
set -o pipefail
OUT=gpurun_out/r04_rehearsal
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
SF_BENCH_SINGLE_DEVICE=1 timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 > $OUT/bench_2ranks_one_gpu.json 2>$OUT/bench_2ranks.err; echo "2 ranks rc=$?"; tail -c 1500 $OUT/bench_2ranks_one_gpu.json; echo
SF_BENCH_SELF_LOOP=1 timeout -k 10 400 python bench.py --gpus 1 --steps 3 --warmup 1 > $OUT/bench_self_loop_rccl.json 2>$OUT/bench_self.err; echo "self loop rc=$?"; tail -c 1500 $OUT/bench_self_loop_rccl.json; echo
