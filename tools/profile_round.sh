#!/bin/bash
# Round-end evidence for profiles/: for the C3 (default) and C2 bench workloads
#   1. rocprofv3 --kernel-trace --stats of the bench command (kernel average duration),
#   2. FETCH_SIZE and WRITE_SIZE in separate PMC passes (HBM bytes per launch),
# then tools/hbm_traffic.py folds the counters into profiles/hbm_traffic.json.
# Every rocprofv3 call runs the program directly after `--` and under its own timeout.
# usage (on the GPU box): bash tools/profile_round.sh <round-tag>
set -u
tag=${1:-r01}
export TMPDIR=/tmp
# (the plan-time self-check launches every new fused kernel twice on a few planes: such dispatches
# would enter the per-kernel means of duration and traffic; the code objects are the same without it)
export SF_HIP_SELF_CHECK=0
# The profiler preloads /opt/rocm's libamd_comgr, and hipRTC (PyTorch's copy, the one a plain
# `python bench.py` uses) then compiles through THAT compiler instead of the one bundled with it:
# same source, same kernel name, another code object -- for C5 even another tile shape (the 7.2
# compiler spills at five rows per thread, round 3).  Preloading PyTorch's comgr keeps the profiled
# objects the ones the product runs.  SF_PROFILE_SYSTEM_COMGR=1 turns this off.
if [ "${SF_PROFILE_SYSTEM_COMGR:-0}" != "1" ]; then
  torch_comgr=$(python3 -c "import os, torch; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))" 2>/dev/null)
  [ -f "$torch_comgr" ] && export LD_PRELOAD="$torch_comgr${LD_PRELOAD:+:$LD_PRELOAD}"
fi
# workloads: c3 c2 c5 = the bench workloads; box = the generator's 27-point chain (compact
# kernel); wide = its radius-2 cross chain (wide-star kernel); generic = c3 forced onto the
# generic operator kernel (16 / 40 operators); dense = the generator's 125-point box (dense kernel); fork = its
# fork / join program (several kernels: one record each)
for wl in ${SF_PROFILE_WORKLOADS:-c3 c2 c5 box wide cross3 dense fork generic}; do
  [ "$wl" = none ] && continue
  out=gpurun_out/prof_${tag}_$wl
  rm -rf $out; mkdir -p $out
  case $wl in
    # (the generator's workloads: 30 timed steps behind 5 untimed ones, as in bench.py's `other_configs` -- the mean of a run of
    #  a few dozen launches after a pause is dominated by the clock settling, NOTES.md round 5)
    box) base="--workload box --stages 16 --steps 30 --warmup 5"; short="--workload box --stages 16";;
    wide) base="--workload wide --stages 16 --steps 30 --warmup 5"; short="--workload wide --stages 16";;
    cross3) base="--workload cross3 --stages 8 --steps 30 --warmup 5"; short="--workload cross3 --stages 8";;
    dense) base="--workload dense --stages 4 --steps 30 --warmup 5"; short="--workload dense --stages 4";;
    fork) base="--workload fork --stages 16 --steps 30 --warmup 5"; short="--workload fork --stages 16";;
    generic) base="--workload c3 --stages 40 --options generic_only=1"; short="--workload c3 --stages 40 --options generic_only=1";;
    *) base="--workload $wl"; short="--workload $wl --stages 100";;
  esac
  args="bench.py $short --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs"
  # the trace pass runs the bench command itself (default steps / warm-up), so its
  # average kernel duration is the one the bench line reports
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $base --no-cpu-baseline --no-other-configs > $out/trace.log 2>&1
  i=0
  for pmc in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc_$i -- python3 $args > $out/pmc_$i.log 2>&1
  done
  python3 tools/profile_summary.py $out > /dev/null
  cp $out/summary.txt gpurun_out/${tag}_bench_${wl}_rocprof_summary.txt
  f=$(ls -t $out/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_bench_${wl}_kernel_stats.csv
done
# the slab kernels of bench.py --gpus N (own code objects: the global plane count is a constant of the source):
# rank 0's launches of one process without neighbours, tools/slab_traffic.py
for n in ${SF_PROFILE_SLABS:-2 4 8}; do
  [ "$n" = none ] && continue
  out=gpurun_out/prof_${tag}_slab$n
  rm -rf $out; mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/slab_traffic.py --world $n --out $out/planes.json > $out/trace.log 2>&1
  i=0
  for pmc in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc_$i -- python3 tools/slab_traffic.py --world $n --out $out/planes.json > $out/pmc_$i.log 2>&1
  done
  python3 tools/profile_summary.py $out > /dev/null
  cp $out/summary.txt gpurun_out/${tag}_bench_slab${n}_rocprof_summary.txt
done
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic.json 2>/dev/null
python3 tools/hbm_traffic.py $tag gpurun_out/hbm_traffic.json
