#!/bin/bash
# GPU: HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, separate PMC passes) and average duration of the kernels of ONE
# tools/dense_probe.py variant.   usage: bash tools/probe_traffic.sh <tag> <workload> "<plan options>"
# (FETCH_SIZE on gfx950 reports half of a wide coalesced read stream: tools/profile_summary.py prints the counter as
#  reported; double it -- MI355X_MICROARCH.md, HBM)
export TMPDIR=/tmp SF_HIP_SELF_CHECK=0
tag=$1; wl=$2; opts=${3:-}
out=gpurun_out/probe_${tag}
rm -rf $out; mkdir -p $out
torch_comgr=$(python3 -c "import os, torch; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))" 2>/dev/null)
[ -f "$torch_comgr" ] && export LD_PRELOAD="$torch_comgr${LD_PRELOAD:+:$LD_PRELOAD}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/dense_probe.py $wl --no-check --reps 10 --variants "$opts" > $out/trace.log 2>&1
i=0
for pmc in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc_$i -- python3 tools/dense_probe.py $wl --no-check --reps 2 --variants "$opts" > $out/pmc_$i.log 2>&1
done
python3 tools/profile_summary.py $out > /dev/null
cp $out/summary.txt gpurun_out/probe_${tag}_summary.txt
cat $out/summary.txt
