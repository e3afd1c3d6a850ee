#!/bin/bash
# SQ-level PMC passes for one command: tools/profile_mem.sh <tag> -- python3 ...
# Only SQ_* counters: on this pool the TA_* / TCP_* counters take rocprofv3 down
# (crash, then a hang until the caller's limit), so they are never requested.
# Every pass runs under its own `timeout`.
set -u
tag=$1; shift; shift
out=gpurun_out/profmem_$tag
mkdir -p $out
export TMPDIR=/tmp
i=0
for pmc in "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc_$i -- "$@" > $out/pmc_$i.log 2>&1
done
python3 tools/profile_summary.py $out
