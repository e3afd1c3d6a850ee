#!/bin/bash
# rocprofv3 passes for one command (kernel trace + separate PMC passes).
# usage: tools/profile.sh <tag> -- python3 <script> <args...>
# Results: gpurun_out/prof_<tag>/{trace,pmc_*}/...csv and a summary printed by
# tools/profile_summary.py
set -u
tag=$1; shift; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- "$@" > $out/trace.log 2>&1
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc_$i -- "$@" > $out/pmc_$i.log 2>&1
done
python3 tools/profile_summary.py $out
