#!/bin/bash
# GPU: tools/spill_probe2.py under several code-generation flags (which one makes the
# wrong results of SGPR-spilling code objects go away?).
# usage: spill_flags.sh <out log> [flag set ...]   (default: the sets of round 2)
out=${1:-gpurun_out/r02_spill_flags.log}
shift
export SF_HIP_UNSAFE_SGPR_SPILLS=1 SF_HIP_REPORT_SGPR_SPILLS=1 SF_HIP_CACHE_DIR=off
: > "$out"
run() { SF_HIP_EXTRA_FLAGS="$1" timeout -k 10 400 python tools/spill_probe2.py 14 >> "$out" 2>&1 || echo "# (exit $?)" >> "$out"; }
if [ $# -eq 0 ]; then
  set -- "" "-mllvm -amdgpu-waitcnt-forcezero=1" "-mllvm -enable-post-misched=0" "-mllvm -sgpr-regalloc=basic" \
    "-mllvm -amdgpu-opt-exec-mask-pre-ra=0" "-mllvm -amdgpu-prealloc-sgpr-spill-vgprs=1" "-mllvm -amdgpu-dpp-combine=0" \
    "-mllvm -amdgpu-enable-pre-ra-optimizations=0" "-mllvm -amdgpu-use-aa-in-codegen=0"
fi
for flags in "$@"; do run "$flags"; done
grep -E "^# (flags|failing)" "$out"
