#!/bin/bash
# GPU: tools/spill_probe2.py on the hand-assembled code objects of tools/asm_objects.py.
# usage: asm_objects_run.sh <objects dir> <out log> [variant ...]
dir=$1; out=$2; shift 2
export SF_HIP_UNSAFE_SGPR_SPILLS=1 SF_HIP_REPORT_SGPR_SPILLS=1 SF_HIP_CACHE_DIR=off
: > "$out"
for v in "$@"; do
  echo "## variant $v" >> "$out"
  SF_HIP_OBJECT_DIR=$dir/$v timeout -k 10 300 python tools/spill_probe2.py 14 >> "$out" 2>&1 || echo "# (exit $?)" >> "$out"
done
grep -E "^(## variant|# failing)" "$out"
