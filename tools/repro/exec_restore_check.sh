#!/bin/bash
# Static reproducer of the compiler fault of DESIGN.md 5.1 (ROCm 7.2 LLVM, gfx950): no GPU needed.
# exec_restore_kernel.hip is a kernel this library generated in the middle of round 2 (jacobi3d, three fused
# operators, 64x2 threads x 6 rows; git 54b8989) -- its results on the GPU are wrong in the rows of one thread row.
# The script compiles it with the flags of the library's hipRTC call and prints every join block in which
# register-allocator code (vector copies / AGPR or scratch spills, SGPR split copies) sits AHEAD of the
# `s_or_b64 exec, exec, s[..]` that restores EXEC -- i.e. runs for the lanes of the `if` body only.
# With `-mllvm -sgpr-regalloc=basic` (second run) there is none.
# usage: tools/repro/exec_restore_check.sh
set -eu
here=$(dirname "$(readlink -f "$0")")
tmp=$(mktemp -d)
for extra in "" "-mllvm -sgpr-regalloc=basic"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -include hip/hip_runtime.h \
      -DSF_KERNEL_NAME=sf_star3d_f32_t3_2194cdcc --cuda-device-only $extra -S "$here/exec_restore_kernel.hip" -o "$tmp/k.s" 2>/dev/null
  echo "== flags: ${extra:-(none)}   $(grep -E 'sgpr_spill_count' "$tmp/k.s" | tr -s ' ')"
  awk '
    /^; %bb\.|^\.LBB/ { label = $0; n = 0; next }
    /^[ \t]+(v_accvgpr_(write|read|mov)_b32|v_mov_b(32|64)_e32|scratch_(load|store)|s_mov_b(32|64) (s|vcc))/ { run[n++] = $0; next }
    /^[ \t]+s_or_b64 exec, exec, s\[/ { if (n > 0 && label != "") { print label; for (i = 0; i < n; i++) print run[i]; print $0 " ; <-- EXEC restored only here"; print ""; hits++ } label = ""; n = 0; next }
    /^[ \t]+[a-z]/ { label = ""; n = 0 }
    END { print (hits + 0) " join block(s) with allocator code ahead of the EXEC restore" }' "$tmp/k.s"
done
rm -rf "$tmp"
