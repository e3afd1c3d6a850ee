#!/bin/bash
# GPU: the fuzz campaigns of a round on every program generator (tests/random_programs.py), undivided and under slab
# decomposition, the plan-time self-check on; one log per campaign under gpurun_out/<tag>_fuzz_*.log (copy the ones that
# are to be judged into profiles/).   usage: bash tools/fuzz_round.sh <tag> [seconds per campaign, default 60] [undivided|slab|all] ["generators"]
# (round 5's longer campaign on the dense families: fuzz_round.sh r05_fuzz2 200 all "dense dense_sum box_sum")
tag=${1:-r05}; secs=${2:-60}; part=${3:-all}
gens=${4:-"star wide compact dense dense_sum box_sum sparse_sum weighted_cross dag copy"}
[ "$part" = slab ] && gens=""
first=$((RANDOM % 5000))
for g in $gens; do
  timeout -k 10 $((secs + 60)) python tools/star_fuzz.py --generator $g --seeds 100000 --first $first --seconds $secs > gpurun_out/${tag}_fuzz_$g.log 2>&1
  echo "fuzz $g rc=$? $(tail -1 gpurun_out/${tag}_fuzz_$g.log)"
done
[ "$part" = slab ] || SF_HIP_OPTIONS="dense.t2=2" timeout -k 10 $((secs + 60)) python tools/star_fuzz.py --generator box_sum --seeds 100000 --first $first --seconds $secs > gpurun_out/${tag}_fuzz_box_sum_forced.log 2>&1
[ "$part" = slab ] || echo "fuzz box_sum (dense.t2=2) rc=$? $(tail -1 gpurun_out/${tag}_fuzz_box_sum_forced.log)"
[ "$part" = slab ] || SF_HIP_OPTIONS="dense.t2=3" timeout -k 10 $((secs + 60)) python tools/star_fuzz.py --generator box_sum --seeds 100000 --first $first --seconds $secs > gpurun_out/${tag}_fuzz_box_sum_three.log 2>&1
[ "$part" = slab ] || echo "fuzz box_sum (dense.t2=3, three per launch) rc=$? $(tail -1 gpurun_out/${tag}_fuzz_box_sum_three.log)"
[ "$part" = slab ] || SF_HIP_OPTIONS="dense.t2=3" timeout -k 10 $((secs + 60)) python tools/star_fuzz.py --generator weighted_cross --seeds 100000 --first $first --seconds $secs > gpurun_out/${tag}_fuzz_weighted_cross_forced.log 2>&1
[ "$part" = slab ] || echo "fuzz weighted_cross (dense.t2=3) rc=$? $(tail -1 gpurun_out/${tag}_fuzz_weighted_cross_forced.log)"
[ "$part" = slab ] || SF_HIP_OPTIONS="dense.t2=2" timeout -k 10 $((secs + 60)) python tools/star_fuzz.py --generator sparse_sum --seeds 100000 --first $first --seconds $secs > gpurun_out/${tag}_fuzz_sparse_sum_forced.log 2>&1
[ "$part" = slab ] || echo "fuzz sparse_sum (dense.t2=2) rc=$? $(tail -1 gpurun_out/${tag}_fuzz_sparse_sum_forced.log)"
[ "$part" = undivided ] && exit 0
slabs="mixed star wide compact dense box_sum sparse_sum weighted_cross dag"
[ -n "${4:-}" ] && slabs=$(for g in $4; do case $g in dense_sum|copy) ;; *) echo $g;; esac; done)
for g in $slabs; do
  opt=""; [ $g = sparse_sum ] && opt="dense.t2=2"; { [ $g = box_sum ] || [ $g = weighted_cross ]; } && opt="dense.t2=3"
  SF_HIP_OPTIONS="$opt" timeout -k 10 $((secs + 60)) python tools/slab_fuzz.py --generator $g --seeds 100000 --first $first --seconds $secs > gpurun_out/${tag}_fuzz_slab_$g.log 2>&1
  echo "slab fuzz $g rc=$? $(tail -1 gpurun_out/${tag}_fuzz_slab_$g.log)"
done
