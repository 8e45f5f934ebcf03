#!/bin/bash
# usage (GPU box): [env ...] tools/_tl.sh <tag> [R]  -> gpurun_out/tl_<tag>.txt: kernel timeline of one replay of the bench step (R given: of the emulated rank of R)
cd $GRAFT_REPO_ROOT
tag=$1; R=$2
out=$GRAFT_REPO_ROOT/gpurun_out/prof_tl_$tag
export NERFSIG_BENCH_VARIANT=0
if [ -n "$R" ]; then
(cd /tmp && export TMPDIR=/tmp NERFSIG_CAPTURE_COLLECTIVES=1 && rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/emulate_rank.py $R --steps 20 --warmup 5 --no-secondary --windows 1 > $out.log 2>&1) || exit 1
else
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --windows 1 > $out.log 2>&1) || exit 1
fi
python tools/timeline.py $out k_adam_prepare 8 > gpurun_out/tl_$tag.txt
