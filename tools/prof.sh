#!/bin/bash
# usage (on the GPU box): tools/prof.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/ + printed kernel table
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $out.log 2>&1
python $GRAFT_REPO_ROOT/tools/kstats.py $out 13 22
