#!/bin/bash
# usage (GPU box): tools/prof_rank8.sh <tag>  -> gpurun_out/<tag>_rank8_*: the emulated rank-of-8 line and the kernel timeline of one of its replays
tag=$1
cd $GRAFT_REPO_ROOT
export NERFSIG_CAPTURE_COLLECTIVES=1
python tools/emulate_rank.py 8 --steps 20 --warmup 5 --no-secondary --windows 3 > gpurun_out/${tag}_rank8_bench.json 2> gpurun_out/${tag}_rank8_bench.err || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_rank8
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/emulate_rank.py 8 --steps 20 --warmup 5 --no-secondary --windows 1 > $out.log 2>&1) || exit 1
python tools/kstats.py $out 30 40 > gpurun_out/${tag}_rank8_summary.txt
python tools/timeline.py $out k_adam_prepare 8 > gpurun_out/${tag}_rank8_timeline.txt
