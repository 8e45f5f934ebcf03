#!/bin/bash
# usage (GPU box): tools/prof_stage1.sh <tag>  -> gpurun_out/<tag>_stage1.json (tools/stage1_bench.py --json), rocprofv3 kernel stats of the same command
tag=$1
cd $GRAFT_REPO_ROOT
python tools/stage1_bench.py content --json > gpurun_out/${tag}_stage1.json 2> gpurun_out/${tag}_stage1.txt || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_stage1
rm -rf $out
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/stage1_bench.py content --json --windows 2 > $out.log 2>&1) || exit 1
python tools/kstats.py $out 30 40 > gpurun_out/${tag}_stage1_kernel_summary.txt
cp $(ls $out/*/*_kernel_stats.csv | tail -1) gpurun_out/${tag}_stage1_kernel_stats.csv
