#!/bin/bash
# usage (GPU box): tools/pmc.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...] -- <python script and args>
# one rocprofv3 run per counter group (counters in their own run, kernel-trace only), then a per-kernel mean table.
tag=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
i=0
for g in "${groups[@]}"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$i
  # (one derived counter such as FETCH_SIZE per group: two of them exceed the hardware's counters and the run aborts and hangs)
  timeout -k 10 240 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -- python "$@" > $out.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc.py $out
  i=$((i+1))
done
