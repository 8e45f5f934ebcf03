#!/bin/bash
# usage (GPU box): tools/refresh_prof.sh  -> gpurun_out/refresh_prof_{full,partial}.txt: per-kernel averages of the device-side grid refresh
cd $GRAFT_REPO_ROOT
for form in full partial; do
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_refresh_$form
  rm -rf $out
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/refresh_prof.py $form 10 > $out.log 2>&1) || { echo "$form FAILED"; tail -5 $out.log; exit 1; }
  { tail -1 $out.log; python tools/kstats.py $out 12 30; } > gpurun_out/refresh_prof_$form.txt
  python tools/refresh_prof.py $form 10 --graph >> gpurun_out/refresh_prof_$form.txt 2>/dev/null
done
