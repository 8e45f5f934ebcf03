#!/bin/bash
# usage (GPU box): tools/_ab_prof.sh "<lib> ..." "<kernel name part> ..."  -> per build, the headline step's windows and the named kernels' mean time in a trace of the same command
cd $GRAFT_REPO_ROOT
for lib in $1; do
  if [ "$lib" = built ]; then unset NERFSIG_LIB; else export NERFSIG_LIB=$GRAFT_REPO_ROOT/$lib; fi
  tag=$(basename $lib .so)
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_ab_$tag
  rm -rf $out
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --windows 3 > $out.log 2>&1) || { echo "$tag profile FAILED"; tail -5 $out.log; exit 1; }
  python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l); print(sys.argv[2], 'ms/step (under the profiler)', [round(x,4) for x in d['timing']['ms_per_step_windows']])" $out.log $tag
  python tools/kernel_time.py $out $2
done
