#!/bin/bash
# usage (GPU box): tools/_ab_owner.sh "<lib> ..." [samples per ray]  -> tools/_ab_owner.py under each build, with the entries / owners split from a kernel trace
cd $GRAFT_REPO_ROOT
for lib in $1; do
  if [ "$lib" = built ]; then unset NERFSIG_LIB; else export NERFSIG_LIB=$GRAFT_REPO_ROOT/$lib; fi
  tag=$(basename $lib .so)
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_owner_$tag
  rm -rf $out
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/_ab_owner.py ${2:-152} > $out.log 2>&1) || { echo "$tag FAILED"; tail -5 $out.log; exit 1; }
  grep "hg_levels_scatter us" $out.log
  python tools/kernel_time.py $out k_level_entries k_scatter_binned k_levels_count k_levels_scan
done
