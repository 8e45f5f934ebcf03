#!/bin/bash
# same-box A/B of the weight-gradient kernel (variants built by tools/build_variant.sh <name> -DNSIG_WGRAD_AHEAD=.. -DNSIG_WGRAD_WGS=..)
cd $GRAFT_REPO_ROOT
for v in default "$@"; do
  if [ $v = default ]; then unset NERFSIG_LIB; else export NERFSIG_LIB=$GRAFT_REPO_ROOT/tools/_build/libnerfsig_$v.so; fi
  echo "== $v"
  python tools/stage1_bench.py content --windows 3 2>&1 | grep -E "captured step|eagerly|field_wgrad|hg_levels_scatter"
done
