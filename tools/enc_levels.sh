#!/bin/bash
# usage (GPU box): tools/enc_levels.sh -> cost of every level of k_encode_planes alone (all 1.29 M points on ONE XCD slot)
cd $GRAFT_REPO_ROOT
for l in $(seq 0 16); do
  mask=$(( 0x1ffff & ~(1 << l) ))
  echo -n "level $l alone: "; NERFSIG_ENC_SKIP=$mask python tools/encode_only.py 10 2>/dev/null | tail -1
done
echo -n "none (launch + point loads only): "; NERFSIG_ENC_SKIP=0x1ffff python tools/encode_only.py 10 2>/dev/null | tail -1
echo -n "all: "; python tools/encode_only.py 10 2>/dev/null | tail -1
