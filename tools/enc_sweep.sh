#!/bin/bash
# usage (GPU box): tools/enc_sweep.sh  -> per-level cost and sc1 variants of k_encode_planes on the block render's points
cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; env $1 python tools/encode_only.py 10 2>/dev/null | tail -1; }
run "X=0"
for f in 0 6 8 10 12 14 16; do run "NERFSIG_ENC_SC1_FROM=$f"; done
run "NERFSIG_ENC_SKIP=0x3f"        # without levels 0-5
run "NERFSIG_ENC_SKIP=0xff"        # without levels 0-7
run "NERFSIG_ENC_SKIP=0x1ff00"     # only levels 0-7
run "NERFSIG_ENC_SKIP=0x1fc00"     # only levels 0-9
run "NERFSIG_ENC_SKIP=0x003ff"     # only levels 10-16
run "NERFSIG_ENC_SKIP=0x0ffff"     # only the codebook level
