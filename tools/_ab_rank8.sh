#!/bin/bash
# usage (GPU box): tools/_ab_rank8.sh "<ENV=val ...>" ["<ENV=val ...>" ...]: the emulated rank-of-8 step (collectives captured) under each environment,
# interleaved twice on the same box
for rep in 1 2; do
  for e in "" "$@"; do
    ms=$(env $e NERFSIG_CAPTURE_COLLECTIVES=1 python tools/emulate_rank.py 8 --steps 100 --warmup 10 --no-secondary --windows 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['timing']['ms_per_step_windows'])")
    echo "rep $rep [${e:-default}] $ms"
  done
done
