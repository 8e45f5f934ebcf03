#!/bin/bash
# usage (GPU box): tools/_ab_env_rank.sh VAR "a b ..." "R1 R2 .." [reps]  -> same-box A/B of the emulated rank-of-R step under VAR=a, VAR=b, ...
cd $GRAFT_REPO_ROOT
var=$1; vals=$2; ranks=$3; reps=${4:-2}
get() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l); print(sys.argv[2], 'ms/step', round(d['ms_per_step'],4), [round(x,4) for x in d['timing']['ms_per_step_windows']], 'loss', round(d['config'].get('loss'),6))" $1 "$2"; }
for rep in $(seq $reps); do
for R in $ranks; do
for v in $vals; do
  export $var=$v
  NERFSIG_CAPTURE_COLLECTIVES=1 timeout -k 10 200 python tools/emulate_rank.py $R --steps 20 --warmup 5 --no-secondary --windows 3 > gpurun_out/ab_envr.json 2> gpurun_out/ab_envr.err || { echo "R=$R $var=$v FAILED"; tail -3 gpurun_out/ab_envr.err; continue; }
  get gpurun_out/ab_envr.json "R=$R $var=$v"
done
done
done
