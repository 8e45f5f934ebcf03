#!/bin/bash
# usage (GPU box): tools/_ab_env.sh VAR "a b ..." [reps]  -> same-box A/B of the bench step under VAR=a, VAR=b, ...
cd $GRAFT_REPO_ROOT
var=$1; vals=$2; reps=${3:-3}
get() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l); print(sys.argv[2], 'ms/step', round(d['ms_per_step'],4), [round(x,4) for x in d['timing']['ms_per_step_windows']], 'fwd us', round(d['roofline_mlp']['avg_launch_s']*1e6,1), 'bwd us', round(d['roofline']['backward_mlp_plus_scatter']['k_field_bwd_s']*1e6,1), 'scatter us', round(d['roofline']['backward_mlp_plus_scatter']['scatter_s']*1e6,1), 'fixed-blocks ms', round(d['config']['fixed_blocks_variant']['ms_per_step'],4), 'loss', round(d['config'].get('loss'),6))" $1 "$2"; }
for rep in $(seq $reps); do
for v in $vals; do
  export $var=$v
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --windows 3 > gpurun_out/ab_env.json 2> gpurun_out/ab_env.err || { echo "$var=$v FAILED"; tail -3 gpurun_out/ab_env.err; continue; }
  get gpurun_out/ab_env.json "$var=$v"
done
done
