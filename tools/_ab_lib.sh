#!/bin/bash
# usage (GPU box): tools/_ab_lib.sh <other lib> [reps]  -> same-box A/B of the bench step and of k_field_fwd's eager launches: the built library against another build
cd $GRAFT_REPO_ROOT
other=$1; reps=${2:-3}
get() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l); print(sys.argv[2], 'ms/step', round(d['ms_per_step'],4), [round(x,4) for x in d['timing']['ms_per_step_windows']], 'k_field_fwd us', round(d['roofline_mlp']['avg_launch_s']*1e6,1), 'bwd us', round(d['roofline']['backward_mlp_plus_scatter']['k_field_bwd_s']*1e6,1), 'enc us', round(d['roofline']['avg_launch_s']*1e6,1), 'fixed-blocks ms', round(d['config']['fixed_blocks_variant']['ms_per_step'],4), 'loss', d['config'].get('loss'))" $1 "$2"; }
for rep in $(seq $reps); do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --windows 3 > gpurun_out/ab_lib_a.json 2> gpurun_out/ab_lib_a.err || { echo "built FAILED"; tail -5 gpurun_out/ab_lib_a.err; }
  get gpurun_out/ab_lib_a.json "built lib "
  NERFSIG_LIB=$other timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --windows 3 > gpurun_out/ab_lib_b.json 2> gpurun_out/ab_lib_b.err || { echo "other FAILED"; tail -5 gpurun_out/ab_lib_b.err; }
  get gpurun_out/ab_lib_b.json "other lib "
done
