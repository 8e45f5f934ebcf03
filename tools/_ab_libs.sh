#!/bin/bash
# usage (GPU box): tools/_ab_libs.sh "<lib> <lib> ..." [reps] [prof]  -> same-box A/B of several builds of the library ("built" = the product build, anything else a path for
# NERFSIG_LIB): the headline step (bench.py), the stage-1 step with its per-kernel table (tools/stage1_bench.py), and with `prof` one rocprofv3 kernel trace of the headline
# step per build (the optimiser pass and the scatter owners by name)
cd $GRAFT_REPO_ROOT
libs=$1; reps=${2:-2}; prof=$3
head_line() { python -c "
import json,sys
d=json.load(open(sys.argv[1])); b=d['roofline']['backward_mlp_plus_scatter']      # (the complete record: bench.py's bench_detail.json)
print(sys.argv[2], 'headline ms/step', round(d['ms_per_step'],4), [round(x,4) for x in d['timing']['ms_per_step_windows']], 'scatter us', round(b['scatter_s']*1e6,1), 'encoder us', round(d['roofline']['avg_launch_s']*1e6,1), 'loss', round(d['config'].get('loss'),6))" $1 "$2"; }
s1_line() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l); k=d['kernels_us_per_step']
print(sys.argv[2], 'stage-1 ms/step', round(d['ms_per_step'],4), [round(w['ms_per_step'],4) for w in d['windows']], 'sparse', round(d['sparse_grid']['ms_per_step'],4), 'points', d['points_per_step'], '| scatter', k.get('hg_levels_scatter'), 'adam', k.get('opt_adam_dense'), 'bwd', k.get('field_bwd_wgrad'), 'loss', d['loss_last'])" $1 "$2"; }
for rep in $(seq $reps); do
for lib in $libs; do
  if [ "$lib" = built ]; then unset NERFSIG_LIB; else export NERFSIG_LIB=$GRAFT_REPO_ROOT/$lib; fi
  tag=$(basename $lib .so)
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --windows 3 > gpurun_out/ab_${tag}_head.line 2> gpurun_out/ab_${tag}_head.err || { echo "$tag headline FAILED"; tail -5 gpurun_out/ab_${tag}_head.err; exit 1; }
  cp gpurun_out/bench_detail.json gpurun_out/ab_${tag}_head.json
  head_line gpurun_out/ab_${tag}_head.json "$tag"
  timeout -k 10 300 python tools/stage1_bench.py content --json --steps 64 --windows 3 > gpurun_out/ab_${tag}_s1.json 2> gpurun_out/ab_${tag}_s1.err || { echo "$tag stage-1 FAILED"; tail -5 gpurun_out/ab_${tag}_s1.err; exit 1; }
  s1_line gpurun_out/ab_${tag}_s1.json "$tag"
done
done
if [ -n "$prof" ]; then
for lib in $libs; do
  if [ "$lib" = built ]; then unset NERFSIG_LIB; else export NERFSIG_LIB=$GRAFT_REPO_ROOT/$lib; fi
  tag=$(basename $lib .so)
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_ab_$tag
  rm -rf $out
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --windows 1 > $out.log 2>&1) || { echo "$tag profile FAILED"; tail -5 $out.log; exit 1; }
  python tools/kernel_time.py $out k_codebook_adam_sel k_scatter_binned k_scatter_merge k_adam_dense
done
fi
