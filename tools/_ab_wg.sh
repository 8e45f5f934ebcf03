#!/bin/bash
# same-box A/B: the decoder's weight gradients in one launch behind the data-gradient chain ("tail") or layer by layer beside it (default)
cd $GRAFT_REPO_ROOT
get() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l); print(sys.argv[2], d['ms_per_step'], d['timing']['ms_per_step_windows'], d['config'].get('loss'))" $1 "$2"; }
for rep in 1 2 3; do
for w in tail pipelined; do
  export NERFSIG_DEC_WGRAD=$w
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --windows 3 > gpurun_out/ab_wg1_$w.json 2> gpurun_out/ab_wg1_$w.err || { echo "N=1 $w FAILED"; tail -5 gpurun_out/ab_wg1_$w.err; continue; }
  get gpurun_out/ab_wg1_$w.json "N=1 $w"
done
for R in ${RANKS:-8}; do
for w in tail pipelined; do
  export NERFSIG_DEC_WGRAD=$w
  NERFSIG_CAPTURE_COLLECTIVES=1 timeout -k 10 200 python tools/emulate_rank.py $R --steps 20 --warmup 5 --no-secondary --windows 3 > gpurun_out/ab_wg${R}_$w.json 2> gpurun_out/ab_wg${R}_$w.err || { echo "R=$R $w FAILED"; tail -5 gpurun_out/ab_wg${R}_$w.err; continue; }
  get gpurun_out/ab_wg${R}_$w.json "R=$R $w"
done
done
done
