#!/bin/bash
# same-box A/B of the backward schedules (GraphedWatermarkLoop: NERFSIG_BACKWARD_SCHEDULE=tail|beside) at one rank and at emulated ranks of 2, 4, 8
cd $GRAFT_REPO_ROOT
get() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l); print(sys.argv[2], d['ms_per_step'], d['timing']['ms_per_step_windows'], d['config'].get('loss'))" $1 "$2"; }
for rep in 1 2; do
for w in tail beside auto; do
  export NERFSIG_BACKWARD_SCHEDULE=$w
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --windows 3 > gpurun_out/ab_sched1_$w.json 2> gpurun_out/ab_sched1_$w.err || { echo "N=1 $w FAILED"; tail -5 gpurun_out/ab_sched1_$w.err; continue; }
  get gpurun_out/ab_sched1_$w.json "N=1 $w"
done
for R in ${RANKS:-8}; do
for w in tail beside auto; do
  export NERFSIG_BACKWARD_SCHEDULE=$w
  NERFSIG_CAPTURE_COLLECTIVES=1 timeout -k 10 200 python tools/emulate_rank.py $R --steps 20 --warmup 5 --no-secondary --windows 3 > gpurun_out/ab_sched${R}_$w.json 2> gpurun_out/ab_sched${R}_$w.err || { echo "R=$R $w FAILED"; tail -5 gpurun_out/ab_sched${R}_$w.err; continue; }
  get gpurun_out/ab_sched${R}_$w.json "R=$R $w"
done
done
done
