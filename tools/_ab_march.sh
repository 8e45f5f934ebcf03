cd $GRAFT_REPO_ROOT
run() { NERFSIG_MARCH_FUSED=$1 python bench.py --no-secondary --no-cpu-baseline --windows 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['timing']['ms_per_step_windows'], d['config']['fixed_blocks_variant']['ms_per_step'])"; }
for i in 1 2; do for v in 0 nf sw 1; do run $v; done; done 2>&1 | tee gpurun_out/r03_e_march_fused_ab4.txt
export TMPDIR=/tmp
for v in 0 1; do
  NERFSIG_MARCH_FUSED=$v NERFSIG_BENCH_VARIANT=0 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_e_$v -- python3 bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --windows 1 > gpurun_out/r03_e_prof_$v.json 2> gpurun_out/r03_e_prof_$v.err
  python tools/kstats.py gpurun_out/prof_r03_e_$v 30 40 > gpurun_out/r03_e_kstats_$v.txt 2>&1
done
find gpurun_out/prof_r03_e_0 gpurun_out/prof_r03_e_1 -name "*.db" -delete 2>/dev/null; find gpurun_out/prof_r03_e_0 gpurun_out/prof_r03_e_1 -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
tail -3 gpurun_out/r03_e_kstats_1.txt
