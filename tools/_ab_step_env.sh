cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for e in "" "$@"; do
    r=$(env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --windows 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['timing']['ms_per_step_windows'], 'fwd us', round(d['roofline_mlp']['avg_launch_s']*1e6,1), 'enc us', round(d['roofline']['avg_launch_s']*1e6,1))")
    echo "rep $rep [${e:-default}] $r"
  done
done
