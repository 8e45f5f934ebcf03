#!/bin/bash
# usage (GPU box): tools/pmc_stage1.sh <tag>: memory-side and LDS counters of the stage-1 step's kernels (rocprofv3 --pmc, one group per run, kernel trace only) over
# tools/stage1_bench.py; gpurun_out/<tag>_stage1_pmc.txt = per-kernel means over the last 10 launches (the eager pass on one stream).
tag=$1
cd /tmp && export TMPDIR=/tmp
groups=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" \
        "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY")
dirs=()
i=0
for g in "${groups[@]}"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_s1_$i
  rm -rf $out
  timeout -k 10 240 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/stage1_bench.py content --windows 2 --steps 64 > $out.log 2>&1 || { echo "group $i ($g) failed"; tail -5 $out.log; }
  dirs+=("$out")
  i=$((i+1))
done
python3 $GRAFT_REPO_ROOT/tools/pmc_collect.py 10 "${dirs[@]}" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_stage1_pmc.txt
grep -h "eagerly on one stream" $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_s1_0.log >> $GRAFT_REPO_ROOT/gpurun_out/${tag}_stage1_pmc.txt
