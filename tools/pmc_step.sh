#!/bin/bash
# usage (GPU box): tools/pmc_step.sh <tag> [steps]: one rocprofv3 --pmc run of tools/pmc_step.py per counter group (counters in their own runs, kernel
# trace only), then gpurun_out/<tag>_pmc.txt with the per-(kernel, launch size) means over the eager steps.
tag=$1; steps=${2:-6}
cd /tmp && export TMPDIR=/tmp
groups=("FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
        "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
        "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
        "TCC_BUBBLE_sum")      # (gfx950: the 128-byte read requests to the fabric -- the share of 128-byte requests tools/pmc_traffic.py needs; a run of its own beside TCC_EA0_RDREQ's)
dirs=()
i=0
for g in "${groups[@]}"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$i
  timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/pmc_step.py $steps > $out.log 2>&1 || { echo "group $i ($g) failed"; tail -5 $out.log; }
  dirs+=("$out")
  i=$((i+1))
done
python3 $GRAFT_REPO_ROOT/tools/pmc_collect.py $steps "${dirs[@]}" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc.txt
grep -h "PMC_STEP_END" $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_0.log >> $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc.txt
