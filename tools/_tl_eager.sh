cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/prof_eager
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/trainer_shape.py --steps 60 > $out.log 2>&1) || exit 1
python tools/timeline.py $out k_codebook_adam 5 > gpurun_out/eager_timeline.txt
