cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06za
AB_REPS=2 AB_CONFIGS="c3 --segment-tries 16;c3 --segment-tries 64;c3" bash tools/ab_configs.sh r06za/ab > gpurun_out/r06za/ab.log 2>&1; cat gpurun_out/r06za/ab/ab.jsonl | cut -c1-130
