cd $GRAFT_REPO_ROOT
AB_CONFIGS="c3 --width 1920 --height 1080;c3 --width 1280 --height 720;c3 --width 1280 --height 720 --two-streams;c3 --width 960 --height 540;c3 --width 640 --height 360;c3 --width 1280 --height 720 --eye 10,90" AB_REPS=3 AB_STEPS=60 bash tools/ab_configs.sh ab_r06zx > /dev/null 2>&1; wc -l gpurun_out/ab_r06zx/ab.jsonl
