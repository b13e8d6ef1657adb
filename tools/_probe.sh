cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x -k "test_gpu_schedules or schedules_agree or compaction" 2>&1 | tail -2
