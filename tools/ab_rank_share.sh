#!/bin/bash
# Runs on the GPU box: every ab_libs/lib_*.so takes the in-tree library's place in turn and
# tools/bench_rank_share.py measures the per-rank shares of the strong split (c3, c4 packed) with it:
# gpurun_out/$1/rank_share_<lib>_<cfg>.jsonl   (RS_SKIP_C4=1: the c3 shares only)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-abrs}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
LIB=blackhole-simulation_amd/libgravitas_hip.so
cp $LIB /tmp/lib_orig.so
for so in ab_libs/lib_*.so; do
  name=$(basename $so .so)
  cp $so $LIB
  [ -n "${RS_SKIP_C4:-}" ] || python tools/bench_rank_share.py c4 packed > $O/rank_share_${name}_c4.jsonl 2> $O/err_${name}_c4.txt
  python tools/bench_rank_share.py c3 > $O/rank_share_${name}_c3.jsonl 2> $O/err_${name}_c3.txt
done
cp /tmp/lib_orig.so $LIB
for f in $O/rank_share_*.jsonl; do echo $f; python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(d["n_gpus"], d["frames_in_flight"], d["slowest_rank_ms"], d["compute_side_efficiency"])
PY
done
