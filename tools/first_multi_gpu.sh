#!/bin/bash
# The first thing to run on a node with >= 2 MI355X: the multi-device parity tests (they skip
# themselves on one device), then the scaling points of BASELINE's metric under both hosts --
#   native  : one process, grv_engine_create_multi, one RCCL send/recv group per frame
#   torchrun: one process per GPU, dist.gather over nccl (what the round-end driver starts)
# -- for the f64 frame (c3, strong split) and the 8K f32 march (c4).  One JSON line per point in
# gpurun_out/$1/scale.jsonl; every line carries transport, rccl_version, rank_devices and per-rank
# integrate times, so a point can be audited without trusting n_gpus.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${1:-multi}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python -c "import torch; print(torch.cuda.device_count() if torch.cuda.is_available() else 0)")
echo "HIP devices visible: $NDEV" | tee $O/devices.txt
timeout 3600 python -m pytest tests/test_gpu_multi_real.py tests/test_gpu_multi_native.py tests/test_gpu_dist_shared_device.py \
    -m gpu -q -rs > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -15 $O/pytest.log
for cfg in c3 c4; do
  for n in 1 2 4 8; do
    [ $n -gt $NDEV ] && continue
    for launcher in native torchrun; do
      [ $n -eq 1 ] && [ $launcher = torchrun ] && continue
      timeout 900 python bench.py --config $cfg --gpus $n --launcher $launcher --no-cpu-baseline \
          2> $O/err_${cfg}_${n}_$launcher.txt | tee -a $O/scale.jsonl | cut -c1-220
      # the same point with the gather in the compute pass's own rgba16float format (half the exchange)
      [ $n -gt 1 ] && timeout 900 python bench.py --config $cfg --gpus $n --launcher $launcher --no-cpu-baseline --exchange rgba16f \
          2> $O/err_${cfg}_${n}_${launcher}_h.txt | tee -a $O/scale_rgba16f.jsonl | cut -c1-220
    done
  done
done
python - "$O/scale.jsonl" <<'PY'
import json, sys
pts = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
base = {}
for p in pts:
    key = (p["config"]["baseline_config"], "native" if "native" in p["launcher"] else "torchrun" if "torchrun" in p["launcher"] else "single")
    if p["n_gpus"] == 1:
        base[p["config"]["baseline_config"]] = p["value"]
for p in pts:
    b = base.get(p["config"]["baseline_config"])
    print("%-11s N=%d %-9s %10.0f Mray-steps/s  %7.3f ms/frame  efficiency %s  transport %s  rank ms %s"
          % (p["config"]["baseline_config"], p["n_gpus"], p["launcher"].split(" ")[0], p["value"], p["ms_per_step"],
             ("%.3f" % (p["value"] / (b * p["n_gpus"]))) if b else "-", p.get("transport"),
             (p.get("rank_integrate_ms") or {}).get("per_rank")))
PY
