#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05g; mkdir -p $o
NS_VARIANTS=1 NS_FB_LEVEL_ORDER=rev timeout 300 python -m pytest tests/test_ngp_gpu.py -x -q -m gpu -k "table_gradient or backward" 2>&1 | tail -2
for rep in 1 2 3; do
  for arm in fwd rev; do
    if [ $arm = rev ]; then export NS_VARIANTS=1 NS_FB_LEVEL_ORDER=rev; else unset NS_VARIANTS NS_FB_LEVEL_ORDER; fi
    echo "$arm step: $(timeout 200 python tools/r05_step_ablation.py 320 base 2>/dev/null | tail -1)"
    timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --allow-env-overrides > $o/bench_${arm}_$rep.json 2> $o/err.txt
    python - <<PY
import json
try:
    d = json.load(open("$o/bench_${arm}_$rep.json"))
    print("$arm rep $rep: total %.1f median %.1f | seq %.1f | legs %s" % (d["value"], d["windows_frames_per_s"]["median"], d["sequential"]["frames_per_s"], d["breakdown"]["ms_per_frame_by_leg"]))
except Exception as e:
    print("$arm rep $rep failed", e)
PY
  done
done
