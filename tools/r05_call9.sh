#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05i; mkdir -p $o
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id:" | head -1
for rep in 1 2 3 4; do
  for arm in ordered legacy; do
    if [ $arm = legacy ]; then export NS_VARIANTS=1 NS_MARCH_UNORDERED=1; else unset NS_VARIANTS NS_MARCH_UNORDERED; fi
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
