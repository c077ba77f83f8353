#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05d; mkdir -p $o
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $o/tests.log; tail -4 $o/tests.log
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $o/bench_$rep.json 2> $o/bench_$rep.err
  python - <<PY
import json
try:
    d = json.load(open("$o/bench_$rep.json"))
    print("rep $rep: total %.1f median %.1f min %.1f max %.1f | seq %.1f | legs %s" % (d["value"], d["windows_frames_per_s"]["median"], d["windows_frames_per_s"]["min"], d["windows_frames_per_s"]["max"], d["sequential"]["frames_per_s"], d["breakdown"]["ms_per_frame_by_leg"]))
except Exception as e:
    print("rep $rep failed", e); print(open("$o/bench_$rep.err").read()[-1500:])
PY
done
