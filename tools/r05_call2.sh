#!/bin/bash
# round 5, second GPU call: full GPU suite on the new tree; tracker -> mapper queue depth A/B (one box, interleaved arms)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05b; mkdir -p $o
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $o/tests.log; tail -3 $o/tests.log
for rep in 1 2 3; do
  for d in 2 8 16; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --queue-depth $d > $o/bench_q${d}_$rep.json 2> $o/bench_q${d}_$rep.err
    python - <<PY
import json
try:
    d = json.load(open("$o/bench_q${d}_$rep.json"))
    print("depth $d rep $rep: total %.1f median %.1f min %.1f max %.1f | seq %.1f | legs %s" % (d["value"], d["windows_frames_per_s"]["median"], d["windows_frames_per_s"]["min"], d["windows_frames_per_s"]["max"], d["sequential"]["frames_per_s"], d["breakdown"]["ms_per_frame_by_leg"]))
except Exception as e:
    print("depth $d rep $rep failed", e)
PY
  done
done
