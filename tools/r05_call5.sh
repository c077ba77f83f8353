#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05e; mkdir -p $o
timeout 300 python -m pytest tests/test_update_op_gpu.py tests/test_encoder_gpu.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3 4 5 6; do
  for arm in eager graphs; do
    f=""; [ $arm = graphs ] && f="--encoder-graphs"
    timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline $f > $o/bench_${arm}_$rep.json 2> $o/bench_${arm}_$rep.err
    python - <<PY
import json
try:
    d = json.load(open("$o/bench_${arm}_$rep.json"))
    print("$arm rep $rep: total %.1f median %.1f | seq %.1f | legs %s" % (d["value"], d["windows_frames_per_s"]["median"], d["sequential"]["frames_per_s"], d["breakdown"]["ms_per_frame_by_leg"]))
except Exception as e:
    print("$arm rep $rep failed", e); print(open("$o/bench_${arm}_$rep.err").read()[-800:])
PY
  done
done
