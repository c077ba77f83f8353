#!/bin/bash
# feature-encoder layers as one launch each (statistics in the convolution's epilogue, normalisation on load) vs three
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05b10; mkdir -p $o
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > $o/gpu_tests.log; cat $o/gpu_tests.log
for rep in 1 2; do
  echo "fused:   $(timeout 120 python tools/enc_bench.py 2>/dev/null | tail -1)"
  echo "unfused: $(NS_VARIANTS=1 NS_ENC_UNFUSED=1 timeout 120 python tools/enc_bench.py 2>/dev/null | tail -1)"
done 2>&1 | tee $o/enc_bench.txt
for rep in 1 2 3; do
  for arm in fused unfused; do
    if [ $arm = unfused ]; then export NS_VARIANTS=1 NS_ENC_UNFUSED=1; else unset NS_VARIANTS NS_ENC_UNFUSED; fi
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
done 2>&1 | tee $o/bench_ab.txt
