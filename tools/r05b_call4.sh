#!/bin/bash
# round 5, second session, call 4: the table's optimiser state as one 32-byte record per entry against three dense arrays
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05b4; mkdir -p $o
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id:" | head -1
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $o/gpu_tests.log; cat $o/gpu_tests.log
for rep in 1 2; do
  echo "records:  $(timeout 200 python tools/r05_accum_cold.py 2>/dev/null | tail -1)"
  echo "separate: $(timeout 200 python tools/r05_accum_cold.py separate 2>/dev/null | tail -1)"
done 2>&1 | tee $o/accum_cold.txt
mb() { timeout 120 python bench.py --microbench $1 --reps 30 $3 2>/dev/null | grep '^{' | tail -1 | cut -c1-200 | sed "s/^/$2 /"; }
for rep in 1 2; do
  mb ngp_encode_bwd records
  NS_VARIANTS=1 NS_ADAM_SEPARATE=1 mb ngp_encode_bwd separate --allow-env-overrides
done 2>&1 | tee $o/microbench.txt
for rep in 1 2 3; do
  for arm in records separate; do
    if [ $arm = separate ]; then export NS_VARIANTS=1 NS_ADAM_SEPARATE=1; else unset NS_VARIANTS NS_ADAM_SEPARATE; fi
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
