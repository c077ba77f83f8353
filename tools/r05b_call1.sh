#!/bin/bash
# round 5, second session, call 1: store-decoupled volume build + batched Adam flush against the library of commit b5898d6
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05b1; mkdir -p $o
NEW=nerf-slam_amd/lib/libnerfslam_hip.so; cp $NEW /tmp/new.so; OLD=tools/_bin/libnerfslam_hip_head.so
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id:" | head -1
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $o/gpu_tests.log; cat $o/gpu_tests.log
mb() { timeout 120 python bench.py --microbench $1 --reps 30 $3 2>/dev/null | grep '^{' | tail -1 | cut -c1-220 | sed "s/^/$2 /"; }
for rep in 1 2; do
  cp /tmp/new.so $NEW; mb corr_volume new; mb ngp_encode_bwd new
  NS_VARIANTS=1 NS_VOL_FD=2 mb corr_volume new_fd2 --allow-env-overrides
  cp $OLD $NEW; mb corr_volume old; mb ngp_encode_bwd old
done 2>&1 | tee $o/microbench.txt
cp /tmp/new.so $NEW
timeout 200 python tools/small_conv_bench.py > $o/small_conv.txt 2>&1; tail -16 $o/small_conv.txt | head -15
for rep in 1 2; do
  for arm in new old; do
    if [ $arm = old ]; then cp $OLD $NEW; else cp /tmp/new.so $NEW; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $o/bench_${arm}_$rep.json 2> $o/err.txt
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
cp /tmp/new.so $NEW
