#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sc in 1 2; do for gs in 1e-3 2e-6; do NS_FB_SCATTER=$sc python tools/r04_bwd_ab.py 0.9 rays $gs 2>&1 | grep "gradient sigma\|bit-identical" | sed "s/^/SCATTER=$sc /" | cut -c1-200; done; done
timeout 300 python -m pytest tests/test_ngp_gpu.py -x -q -m gpu 2>&1 | tail -2
NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 800 320 2>&1 | tail -2 | head -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04s_bench.json 2>/dev/null; python - <<PY
import json
d=json.load(open("gpurun_out/r04s_bench.json")); r=d["roofline"]
print(round(d["value"],1), [round(w["frames_per_s"],1) for w in d["windows"]], d["breakdown"]["ms_per_frame_by_leg"])
print(r["kernel"], round(r["avg_launch_us"],1), r.get("standalone_us_by_kernel"), r.get("in_step_us_by_kernel"), r.get("records"), r.get("touched_table_entries"), round(r["frac"],3))
PY
