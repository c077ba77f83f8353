#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r06c3; mkdir -p $o
for ppl in ; do
  NS_VARIANTS=1 NS_BA_PPL=$ppl timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/prof -o ba -- python tools/ba_c1280_bench.py 10 > /dev/null 2>&1
  echo "PPL=$ppl"; grep "ba_linearize_slot" $o/prof/ba_kernel_stats.csv | cut -c1-120; rm -rf $o/prof
done
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/prof -o ba -- python tools/ba_c1280_bench.py 50 c640 > /dev/null 2>&1
grep "^\"ba_\|^\"void ba_" $o/prof/ba_kernel_stats.csv | grep -v solve_depth | cut -c1-120; rm -rf $o/prof
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_parity_c640_gpu.py tests/test_parity_c1280_gpu.py tests/test_parity_c1280_full_gpu.py tests/test_parallel_ba.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_rccl_gpu.py tests/test_bench_pipeline_gpu.py tests/test_slam_gpu.py -x -q -m gpu 2>&1 | tail -8
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/rccl_one_rank.json"))
    print(json.dumps({k: d.get(k) for k in ("ms_per_step", "list_exchange", "timed_self_exchange", "checks")}))
except Exception as e:
    print("no rccl record", e)
PY
