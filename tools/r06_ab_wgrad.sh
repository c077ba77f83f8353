#!/bin/bash
# A/B: the MLP weight-gradient chain forked beside the scatter pass (default) vs after it, in the pipeline,
# two runs per arm interleaved in one call; plus the kernel table of each arm (in-pipeline scatter / accumulate durations).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r06abw; mkdir -p $o
for r in 1 2; do for v in 0 1; do
  NS_BENCH_NGP_CFG=wgrad_after_scatter=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-sensitivity > $o/arm${v}_run$r.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open("$o/arm${v}_run$r.json") if l.startswith("{")][-1])
print("arm $v run $r", round(d["value"],1), d.get("breakdown") or d.get("extra",{}).get("breakdown"))
PY
done; done
for v in 0 1; do
  NS_BENCH_NGP_CFG=wgrad_after_scatter=$v timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $o/prof$v -o b -- python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-sensitivity > /dev/null 2>&1
  cp $o/prof$v/b_kernel_stats.csv $o/arm${v}_kernel_stats.csv; rm -rf $o/prof$v
  grep -E "fscatter|faccum|wgrad_tr|mlp_bwd" $o/arm${v}_kernel_stats.csv | cut -d, -f1-5
done
NS_BENCH_NGP_CFG=wgrad_after_scatter=1 timeout 300 python -m pytest tests/test_ngp_gpu.py -q -x 2>&1 | tail -3
