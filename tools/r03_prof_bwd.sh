cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r03_bwd; mkdir -p $o
python -m pytest tests/test_ngp_gpu.py -m gpu -q -x -k "fused" 2>&1 | tail -5 > $o/test.log
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/prof -o bwd -- python tools/ngp_bwd_bench.py 1.0 > $o/bench.log 2>&1
cat $o/test.log $o/bench.log
python - <<PY
import csv, glob
for f in glob.glob("$o/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us")
PY
