cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r02u; mkdir -p $o
f() { grep "steps/s\|PSNR\|Error" | sed 's/samples\/step.*loss/loss/;s/after 500 steps: //;s/over 2 training views, mean |depth error| on the object/depth/' | tr '\n' ' '; echo; }
echo "== subset rule"; for i in 1 2 3 4 5 6; do timeout 60 python tools/ngp_bench.py 200 300 2>&1 | f; done
echo "== ngp rule"; for i in 1 2 3 4 5 6; do NS_NGP_GRID_RULE=ngp timeout 60 python tools/ngp_bench.py 200 300 2>&1 | f; done
echo "== extr"; for i in 1 2 3 4; do NS_NGP_EXTRINSICS=1 timeout 60 python tools/ngp_bench.py 200 300 2>&1 | f; done
timeout 300 python -m pytest tests -m gpu -q --timeout=100 -x > $o/pytest.log 2>&1; tail -3 $o/pytest.log
timeout 170 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; tail -c 300 $o/bench.err; head -c 300 $o/bench.json; echo
python - <<PY
import json
d=json.loads(open("$o/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["sequential"]["frames_per_s"], d["extra"]["quality"])
PY
