#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ngp_gpu.py -x -q -m gpu -k "hash_encode or jacobian or converges or occupancy or renders" 2>&1 | tail -2
for i in 1 2; do
  echo "two items : $(python bench.py --microbench ngp_encode_fwd --reps 50 2>/dev/null | tail -1 | cut -c1-120)"
  echo "one item  : $(NS_ENC_FWD_SINGLE=1 python bench.py --microbench ngp_encode_fwd --reps 50 2>/dev/null | tail -1 | cut -c1-120)"
done
for i in 1 2; do
for v in "" "NS_ENC_FWD_SINGLE=1"; do
  echo "$v: $(env $v NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 800 320 2>&1 | tail -2 | head -1 | cut -c1-40) | $(env $v python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), [round(w['frames_per_s'],1) for w in d['windows']], round(d['sequential']['frames_per_s'],1))")"
done; done
