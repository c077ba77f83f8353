#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_corr_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
  echo "ahead : $(python bench.py --microbench corr_lookup_enc --reps 50 2>/dev/null | tail -1 | cut -c1-110)"
  echo "serial: $(NS_LOOKUP_ENC_SERIAL=1 python bench.py --microbench corr_lookup_enc --reps 50 2>/dev/null | tail -1 | cut -c1-110)"
done
echo "coop  : $(python bench.py --microbench corr_lookup_coop --reps 50 2>/dev/null | tail -1 | cut -c1-110)"
