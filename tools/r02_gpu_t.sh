cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r02t; mkdir -p $o
timeout 300 python tools/zero_debug2.py 0 25 > $o/ze0.log 2>&1; echo "eager rc $? reps $(grep -c ' ok' $o/ze0.log)"
timeout 300 python tools/zero_debug2.py 1 25 > $o/ze1.log 2>&1; echo "graph rc $? reps $(grep -c ' ok' $o/ze1.log)"
for i in 1 2; do timeout 300 python -m pytest tests -m gpu -q --timeout=100 -x > $o/pytest_$i.log 2>&1; echo "suite $i rc $?"; tail -1 $o/pytest_$i.log | cut -c1-100; done
