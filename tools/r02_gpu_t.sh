cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r02t; mkdir -p $o
timeout 200 python tools/c1280_debug.py > $o/c1280_debug.log 2>&1; grep -v "Gloo\|dist-packages\|^$" $o/c1280_debug.log | tail -40 | cut -c1-220
