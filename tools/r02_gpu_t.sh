cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r02t; mkdir -p $o
timeout 60 python tools/cam_debug.py > $o/cam.log 2>&1; tail -20 $o/cam.log
