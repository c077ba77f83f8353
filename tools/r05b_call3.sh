#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05b3; mkdir -p $o
timeout 200 python tools/small_conv_bench.py > $o/small_conv.txt 2>&1; tail -16 $o/small_conv.txt | head -15
