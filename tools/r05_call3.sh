#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05c; mkdir -p $o
timeout 400 python tools/r05_pipeline_diag.py 20 > $o/diag.jsonl 2> $o/diag.err; cat $o/diag.jsonl; tail -3 $o/diag.err
