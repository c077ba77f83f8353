cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r02t; mkdir -p $o
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; tail -3 $o/smoke.log
timeout 200 python -m pytest tests/test_bench_pipeline_gpu.py tests/test_ngp_gpu.py -m gpu -q -x --timeout=150 -k "multi_gpu or ingest" > $o/pytest.log 2>&1; tail -8 $o/pytest.log
