#!/bin/bash
# counters of the c1280-scale BA kernels: separate rocprofv3 passes (kernel trace; FETCH_SIZE; WRITE_SIZE + L2; SQ split; MFMA / clock)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r06pmc${1:-}; mkdir -p $o; N=8   # $1: suffix of the output directory, $2: "c640" for the tracking-window size
timeout 300 python tools/ba_c1280_bench.py $N $2 > $o/bench.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace -f csv -d $o/trace -o t -- python tools/ba_c1280_bench.py $N $2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $o/fetch -o f -- python tools/ba_c1280_bench.py $N $2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $o/write -o w -- python tools/ba_c1280_bench.py $N $2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -f csv -d $o/sq -o s -- python tools/ba_c1280_bench.py $N $2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -f csv -d $o/sq2 -o s -- python tools/ba_c1280_bench.py $N $2 > /dev/null 2>&1
python tools/ba_pmc.py $o $N $o/bench.json $o/ba_pmc.json
rm -rf $o/trace $o/fetch $o/write $o/sq $o/sq2
