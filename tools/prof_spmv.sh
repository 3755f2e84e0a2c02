#!/bin/bash
# usage: tools/prof_spmv.sh <tag>   (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1
O=gpurun_out/prof_$T
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o run -- python3 tools/spmv_exp2.py > $O/trace.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_l2 -o run -- python3 tools/spmv_exp2.py > $O/pmc_l2.log 2>&1
rocprofv3 --pmc SQ_WAVES_sum SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq -o run -- python3 tools/spmv_exp2.py > $O/pmc_sq.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE -d $O/pmc_ta -o run -- python3 tools/spmv_exp2.py > $O/pmc_ta.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o run -- python3 tools/spmv_exp2.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o run -- python3 tools/spmv_exp2.py > $O/pmc_write.log 2>&1
find $O -name "*.csv" | head -30
