#!/bin/bash
# SQ counters of the headline workload's kernels (the two counter passes of tools/refresh_profiles.sh alone):  gpurun -- bash tools/sq_now.sh [pattern]
R=$PWD; O=$R/gpurun_out/sq_now; rm -rf $O; mkdir -p $O
B="--cpu-frames 0 --no-curve --no-own --no-plugin"
cd /tmp && export TMPDIR=/tmp
AICAM_NO_TAPER=1 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/sqa -- python3 $R/bench.py --steps 1 --warmup 1 --resident --single-stream $B > /dev/null 2> $O/sqa.log
AICAM_NO_TAPER=1 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/sqb -- python3 $R/bench.py --steps 1 --warmup 1 --resident --single-stream $B > /dev/null 2> $O/sqb.log
cd $R
python tools/pmc_sq.py $O/sqa $O/sqb 60 > $O/sq_counters.txt
rm -rf $O/sqa $O/sqb
head -1 $O/sq_counters.txt; grep -i "${1:-c32s2}" $O/sq_counters.txt
