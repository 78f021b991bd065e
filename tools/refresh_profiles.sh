#!/bin/bash
# Regenerate the judged artefacts under profiles/ on a GPU box (run from the repo root through gpurun):
#   bench line (default bench.py), rocprofv3 kernel trace + stats of the same workload, per-layer conv table,
#   HBM traffic per conv launch from two separate PMC passes (FETCH_SIZE, WRITE_SIZE; full 512-frame groups), SQ counters.
#   tools/refresh_profiles.sh [tag]      tag (default r02) prefixes every file copied into profiles/
set -e
R=$PWD
TAG=${1:-r05}
O=$R/gpurun_out/refresh_$TAG
rm -rf $O && mkdir -p $O
B="--cpu-frames 0 --no-curve --no-own --no-plugin"     # the traces describe the headline workload only
cd /tmp && export TMPDIR=/tmp
# the bench command itself (from-host span, default = two streams) under the kernel trace: the UNION of its conv kernels' intervals must agree
# with the HIP-event figure of the bench line (roofline.kernel_ms_per_step); tools/prof_layers.py prints it
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py $B > $O/bench_under_rocprof.json 2> $O/trace.log
# the same workload on ONE stream: per-launch durations that describe the kernels (nothing beside them) -> the per-layer table
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1 -- python3 $R/bench.py $B --single-stream > $O/bench_single_stream_under_rocprof.json 2> $O/trace1.log
# counters: separate passes, full launch groups only, clip resident (the conv launches are the same)
AICAM_NO_TAPER=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --resident --single-stream $B > /dev/null 2> $O/fetch.log
AICAM_NO_TAPER=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 1 --resident --single-stream $B > /dev/null 2> $O/write.log
AICAM_NO_TAPER=1 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/sqa -- python3 $R/bench.py --steps 1 --warmup 1 --resident --single-stream $B > /dev/null 2> $O/sqa.log
AICAM_NO_TAPER=1 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/sqb -- python3 $R/bench.py --steps 1 --warmup 1 --resident --single-stream $B > /dev/null 2> $O/sqb.log
cd $R
python tools/pmc_sq.py $O/sqa $O/sqb 40 > $O/sq_counters.txt
T=$(ls -d $O/trace/*/ | head -1)
T1=$(ls -d $O/trace1/*/ | head -1)
python tools/prof_layers.py $T1 512 15360 100 > $O/conv_layers.txt                 # per-layer table: the one-stream trace
python tools/prof_layers.py $T 512 15360 0 2> /dev/null | head -30 > $O/kernel_summary.txt || true     # kernel totals + the conv class's union: the bench command's own trace
cp $(ls $T/*_kernel_stats.csv | head -1) $O/kernel_stats.csv
cp $(ls $T1/*_kernel_stats.csv | head -1) $O/kernel_stats_single_stream.csv
python tools/pmc_traffic.py $O/fetch $O/write "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1 --resident with AICAM_NO_TAPER=1 (full 512-frame launch groups only); FETCH_SIZE x2 (gfx950 correction of MI355X_MICROARCH.md)" $TAG > $O/pmc_traffic.txt
# the bench line last: its roofline.traffic reads the profiles/pmc_traffic.json written just above
python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-300
rm -rf $O/trace $O/trace1 $O/fetch $O/write $O/sqa $O/sqb          # the raw traces stay on the box (tens of MB); the summaries travel back
for f in bench.json bench_under_rocprof.json bench_single_stream_under_rocprof.json kernel_stats.csv kernel_stats_single_stream.csv kernel_summary.txt conv_layers.txt sq_counters.txt; do cp $O/$f profiles/${TAG}_$f; done
# profiles/ of the GPU box does not travel back, gpurun_out/ does: stage the judged files there as well.
# Afterwards, in the container:  cp gpurun_out/profiles_$TAG/* profiles/
mkdir -p $R/gpurun_out/profiles_$TAG
cp profiles/${TAG}_* profiles/pmc_traffic.json $R/gpurun_out/profiles_$TAG/
ls -la $O
