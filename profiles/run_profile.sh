#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/run_profile.sh <tag> [bench args]
# Writes rocprofv3 kernel-trace stats and separate PMC passes under gpurun_out/prof_<tag>/.
set -e
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline $@"
# the kernel trace runs the benchmark's DEFAULT step counts (200 + 10 warm-up): the first dozens of launches of a
# process run at lower clocks (10 steps: 0.217 ms per search, 200 steps: 0.203 ms), and the judged number is the default run's
TRACE_ARGS="--no-cpu-baseline $@"
cd $REPO
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $TRACE_ARGS > $OUT/bench_trace.json
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_write.json
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_sq.json
# matrix-pipe counters (names differ between ROCm releases: a failed pass is not fatal)
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_mfma -- python3 bench.py $ARGS > $OUT/bench_mfma.json 2> $OUT/pmc_mfma.err || echo "mfma pass failed (see pmc_mfma.err)"
find $OUT -name '*.csv' | head -40
