#!/bin/bash
# Usage (GPU box, repo root): bash profiles/trace_pass.sh <tag> [bench args] — kernel trace of the default bench run plus the
# matrix-core / issue counters of a short one, under gpurun_out/prof_<tag>/ (profiles/summarize.py turns it into a summary)
set -e
TAG=$1; shift
OUT=$(pwd)/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline "$@" > $OUT/bench_trace.json
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_mfma -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_mfma.json 2> $OUT/pmc_mfma.err || echo "mfma pass failed"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_sq.json
