#!/bin/bash
# Usage (GPU box, repo root): bash profiles/vision_pass.sh [games] — kernel trace + SQ / matrix-core counters of the
# vision_model search (default 4096 games: k_search_fused; 2048: k_search_vis4, the head MLPs on the matrix core)
set -e
REPO=$(pwd)
GAMES=${1:-4096}
OUT=$REPO/gpurun_out/prof_vision_$GAMES
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_vision.py --games $GAMES --steps 5 > $OUT/bench_trace.json
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 tools/bench_vision.py --games $GAMES --steps 3 > $OUT/bench_sq.json
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_act -- python3 tools/bench_vision.py --games $GAMES --steps 3 > $OUT/bench_act.json 2> $OUT/pmc_act.err || echo "act pass failed"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma -- python3 tools/bench_vision.py --games $GAMES --steps 3 > $OUT/bench_mfma.json 2> $OUT/pmc_mfma.err || echo "mfma pass failed"
export PROF_DIR=gpurun_out/prof_vision_$GAMES
python3 - <<'PY'
import csv, glob, collections, os
D = os.environ["PROF_DIR"]
print("## kernel-trace --stats")
for f in glob.glob(D + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("|", r["Name"][:70], "|", r["Calls"], "|", r["AverageNs"], "|", r["Percentage"], "|")
for p in ("pmc_sq", "pmc_act", "pmc_mfma"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{D}/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"## --pmc {p} (mean per launch)")
    for k, c in agg.items():
        if "search" in k or "inference" in k:
            for a, v in sorted(c.items()):
                print("|", k, "|", a, "|", round(sum(v) / len(v), 1), "|", len(v), "|")
PY
cat $OUT/bench_trace.json
