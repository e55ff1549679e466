#!/bin/bash
# Usage (GPU box, repo root): bash profiles/vision_pass.sh — kernel trace + SQ counters of the vision_model search
set -e
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_vision
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_vision.py --games 4096 --steps 5 > $OUT/bench_trace.json
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 tools/bench_vision.py --games 4096 --steps 3 > $OUT/bench_sq.json
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_act -- python3 tools/bench_vision.py --games 4096 --steps 3 > $OUT/bench_act.json 2> $OUT/pmc_act.err || echo "act pass failed"
python3 - <<'PY'
import csv, glob, collections
print("## kernel-trace --stats")
for f in glob.glob("gpurun_out/prof_vision/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("|", r["Name"][:70], "|", r["Calls"], "|", r["AverageNs"], "|", r["Percentage"], "|")
for p in ("pmc_sq", "pmc_act"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/prof_vision/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"## --pmc {p} (mean per launch)")
    for k, c in agg.items():
        if "search" in k or "inference" in k:
            for a, v in sorted(c.items()):
                print("|", k, "|", a, "|", round(sum(v) / len(v), 1), "|", len(v), "|")
PY
cat $OUT/bench_trace.json
