#!/bin/bash
# Usage (GPU box, repo root): bash profiles/icache_pass.sh  — instruction-cache counters of the search kernels
# (one rocprofv3 --pmc pass per benchmark; summaries by profiles/summarize.py-style grep afterwards)
set -e
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_icache
mkdir -p $OUT
export TMPDIR=/tmp
C="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/tag123 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/tag123.json
timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/tag127 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --weights tag127 > $OUT/tag127.json
timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/vision -- python3 tools/bench_vision.py --steps 3 > $OUT/vision.json
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT", "gpurun_out/prof_icache")
for tag in ("tag123", "tag127", "vision"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"gpurun_out/prof_icache/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQC_ICACHE_REQ": n[k] += 1
    for k, c in agg.items():
        if "search" in k:
            d = max(n[k], 1)
            print(tag, k, "launches", n[k], {a: round(b / d) for a, b in c.items()})
PY
