#!/usr/bin/env python3
"""Summary of a rocprofv3 kernel trace (rocpd sqlite output) of tools/bench_train.py --only-vision: per case (batch
32 / 128 / 512, unroll 5) one steady-state training step — launches, busy time per stream, the top kernels.

  rocprofv3 --kernel-trace --stats -d gpurun_out/vt -o vt -- python3 tools/bench_train.py --only-vision --steps 20
  python3 profiles/summarize_train_db.py gpurun_out/vt/vt_results.db > profiles/r3_vision_train_summary.md
"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall()
fin = [i for i, r in enumerate(rows) if "k_vt_finalize" in r[0]]
steps = 25                                   # 5 warm-up + 20 timed steps per case
print("# r3 — vision_model training step, kernel trace (rocprofv3 --kernel-trace, rocpd database)\n")
print("`tools/bench_train.py --only-vision --steps 20`: one steady-state step per case (the 11th of 25), unroll 5.\n"
      "Two streams: the representation + dynamics chain on the caller's (stream 0), the six prediction passes beside it.\n"
      "Wall times of the same command without the profiler are in DESIGN.md §5 (the profiler widens the gaps between launches).\n")
for case, B in enumerate((32, 128, 512)):
    a, b = fin[case * steps + 10], fin[case * steps + 11]
    seg = rows[a + 1:b + 1]
    per_stream = defaultdict(lambda: [0, 0.0])
    per_kernel = defaultdict(lambda: [0, 0.0])
    for name, s, e, st in seg:
        k = name.split("::")[-1].split("(")[0] if "k_vt_finalize" not in name else "k_vt_finalize"
        per_stream[0 if st == 0 else 1][0] += 1
        per_stream[0 if st == 0 else 1][1] += (e - s) / 1e3
        per_kernel[k][0] += 1
        per_kernel[k][1] += (e - s) / 1e3
    print(f"## batch {B}\n")
    print(f"launches {len(seg)}; busy time main stream {per_stream[0][1]:.0f} us ({per_stream[0][0]} launches), "
          f"side stream {per_stream[1][1]:.0f} us ({per_stream[1][0]} launches)\n")
    print("| kernel | launches | total us | average us |\n|---|---|---|---|")
    for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"| `{k}` | {v[0]} | {v[1]:.1f} | {v[1] / v[0]:.2f} |")
    print()
