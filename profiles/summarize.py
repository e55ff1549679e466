#!/usr/bin/env python3
"""Summarise a gpurun_out/prof_<tag>/ directory written by run_profile.sh into
profiles/<tag>_summary.md (kernel stats + per-launch PMC counters)."""
import csv
import glob
import json
import os
import sys


def rows(pattern):
    out = []
    for f in glob.glob(pattern, recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


def main():
    src, tag = sys.argv[1], sys.argv[2]
    lines = [f"# rocprofv3 summary — {tag}", ""]
    st = rows(os.path.join(src, "trace", "**", "*kernel_stats.csv"))
    lines += ["## kernel-trace --stats", "", "| kernel | calls | total ns | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|---|"]
    for r in st:
        lines.append(f"| {r['Name'][:90]} | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |")
    # working launches of the search kernels: a run through mzs_search_streams adds empty redo launches of the
    # same kernel (a few microseconds each), which the --stats average above mixes in
    tr = rows(os.path.join(src, "trace", "**", "*kernel_trace.csv"))
    per = {}
    for r in tr:
        if "k_search" in r["Kernel_Name"] or "k_inference" in r["Kernel_Name"]:
            per.setdefault(r["Kernel_Name"][:90], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if per:
        lines += ["", "## working launches (duration > half the median of the launches above 20 us; the first launch of a "
                  "process, which loads the code object, excluded when it is an outlier)", "",
                  "| kernel | launches | working | avg ns | median ns | min ns | max ns |", "|---|---|---|---|---|---|---|"]
        for k, v in per.items():
            big = sorted(t for t in v if t > 20000) or sorted(v)
            med = big[len(big) // 2]
            w = sorted(t for t in v if 0.5 * med < t < 3 * med)
            lines.append(f"| {k} | {len(v)} | {len(w)} | {sum(w) / len(w):.0f} | {w[len(w) // 2]} | {w[0]} | {w[-1]} |")
    for name in ("fetch", "write", "sq", "mfma"):
        pm = rows(os.path.join(src, f"pmc_{name}", "**", "*counter_collection.csv"))
        agg = {}
        for r in pm:
            k = (r["Kernel_Name"][:60], r["Counter_Name"])
            a = agg.setdefault(k, [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
        lines += ["", f"## --pmc pass: {name} (mean per launch)", "", "| kernel | counter | mean | launches |", "|---|---|---|---|"]
        for (k, c), (s, n) in sorted(agg.items()):
            lines.append(f"| {k} | {c} | {s / n:.1f} | {n} |")
    for name in ("trace", "fetch", "write", "sq", "mfma"):
        p = os.path.join(src, f"bench_{name}.json")
        if os.path.exists(p):
            txt = open(p).read().strip().splitlines()
            if txt:
                try:
                    j = json.loads(txt[-1])
                    lines += ["", f"bench line under `{name}` pass: value={j['value']:.4g} {j['unit']}, ms_per_step={j['ms_per_step']:.4f}, kernel_ms={j['roofline']['kernel_ms']:.4f}"]
                except Exception:
                    pass
    open(os.path.join("profiles", f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
