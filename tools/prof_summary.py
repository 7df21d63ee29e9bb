"""Condense a rocprofv3 --kernel-trace --stats CSV directory into a short text summary
(kernel names truncated) suitable for committing under profiles/.
usage: python tools/prof_summary.py <rocprof_out_dir> <out.txt> [title]"""
import csv, glob, os, sys

d, out = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else d
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
with open(out, "w") as fh:
    fh.write(f"# {title}\n# source: rocprofv3 --kernel-trace --stats (kernel_stats.csv), names truncated to 90 chars\n")
    fh.write(f"{'calls':>8} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'total_ms':>10} {'pct':>6}  name\n")
    for r in rows[:40]:
        fh.write(f"{int(r['Calls']):8d} {float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:10.2f} "
                 f"{float(r['MaxNs'])/1e3:10.2f} {float(r['TotalDurationNs'])/1e6:10.3f} "
                 f"{100*float(r['TotalDurationNs'])/tot:6.2f}  {r['Name'][:90]}\n")
print(open(out).read())
