"""Per-launch-shape durations from a rocprofv3 --kernel-trace CSV directory (kernels of this library only):
python tools/trace_shapes.py <rocprof_out_dir>"""
import collections, csv, glob, os, sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "GLOBAL__N" in n:
        n = n.split("GLOBAL__N_1")[1][2:44]
        agg[(n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k[0]:44s} grid {k[1]:>7s} {k[2]:>2s} {k[3]:>2s}  n={len(v):5d} mean={sum(v)/len(v):8.2f} med={v2[len(v)//2]:8.2f} us")
