"""Print the device timeline of a rocprofv3 --kernel-trace CSV (development aid): start / end (ms, relative to the first kernel) and
duration of every kernel dispatch, in start order.  usage: timeline.py <dir with *_kernel_trace.csv> [min_us]"""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
t0 = rows[0][0]
minus = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
for s, e, name, q, st in rows:
    if (e - s) / 1e3 < minus:
        continue
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
    print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e3:9.1f} us  q{q:>3} s{st:>3}  {short}")
