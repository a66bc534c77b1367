"""Timeline of the last call in a rocprofv3 kernel trace (start / end / duration in us, queue, grid, kernel): python tools/trace_timeline.py <kernel_trace.csv> [marker-kernel]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "fbr_kin_kernel"
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
i0, i1 = marks[-2], marks[-1]
t0 = rows[i0]["s"]
for r in rows[i0:i1]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:56]
    print(f"{(r['s'] - t0) / 1e3:9.1f} {(r['e'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:8.1f}  q{r['Queue_Id']:>2} grid {r['Grid_Size_X']:>7} {n}")
