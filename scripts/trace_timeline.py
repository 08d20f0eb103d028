#!/usr/bin/env python3
"""Timeline of the last N dispatches of a rocprofv3 --kernel-trace CSV (start offset, duration, kernel, grid).
Usage: python scripts/trace_timeline.py <..._kernel_trace.csv> [N]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 20):]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} us  {r['Kernel_Name'][:70]}  grid={r['Grid_Size_X']}x{r['Grid_Size_Y']} wg={r['Workgroup_Size_X']}")
