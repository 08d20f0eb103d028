#!/usr/bin/env python3
"""Per-search kernel breakdown from a rocprofv3 --kernel-trace CSV of a bench run (flat or IVF).

A device-resident search starts with its one memset (__amd_rocclr_fillBufferAligned) and its dispatches follow back to
back, so the trace is cut at every memset; the dispatch sequence that occurs most often is the timed loop's search, and
the average device time of each of its kernels, the span of a search and the sum of its kernel times (the difference is
launch gaps) are printed.  A second argument restricts the choice to sequences that contain a kernel of that name (IVF runs:
`ivf_select` -- the k-means assignment searches of the build outnumber the timed searches).
Usage: python scripts/trace_breakdown.py <..._kernel_trace.csv> [kernel-name-part]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cuts = [i for i, r in enumerate(rows) if "fillBufferAligned" in r["Kernel_Name"]]
cycles = [rows[a:b] for a, b in zip(cuts, cuts[1:])]
must = sys.argv[2] if len(sys.argv) > 2 else ""
cycles = [c for c in cycles if any(must in r["Kernel_Name"] for r in c)]
sig = collections.Counter(tuple(r["Kernel_Name"] for r in c) for c in cycles)
if not sig:
    sys.exit("no searches found")
best, count = sig.most_common(1)[0]
sel = [c for c in cycles if tuple(r["Kernel_Name"] for r in c) == best]
n = len(sel)
dur = [sum(int(c[j]["End_Timestamp"]) - int(c[j]["Start_Timestamp"]) for c in sel) / n / 1e3 for j in range(len(best))]
for name, d in zip(best, dur):
    print(f"{d:8.1f} us  {name[:110]}")
span = sum(int(c[-1]["End_Timestamp"]) - int(c[0]["Start_Timestamp"]) for c in sel) / n / 1e3
print(f"search span {span:.1f} us, kernels busy {sum(dur):.1f} us, {len(best)} dispatches (mean of {n} searches with this dispatch sequence)")
