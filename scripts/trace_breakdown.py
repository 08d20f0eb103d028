#!/usr/bin/env python3
"""Per-search kernel breakdown from a rocprofv3 --kernel-trace CSV of an IVF bench run: average device time of every
kernel over the last 8 searches (from the coarse stage's query_stats_kernel -- one per search: the list stage reuses its statistics -- to the fallback ivf_scan_kernel), the
span of a search and the sum of its kernel times (the difference is launch gaps).
Usage: python scripts/trace_breakdown.py <..._kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "ivf_select_kernel" in n]
sel = idx[-8:]
agg, spans = collections.OrderedDict(), []
for s in sel:
    i, cnt = s, 0
    while i > 0:
        if "query_stats_kernel" in names[i]:
            cnt += 1
            if cnt == 1:
                break
        i -= 1
    j = s
    while j < len(names) - 1 and "ivf_scan_kernel" not in names[j]:
        j += 1
    busy = 0
    for r in rows[i:j + 1]:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        busy += d
        k = r["Kernel_Name"][:90]
        agg[k] = agg.get(k, 0) + d
    spans.append((int(rows[j]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]), busy, j - i + 1))
for k, v in agg.items():
    print(f"{v / len(sel) / 1e3:8.1f} us  {k}")
print(f"search span {sum(s[0] for s in spans) / len(spans) / 1e3:.1f} us, kernels busy {sum(s[1] for s in spans) / len(spans) / 1e3:.1f} us, "
      f"{sum(s[2] for s in spans) / len(spans):.0f} dispatches")
