#!/usr/bin/env python3
"""Per-search kernel breakdown of the flat bench from a rocprofv3 --kernel-trace CSV: average device time of every kernel
over the last 8 searches of the timed loop (query_stats_kernel ... merge_kernel), span and busy time.
Usage: python scripts/trace_flat.py <..._kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "refine_tail_kernel" in n or "refine_list_kernel" in n]
sel = idx[-10:-2]
agg, spans = collections.OrderedDict(), []
for s in sel:
    i = s
    while i > 0 and "query_stats_kernel" not in names[i]:
        i -= 1
    while i > 0 and "fillBuffer" in names[i - 1]:
        i -= 1
    j = s
    while j < len(names) - 1 and "merge_kernel" not in names[j]:
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
