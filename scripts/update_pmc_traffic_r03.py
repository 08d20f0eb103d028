#!/usr/bin/env python3
"""profiles/pmc_traffic.json entries from the PMC passes of scripts/final_profiles_r03.sh (run on the merged
gpurun_out/final_r03 directory): mean FETCH_SIZE / WRITE_SIZE (KB) of the dominant scan kernel of each workload, corrected as
MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 64 B per 128-B request of wide coalesced streams -> doubled;
WRITE_SIZE as it is).  bench.py copies `hbm_bytes_per_launch` into roofline.traffic of the matching leg."""
import collections, csv, glob, json, os, sys

root = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
WANT = {  # pass group -> (json key, substrings the kernel name must contain; "a|b" = either)
    "sift1m": ("sift1m_i8", ("scan_pair_x16_kernel|scan_i8x16_kernel|scan_i8_kernel",)),
    "ivf8": ("ivf1024_nprobe8", ("scan_i8_kernel", "true")),
    "ivf128": ("ivf1024_nprobe128", ("scan_i8_kernel", "true")),
    "msmarco": ("msmarco_ivf_nprobe32", ("ivf_kloop_scan_kernel",)),
    "ivf32": ("ivf1024_nprobe32", ("scan_i8_kernel", "true")),
    "gaussian": ("gaussian1m", ("scan_x16_kernel<4|scan_kernel<8",)),
    "glove": ("glove1.2m", ("scan_x16_kernel<2|scan_kernel<4",)),
    "marco": ("marco12.5m", ("scan16_kloop_kernel",)),
}
TAG = sys.argv[3] if len(sys.argv) > 3 else "r03"
CORR = ("gfx950: FETCH_SIZE counts 64 B per 128-B request of wide coalesced streams (global_load / LDS-DMA alike) -> doubled; "
        "WRITE_SIZE exact (MI355X_MICROARCH.md, HBM)")
vals = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    group = os.path.relpath(f, root).split(os.sep)[0].replace("pmc_", "").rsplit("_", 1)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            vals[group][r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
table = json.load(open(out)) if os.path.exists(out) else {}
for group, (key, must) in WANT.items():
    best = None
    for kernel, cs in vals.get(group, {}).items():
        if not all(any(alt in kernel for alt in m.split("|")) for m in must) or "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
            continue
        fetch = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
        write = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        if best is None or fetch + write > best[1] + best[2]:
            best = (kernel, fetch, write, len(cs["FETCH_SIZE"]))
    if best is None:
        print("no PMC rows for", group)
        continue
    kernel, fetch, write, launches = best
    keep = {k: v for k, v in table.get(key, {}).items() if k == "algorithmic_bytes_per_launch"}
    table[key] = {"kernel": kernel.replace("void vdb::", "").replace("(vdb::ScanI8Args)", "").replace("(vdb::ScanArgs)", "").replace("(vdb::ScanArgs,vdb::ScanI8Args)", "").replace(" ", ""),
                  "FETCH_SIZE_KB": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1),
                  "hbm_bytes_per_launch": int((2 * fetch + write) * 1024), "launches_averaged": launches, "correction": CORR,
                  "source": f"profiles/{TAG}_pmc_summary.txt (scripts/final_profiles_{TAG}.sh, separate --pmc passes '{group}_fetch' / '{group}_write')",
                  **keep}
    print(key, table[key]["hbm_bytes_per_launch"])
json.dump(table, open(out, "w"), indent=1)
