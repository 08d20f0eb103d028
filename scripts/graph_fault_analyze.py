"""Resolve a GPU fault address against a $VDBHIP_ALLOC_LOG allocation log (and the script's TORCH_SEGMENT lines):
which allocation owns it, live or freed, and whether every address a captured graph baked in was live at each launch.
usage: graph_fault_analyze.py alloc.log dbg.log"""
import re, sys
alloc, dbg = sys.argv[1], sys.argv[2]
fault = None
torch_segs = []
for line in open(dbg):
    m = re.search(r"on address (0x[0-9a-f]+)", line)
    if m: fault = int(m.group(1), 16)
    m = re.match(r"TORCH_SEGMENT (0x[0-9a-f]+) (\d+)", line)
    if m: torch_segs.append((int(m.group(1), 16), int(m.group(2))))
live, history, baked, name_of = {}, [], [], {}
lines = [l.rstrip("\n") for l in open(alloc)]
problems = 0
for n, line in enumerate(lines, 1):
    parts = line.split()
    tag, ptr, size = " ".join(parts[:-2]), int(parts[-2], 16) if parts[-2] != "(nil)" else 0, int(parts[-1])
    if tag in ("A", "HA"):
        live[ptr] = (size, n, tag); history.append((ptr, size, n, None, tag))
    elif tag in ("F", "HF"):
        live.pop(ptr, None)
        for i in range(len(history) - 1, -1, -1):
            if history[i][0] == ptr and history[i][3] is None:
                history[i] = history[i][:3] + (n,) + history[i][4:]; break
    elif tag == "GRAPH_INSTANTIATE":
        baked = []
    elif line.startswith("  "):
        if ptr: baked.append((tag.strip(), ptr, size)); name_of[ptr] = tag.strip()
    elif tag == "GRAPH_LAUNCH":
        for nm, p, sz in baked:
            ok = p in live or any(a <= p < a + s for a, s in torch_segs) or nm.startswith("arg.")
            if not ok:
                problems += 1; print(f"line {n}: launch {size}: baked {nm} {p:#x} is NOT live")
print(f"graph launches checked: every baked library buffer live at every launch: {problems == 0}")
if fault is not None:
    print(f"fault address {fault:#x}")
    hit = False
    for ptr, size, a, f, tag in history:
        if ptr <= fault < ptr + max(size, 1):
            hit = True
            print(f"  inside {tag} {ptr:#x} +{size} ({name_of.get(ptr, 'not a buffer the graph names')}): allocated at log line {a}, "
                  + (f"FREED at line {f}" if f else "still live at the fault"))
    for a, s in torch_segs:
        if a <= fault < a + s: hit = True; print(f"  inside torch segment {a:#x} +{s}")
    if not hit:
        near = sorted(history, key=lambda h: min(abs(h[0] - fault), abs(h[0] + h[1] - fault)))[:6]
        print("  inside NO allocation this library ever made (live or freed) and no torch segment; nearest ranges:")
        for ptr, size, a, f, tag in near:
            print(f"    {tag} {ptr:#x} .. {ptr + size:#x} ({name_of.get(ptr, '-')}), alloc line {a}, " + (f"freed line {f}" if f else "live")
                  + f", distance {min(abs(ptr - fault), abs(ptr + size - fault)):#x}")
