"""Condensed view of a kernel's memory instructions and waits in hipcc's -S output: one token per instruction, in order
(L = global/buffer load, S = store, D = LDS-DMA load, w<N> = s_waitcnt vmcnt(N), B = s_barrier, | = label/branch) --
shows at a glance where loads are waited for one at a time or stores are drained (a w0 right before / after every S or L)."""
import re, sys
path, name = sys.argv[1], sys.argv[2]
out = []; on = False
for ln in open(path):
    if re.match(r"^_Z\w*:", ln):
        on = name in ln
        if on: out.append("\n== " + ln.split(":")[0] + "\n")
        continue
    if not on: continue
    t = ln.strip()
    if t.startswith("s_endpgm"): on = False; continue
    m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
    if m: out.append("w%s " % m.group(1)); continue
    if re.match(r"(global|buffer|flat)_load\w* .*lds", t) or "load_lds" in t: out.append("D "); continue
    if re.match(r"(global|buffer|flat)_load", t): out.append("L "); continue
    if re.match(r"(global|buffer|flat)_store", t): out.append("S "); continue
    if t.startswith("s_barrier"): out.append("B "); continue
    if t.startswith(".LBB"): out.append("| "); continue
    if t.startswith("s_cbranch"): out.append("br "); continue
print("".join(out))
