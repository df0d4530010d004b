#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace / counter collection) into the
small text tables committed under profiles/.

  rocprof_summary.py stats  <dir>          # per-kernel calls / total / avg from *_kernel_trace.csv
  rocprof_summary.py pmc    <dir> COUNTER  # per-kernel sum and per-launch mean of COUNTER
  rocprof_summary.py mfma   <dir>          # per kernel: effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and the
                                           # MFMA-busy share of its cycles (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles), from ONE
                                           # pass `--kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES`
"""
import csv
import glob
import os
import sys
from collections import OrderedDict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def short(name):
    return name if len(name) <= 90 else name[:87] + "..."


def stats(d):
    rows = []
    for f in find(d, "*kernel_trace.csv"):
        rows += list(csv.DictReader(open(f)))
    agg = OrderedDict()
    for r in rows:
        k = r["Kernel_Name"]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3   # us
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0, r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("LDS_Block_Size", "")])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values()) or 1.0
    print("%-92s %7s %12s %12s %10s %10s %6s %5s %5s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "lds"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-92s %7d %12.1f %12.2f %10.2f %10.2f %6.2f %5s %5s %7s" % (short(k), a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot, a[4], a[5], a[6]))


def pmc(d, counter):
    rows = []
    for f in find(d, "*counter_collection.csv"):
        rows += list(csv.DictReader(open(f)))
    agg = OrderedDict()
    for r in rows:
        if r.get("Counter_Name") != counter:
            continue
        a = agg.setdefault(r["Kernel_Name"], [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
    print("%-92s %7s %18s %18s" % ("kernel", "calls", counter + "_sum", counter + "_per_launch"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-92s %7d %18.1f %18.1f" % (short(k), a[0], a[1], a[1] / a[0]))


def mfma(d):
    dur, name = {}, {}
    for f in find(d, "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3     # us
            name[r["Dispatch_Id"]] = r["Kernel_Name"]
    gui, busy = {}, {}
    for f in find(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                gui[r["Dispatch_Id"]] = gui.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                busy[r["Dispatch_Id"]] = busy.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    agg = OrderedDict()
    for k in dur:
        if k in gui:
            a = agg.setdefault(name[k], [0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += dur[k]; a[2] += gui[k] / 8.0; a[3] += busy.get(k, 0.0) / 1024.0
    print("# GRBM_GUI_ACTIVE is summed over the 8 XCDs (/8 = cycles); SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs (/1024 = busy cycles per SIMD)")
    print("%-92s %7s %12s %10s %12s" % ("kernel", "calls", "total_us", "clock_GHz", "mfma_busy_%"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if a[1] > 0 and a[2] > 0:
            print("%-92s %7d %12.1f %10.3f %12.1f" % (short(k), a[0], a[1], a[2] / a[1] / 1e3, 100.0 * a[3] / a[2]))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "mfma":
        mfma(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3])
