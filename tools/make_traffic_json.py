#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes of tools/pmc_probe.py (FETCH_SIZE and
WRITE_SIZE, collected in SEPARATE runs with --kernel-trace only) into the small
JSON bench.py reads for `roofline.traffic`.

    make_traffic_json.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <clips> > profiles/rNN_traffic.json

Units / corrections (MI355X_MICROARCH.md "HBM"): both counters are in KiB; on
gfx950 FETCH_SIZE reports exactly half of a wide coalesced read stream.  The
probe's first dispatch is a 1 GiB device-to-device copy, which calibrates the
read-side factor in the same run (expected ~2.0); WRITE_SIZE is checked against
the 1 GiB normal_() fill (expected ~1.0).
"""
import csv
import glob
import json
import os
import sys

GIB = float(1 << 30)


def rows(d, counter):
    out = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    return sorted(out)


def main():
    base, clips = sys.argv[1], int(sys.argv[2])
    fetch = rows(os.path.join(base, "pmc_FETCH_SIZE"), "FETCH_SIZE")
    write = rows(os.path.join(base, "pmc_WRITE_SIZE"), "WRITE_SIZE")
    cal_r = [v for _, k, v in fetch if "copyBuffer" in k and v > 100000][0]       # 1 GiB read
    cal_w = [v for _, k, v in write if "distribution" in k and v > 100000][0]     # 1 GiB written
    f_read = GIB / (cal_r * 1024.0)
    f_write = GIB / (cal_w * 1024.0)
    fam = lambda k: k.startswith("void wino_gemm_s3_kernel")        # the dominant kernel (bench.py roofline.kernel)
    if not [1 for _, k, _v in fetch if fam(k)]:
        fam = lambda k: k.startswith("void conv_igemm_f32")        # DT_S3=0 runs
    fr = [v for _, k, v in fetch if fam(k)]
    wr = [v for _, k, v in write if fam(k)]
    out = {
        "workload": {"clips": clips, "T": 30, "size": 416},
        "kernel_family": "wino_gemm_s3_kernel" if any(k.startswith("void wino_gemm_s3_kernel") for _, k, _v in fetch) else "conv_igemm_f32",
        "launches_sampled": len(fr),
        "read_calibration_factor": f_read, "write_calibration_factor": f_write,
        "fetch_bytes_per_launch": sum(fr) * 1024.0 * f_read / len(fr),
        "write_bytes_per_launch": sum(wr) * 1024.0 * f_write / len(wr),
        "note": "FETCH_SIZE/WRITE_SIZE (KiB) from separate rocprofv3 --pmc passes over tools/pmc_probe.py; "
                "read side multiplied by the in-run 1 GiB copy calibration (gfx950 FETCH_SIZE counts 64 B per "
                "128 B request); Infinity-Cache hits are included in FETCH_SIZE, so this is traffic beyond L2, "
                "an upper bound on HBM bytes",
    }
    out["traffic_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]
    # the whole conv family of ONE step (the probe runs the path twice: head calibration + the measured step):
    # MFMA GEMMs + Winograd transforms + the direct (round 6) / fused conv_2 / conv_3 / conv_5 kernels (conv_1 is listed separately: it was never part of this sum)
    fam2 = lambda k: any(t in k for t in ("wino_gemm_s3", "conv_igemm_f32", "wino_input", "wino_output", "wino4s_fused", "conv3_h2", "absmax", "splitk_reduce"))
    passes = 2.0
    out["conv_family_fetch_bytes_per_step"] = sum(v for _, k, v in fetch if fam2(k)) * 1024.0 * f_read / passes
    out["conv_family_write_bytes_per_step"] = sum(v for _, k, v in write if fam2(k)) * 1024.0 * f_write / passes
    out["traffic_bytes_per_step"] = out["conv_family_fetch_bytes_per_step"] + out["conv_family_write_bytes_per_step"]
    by = {}
    for _, k, v in fetch:
        if fam2(k):
            by.setdefault(k.split("(")[0][:60], [0.0, 0.0])[0] += v * 1024.0 * f_read / passes
    for _, k, v in write:
        if fam2(k):
            by.setdefault(k.split("(")[0][:60], [0.0, 0.0])[1] += v * 1024.0 * f_write / passes
    out["per_kernel_bytes_per_step"] = {k: {"fetch": a, "write": b} for k, (a, b) in sorted(by.items())}
    # per kernel FAMILY of bench.py's roofline.families (each family's own launches only)
    fams = {"wino_gemm_s3": ("wino_gemm_s3",), "conv_igemm_f32": ("conv_igemm_f32",), "wino4s_fused": ("wino4s_fused",), "conv3_h2": ("conv3_h2",),
            "conv1_mfma": ("conv1_mfma", "conv1_s3"), "wino_transforms": ("wino_input", "wino_output")}
    out["family_bytes_per_step"] = {
        f: (sum(v for _, k, v in fetch if any(t in k for t in pats)) * f_read + sum(v for _, k, v in write if any(t in k for t in pats)) * f_write) * 1024.0 / passes
        for f, pats in fams.items()}
    out["family_launches_per_step"] = {f: sum(1 for _, k, _v in fetch if any(t in k for t in pats)) / passes for f, pats in fams.items()}
    out["conv1_bytes_per_step"] = {"fetch": sum(v for _, k, v in fetch if "conv1_" in k) * 1024.0 * f_read / passes,
                                   "write": sum(v for _, k, v in write if "conv1_" in k) * 1024.0 * f_write / passes}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
