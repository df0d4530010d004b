"""Builds libmi355_dt.so (HIP, gfx950 only) in-tree with hipcc.

    python -m object_tracking_amd.build        # or: from object_tracking_amd.build import build; build()

hipcc cross-compiles for gfx950 without a GPU present.  The shared library is
written next to this file so that it travels with the source tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmi355_dt.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# decode.hip / targets.hip restate numpy / Python float arithmetic op for op: no FMA contraction there.
SOURCES = {
    "conv_igemm.hip": [],
    "conv1.hip": [],
    "winograd.hip": [],
    # no SLP vectorisation: hipcc otherwise packs the transforms' scalar f32 adds / fmas into v_pk_* with a v_mov per operand --
    # more instructions, and packed f32 issues slowly beside MFMAs (MI355X_MICROARCH.md)
    "wino4s_fused.hip": ["-fno-slp-vectorize"],
    "wino_gemm_s3.hip": [],
    "conv3_h2.hip": [],
    "ingest.hip": [],
    "extract.hip": [],
    "exchange.hip": [],
    "decode.hip": ["-ffp-contract=off"],
    "targets.hip": ["-ffp-contract=off"],
    "recurrent.hip": [],
    "network.hip": [],
}
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
          "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
HEADERS = [os.path.join(CSRC, "dt_internal.h"), os.path.join(HERE, "..", "include", "mi355_dt.h")]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    objs, jobs = [], []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS + [os.path.abspath(__file__)]):
            jobs.append([HIPCC] + COMMON + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
