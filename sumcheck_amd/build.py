"""Build libsumcheck_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU.

Two libraries come out of the same sources:
  libsumcheck_hip.so      the product: one production path (product-tree kernel, carry-free arithmetic, F29 bound tables) plus the
                          generic fallback for very long products;
  libsumcheck_hip_exp.so  -DSC_EXPERIMENTS: additionally the cross-check kernels (saturated Comba arithmetic, node-by-node, LDS-tiled)
                          and the environment knobs that select them.  Loaded only by tests/test_gpu_variants.py (SC_LIB_VARIANT=exp).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libsumcheck_hip.so")
OUT_EXP = os.path.join(HERE, "libsumcheck_hip_exp.so")
SOURCES = ["kernels_big.hip", "kernels.hip", "kernels_tail.hip", "kernels_wide.hip", "kernels_wide16.hip", "gkr.hip", "abi.hip", "protocol.hip", "comm.hip"]
HEADERS = ["fr_device.hpp", "fe_device.hpp", "kernel_common.hpp", "finalize_device.hpp", "fe_mad_chain.inc", "fr_mac.inc", "fr_mul_gen.inc", "kernels.h", "host_fr.hpp", "transcript.hpp", "prover_internal.hpp", "load_factor.hpp", "wide_tree.hpp", os.path.join("..", "..", "include", "sumcheck_hip.h")]
EXTRA = os.environ.get("SC_BUILD_EXTRA", "").split()  # e.g. SC_BUILD_EXTRA="-DSC_TAIL_CLOCKS" for a one-off local build
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _stale(out: str) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _obj_stale(obj: str, src: str, cmd: list) -> bool:
    """per object: recompile only when the source, a header the compiler saw it include (the -MD dependency file of the last
    compile) or this script is newer than the object, or when the object was compiled with ANOTHER command line (obj.cmd: a one-off
    SC_BUILD_EXTRA=-DSC_TAIL_CLOCKS build must not leave its instrumented objects to the next plain build) -- kernels.hip alone is 2.5 minutes"""
    dep = obj + ".d"
    if not (os.path.exists(obj) and os.path.exists(dep)):
        return True
    try:
        if open(obj + ".cmd").read() != " ".join(cmd):
            return True
    except OSError:
        return True
    t = os.path.getmtime(obj)
    try:
        words = open(dep).read().replace("\\\n", " ").split()
    except OSError:
        return True
    files = [w for w in words[1:] if not w.endswith(":")] + [src, os.path.abspath(__file__)]
    return any((not os.path.exists(f)) or os.path.getmtime(f) > t for f in files if not f.startswith("/opt/rocm"))


def _start(experiments: bool, verbose: bool, force: bool = False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bdir = os.path.join(HERE, "build_exp" if experiments else "build")
    os.makedirs(bdir, exist_ok=True)
    procs, objs = [], []
    stale_any = False
    for s in SOURCES:
        o = os.path.join(bdir, s + ".o")
        objs.append(o)
        src = os.path.join(CSRC, s)
        # -Rpass-analysis: registers / scratch / LDS / occupancy of every kernel, for free with every build (tools/kernel_resources.py prints them)
        cmd = [hipcc] + FLAGS + (["-DSC_EXPERIMENTS"] if experiments else []) + EXTRA + ["-Rpass-analysis=kernel-resource-usage", "-MD", "-MF", o + ".d", "-c", src, "-o", o]
        if not force and not _obj_stale(o, src, cmd):
            continue
        stale_any = True
        if os.path.exists(o + ".cmd"):
            os.remove(o + ".cmd")  # (written again once the compile has succeeded: _finish)
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, o, subprocess.Popen(cmd, stderr=open(o + ".log", "w"))))
    return hipcc, procs, objs


def _finish(hipcc, procs, objs, out):
    for cmd, o, p in procs:
        if p.wait() != 0:
            try:
                sys.stderr.write("".join(l for l in open(o + ".log") if "remark:" not in l and "-Rpass" not in l)[-8000:])
            except OSError:
                pass
            for f in (o, o + ".d"):  # never leave a half-written object behind
                if os.path.exists(f):
                    os.remove(f)
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        with open(o + ".cmd", "w") as f:
            f.write(" ".join(cmd))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    return out


def _cmd_mismatch(experiments: bool) -> bool:
    """an object of this build directory was compiled with another command line than this run would use (SC_BUILD_EXTRA changed)"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bdir = os.path.join(HERE, "build_exp" if experiments else "build")
    for s in SOURCES:
        o = os.path.join(bdir, s + ".o")
        cmd = [hipcc] + FLAGS + (["-DSC_EXPERIMENTS"] if experiments else []) + EXTRA + ["-Rpass-analysis=kernel-resource-usage", "-MD", "-MF", o + ".d", "-c", os.path.join(CSRC, s), "-o", o]
        try:
            if open(o + ".cmd").read() != " ".join(cmd):
                return True
        except OSError:
            # no record: an object built before records existed (or no object at all, e.g. a shipped .so without its build directory)
            if os.path.exists(o):
                return True
    return False


def build(force: bool = False, verbose: bool = False, experiments: bool = False) -> str:
    out = OUT_EXP if experiments else OUT
    if not force and not _stale(out) and not _cmd_mismatch(experiments):
        return out
    return _finish(*_start(experiments, verbose, force), out)


def build_all(force: bool = False, verbose: bool = False):
    """both libraries, compiled concurrently"""
    jobs = []
    for exp, out in ((False, OUT), (True, OUT_EXP)):
        if force or _stale(out) or _cmd_mismatch(exp):
            jobs.append((_start(exp, verbose, force), out))
    for st, out in jobs:
        _finish(*st, out)
    return OUT, OUT_EXP


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
