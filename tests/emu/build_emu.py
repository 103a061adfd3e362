"""Build the CPU-emulated kernel library (test harness): the unchanged csrc/*.hip sources compiled
with g++ against tests/emu/dlpd_platform.h.  Output: tests/emu/libdlpd_emu.so"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.abspath(os.path.join(HERE, "..", "..", "deeplocalproteindocking_amd", "csrc"))
OUT = os.path.join(HERE, "libdlpd_emu.so")
OUT_PRODUCT = os.path.join(HERE, "libdlpd_emu_product.so")      # without -DDLPD_TEST_VARIANTS: what libdlpd.so ships
SRCS = ["dlpd_corr.hip", "dlpd_k2.hip", "dlpd_k2q.hip", "dlpd_k3r.hip", "dlpd_k1r.hip", "dlpd_topk.hip", "dlpd_generic.hip", "dlpd_atoms.hip", "dlpd_conv.hip", "dlpd_version.hip"]


def _fresh(deps, out=OUT):
    return os.path.exists(out) and all(os.path.getmtime(out) > os.path.getmtime(d) for d in deps)


def build(force=False, variants=True):
    """variants=False: the translation units as the PRODUCT library compiles them (no test-variant kernels, the
    dispatch tables and ``dlpd_orientation_supported`` of libdlpd.so) -- for the CPU tests of the engine's gating."""
    import fcntl
    OUT = globals()["OUT"] if variants else OUT_PRODUCT
    srcs = [os.path.join(CSRC, s) for s in SRCS]
    deps = srcs + [os.path.join(CSRC, h) for h in ("dlpd_fft.h", "dlpd_internal.h", "dlpd_k1.h", "dlpd_k3.h")] + \
        [os.path.join(HERE, "dlpd_platform.h")]
    if not force and _fresh(deps, OUT):
        return OUT
    # the ranks of a multi-process test must not compile concurrently (and never load a half-written file):
    # one builder under a lock, output moved into place atomically
    with open(OUT + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or not _fresh(deps, OUT):
            tmp = OUT + ".tmp.%d" % os.getpid()
            cmd = ["g++", "-O2", "-g", "-std=c++17", "-shared", "-fPIC", "-fpermissive", "-w"] + \
                  (["-DDLPD_TEST_VARIANTS"] if variants else []) + ["-I", HERE, "-I", CSRC, "-o", tmp] + os.environ.get("DLPD_EMU_FLAGS", "").split()
            for s in srcs:
                cmd += ["-x", "c++", s]
            subprocess.check_call(cmd)
            os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="-f" in sys.argv))
