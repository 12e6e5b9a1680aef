"""Builds the in-tree native library: hand-written HIP kernels + C ABI for gfx950.

    python -m mcbrat3d_amd.build

hipcc cross-compiles without a GPU.  The .so stays in-tree (git-ignored) so that it
travels with the source snapshot to the GPU box."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmcbrat_hip.so")
SOURCES = ["mcbrat_api.hip", "mcbrat_host.cpp"]
DEPS = SOURCES + ["mcbrat_kernels.hip", "mcbrat_blockwalk.hip", "mcbrat_device.h", os.path.join("..", "..", "include", "mcbrat.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]


def sources_sha256():
    """sha256 over the sources the library is built from (DEPS, in order): stored with every counter record
    (scripts/pmc_summary.py -> profiles/pmc_shipped.json) so that bench.py can tell counters of these kernels from stale ones."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X library cannot be built")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False, extra_flags=(), out=None):
    out = out or LIB
    if not force and out == LIB and not stale():
        return LIB
    cmd = [hipcc()] + FLAGS + list(extra_flags) + os.environ.get("MCBRAT_EXTRA_FLAGS", "").split() + ["-o", out] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
