"""Build libopenrec_hip.so (hipcc, gfx950 only) in-tree: openrec_amd/_lib/.

    python -m openrec_amd.build [--force]

The .so is git-ignored but travels with the gpurun snapshot.  Each .hip file
is compiled to an object in parallel and linked with hipcc.
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_lib")
LIB = os.path.join(OUT, "libopenrec_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics",
            "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value",
            "-Wno-bitwise-instead-of-logical"]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return _sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(HERE, "..", "include", "*.h")) + [os.path.abspath(__file__)]


def source_hash():
    """sha256 (16 hex digits) of the sources libopenrec_hip.so is built from: profiles/*_traffic.json carries it, so a PMC
    figure is only ever reported next to the build it was measured on"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(_sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


HASH_FILE = LIB + ".hash"      # source_hash() of the sources the .so was built from (mtimes say nothing on a shipped snapshot)


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(HASH_FILE):
        return True
    try:
        return open(HASH_FILE).read().strip() != source_hash()
    except OSError:
        return True


def _compile(src):
    obj = os.path.join(OUT, os.path.basename(src)[:-4] + ".o")
    hdr_t = max(os.path.getmtime(f) for f in _deps() if not f.endswith(".hip"))
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_t):
        return obj
    cmd = [HIPCC] + CXXFLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=False):
    os.makedirs(OUT, exist_ok=True)
    if not force and not needs_build():
        return LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found at {HIPCC}; cannot build {LIB}")
    if force:
        for o in glob.glob(os.path.join(OUT, "*.o")):
            os.remove(o)
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    with open(HASH_FILE, "w") as f:
        f.write(source_hash() + "\n")
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
