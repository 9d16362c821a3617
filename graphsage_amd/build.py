"""Build the gfx950 shared library in-tree with hipcc (cross-compiles without a GPU).

    python -m graphsage_amd.build [--force] [--verbose]

Output: graphsage_amd/_C/libgraphsage_amd.so   (git-ignored; travels to the GPU box via gpurun)
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_C")
LIB_PATH = os.path.join(OUT_DIR, "libgraphsage_amd.so")
STAMP = os.path.join(OUT_DIR, "build.stamp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast", "-fno-finite-math-only"] + os.environ.get("HIPCC_EXTRA", "").split()


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    incl = os.path.join(HERE, "..", "include", "graphsage_amd.h")
    for p in _sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [incl]:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode())      # names, not paths: the digest must not depend on where the repo lies
            h.update(f.read())                           # (gpurun unpacks it under a scratch path on the GPU box)
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    dig = _digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == dig:
                return LIB_PATH
    if not os.path.exists(HIPCC):
        if os.path.exists(LIB_PATH):
            # GPU box without a matching source digest but with a prebuilt library: use it (its ABI version is checked
            # against the Python binding in _lib.load()).
            sys.stderr.write("graphsage_amd: hipcc not found; using the prebuilt %s although its source digest differs "
                             "from the checked-out sources\n" % LIB_PATH)
            return LIB_PATH
        raise RuntimeError("hipcc not found at %s and no prebuilt %s" % (HIPCC, LIB_PATH))
    objs = []
    procs = []
    for src in _sources():
        obj = os.path.join(OUT_DIR, os.path.basename(src).replace(".hip", ".o"))
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed for %s:\n%s\n" % (src, out.decode(errors="replace")))
        elif verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("hipcc compilation failed")
    cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB_PATH


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(p)
