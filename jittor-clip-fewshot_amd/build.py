"""Builds libclipfs_hip.so (gfx950) in-tree with hipcc.  Usage: python build.py [--force]

The library is the product's only compute path: the Python host side (clipfs/_lib.py) refuses to run
without it.  Object files are cached under csrc/build/ and rebuilt when a source or header changes.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OUT = os.path.join(HERE, "clipfs", "libclipfs_hip.so")
SOURCES = ["common.hip", "gemm.hip", "gemm_bf16.hip", "gemm_f16.hip", "norm.hip", "attention.hip", "attention_f16.hip", "attention_mfma.hip", "attention_mfma16.hip", "lora.hip", "lora_mfma.hip", "elem.hip", "mta.hip", "views.hip", "resnet.hip", "bpe.hip", "tower.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC, "-Wall",
         "-Wno-unused-function", "-ffp-contract=off"] + os.environ.get("HIPCC_EXTRA", "").split()


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _stamp(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def source_stamps():
    """(whole-library stamp, fp32-GEMM stamp) of the sources in the tree -- what clipfs_source_stamp() /
    clipfs_gemm_source_stamp() of a library built from them return."""
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(INCLUDE, "clipfs.h")]
    # common.hip only carries the stamps themselves: leave it out so that a stamp does not depend on itself
    srcs = [os.path.join(CSRC, s) for s in SOURCES if s != "common.hip" and os.path.exists(os.path.join(CSRC, s))]
    return (_stamp(sorted(srcs) + headers),
            _stamp([os.path.join(CSRC, "gemm.hip"), os.path.join(CSRC, "gemm_common.h"), os.path.join(CSRC, "common.h")]))


def build(force: bool = False, verbose: bool = True, tag: str = "") -> str:
    """``tag``: an EXPERIMENT build (extra flags from $HIPCC_EXTRA) kept apart from the product library: objects under
    csrc/build/<tag>/, output clipfs/libclipfs_hip_<tag>.so, loaded only when CLIPFS_LIB_TAG=<tag> is set."""
    bdir = os.path.join(CSRC, "build", tag) if tag else os.path.join(CSRC, "build")
    out = OUT.replace(".so", f"_{tag}.so") if tag else OUT
    os.makedirs(bdir, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(INCLUDE, "clipfs.h")]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    lib_stamp, gemm_stamp = source_stamps()
    stamp_file = os.path.join(bdir, "stamp.txt")
    stamp_now = lib_stamp + " " + gemm_stamp
    stamp_old = open(stamp_file).read().strip() if os.path.exists(stamp_file) else ""
    objs, jobs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(bdir, s.replace(".hip", ".o"))
        objs.append(obj)
        extra = []
        if s == "common.hip":  # carries the source stamps: rebuilt whenever any source changed
            extra = [f'-DCLIPFS_SOURCE_STAMP="{lib_stamp}"', f'-DCLIPFS_GEMM_STAMP="{gemm_stamp}"']
        if force or _newer(obj, [src] + headers) or (extra and stamp_now != stamp_old):
            jobs.append([HIPCC, *FLAGS, *extra, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _newer(out, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
    with open(stamp_file, "w") as f:
        f.write(stamp_now + "\n")
    return out


if __name__ == "__main__":
    tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else ""
    print(build(force="--force" in sys.argv, tag=tag))
