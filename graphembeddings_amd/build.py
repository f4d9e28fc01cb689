"""Builds libge_hip.so (the C-ABI HIP library, include/ge_hip.h) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the built .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libge_hip.so")
SOURCES = ["ge_capi.hip", "ge_complex.hip", "ge_rows.hip", "ge_hole.hip", "ge_1vk.hip", "ge_train.hip", "ge_prep_big.hip", "ge_shard.hip", "ge_spectral.hip", "ge_rank.hip", "ge_rank_pipe.hip", "ge_rank_f16.hip", "ge_known.hip"]
HEADERS = ["ge_common.h", "ge_prep.h", "ge_complex_dev.h", "ge_rank_dev.h", os.path.join("..", "..", "include", "ge_hip.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libge_hip.so cannot be built (ROCm toolchain required)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(p) > t for p in deps)


# per-source extra flags.  ge_hole.hip writes its packed fp32 math (v_pk_fma_f32) with explicit
# 2-vectors; the SLP vectoriser would otherwise re-pair the scalar correlation into pairs that are
# unaligned in LDS and re-read them with bank-conflicting ds_read2_b32.
FILE_FLAGS = {"ge_hole.hip": ["-fno-slp-vectorize"]}
OBJDIR = os.path.join(HERE, "_obj")


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    """Compile every HIP source (one object each, in parallel) and link graphembeddings_amd/libge_hip.so."""
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    cc = _hipcc()
    common = ["-O3", f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", *extra_flags,
              *os.environ.get("GE_CXXFLAGS", "").split()]
    procs = []
    for src in SOURCES:
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        cmd = [cc, *common, *FILE_FLAGS.get(src, []), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, obj, subprocess.Popen(cmd, cwd=CSRC)))
    objs = []
    for cmd, obj, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
        objs.append(obj)
    link = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB + ".tmp", *objs]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
