"""Builds an experimental variant of the library next to the product one: graphembeddings_amd/_variants/libge_<tag>.so,
compiled with extra -D flags (experiments are #ifdef GE_EXP_* blocks that never ship enabled).  Only the sources named
are recompiled; the other objects come from graphembeddings_amd/_obj (build the product library first).
usage: python tools/dev/build_variant.py TAG "ge_complex.hip,ge_train.hip" -DGE_EXP_FOO=1 ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graphembeddings_amd import build as B

tag, srcs, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
out_dir = os.path.join(B.HERE, "_variants")
os.makedirs(out_dir, exist_ok=True)
objs, procs = [], []
for s in B.SOURCES:
    if s in srcs:
        o = os.path.join(out_dir, f"{tag}_{s.replace('.hip', '.o')}")
        cmd = [B._hipcc(), "-O3", f"--offload-arch={B.ARCH}", "-std=c++17", "-fPIC", *flags, *B.FILE_FLAGS.get(s, []), "-c", s, "-o", o]
        procs.append((cmd, subprocess.Popen(cmd, cwd=B.CSRC)))
    else:
        o = os.path.join(B.OBJDIR, s.replace(".hip", ".o"))
    objs.append(o)
for cmd, p in procs:
    if p.wait() != 0:
        raise SystemExit("failed: " + " ".join(cmd))
lib = os.path.join(out_dir, f"libge_{tag}.so")
subprocess.check_call([B._hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", lib, *objs], cwd=B.CSRC)
print(lib)
