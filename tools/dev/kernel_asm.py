"""Cuts one kernel out of a hipcc -S --cuda-device-only listing and prints its memory / wait / branch skeleton.
usage: python tools/dev/kernel_asm.py <file.s> <mangled-name-substring> [out.s]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r"^(\S*" + re.escape(sys.argv[2]) + r"\S*):", s, re.M)
i = m.start()
j = s.index("s_endpgm", i)
body = s[i:j]
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(body)
lines = body.split("\n")
c = collections.Counter()
for l in lines:
    mm = re.match(r"\s+([a-z_0-9]+)", l)
    if mm:
        c[mm.group(1)] += 1
print(len(lines), "lines;", ", ".join(f"{k} {v}" for k, v in c.most_common(16)))
for n, l in enumerate(lines):
    if re.search(r"global_load|global_store|s_waitcnt|scratch_|s_cbranch|^\.LBB|s_barrier|ds_bpermute|ds_swizzle|global_atomic", l):
        print(n, " ".join(l.split()[:5]))
