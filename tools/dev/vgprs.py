"""Condenses `hipcc -Rpass-analysis=kernel-resource-usage` output: one line per kernel matching a substring.
usage: python tools/dev/vgprs.py <remarks file> [substring]"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: Function Name: ", t)[1:]:
    name = b.split()[0]
    if pat not in name:
        continue
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    dem = re.sub(r"\(.*", "", dem).replace("void ge::", "")
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    scr, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{dem:70s} vgpr {g(' VGPRs'):>3s} sgpr {g('TotalSGPRs'):>3s} spill {g('VGPRs Spill'):>3s} scratch {scr:>4s} occ {occ}")
