#!/usr/bin/env python3
"""Prints SGPR / VGPR / spill counts of the device kernels (cross-compiles csrc to ISA with the library's flags).
usage: tools/kernel_regs.py [substring filter] [-DFLAG=...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from badslam_amd import build as b   # noqa: E402

flt = [a for a in sys.argv[1:] if not a.startswith("-D")]
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
flags = [f for f in b.FLAGS if f not in ("-fPIC", "-shared")]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run([b.HIPCC] + flags + defs + ["-S", "--cuda-device-only", "-o", out] + b.sources(), check=True, stderr=subprocess.DEVNULL)
    t = open(out).read()
for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", t):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void bslam::", "")
    if not flt or any(f in name for f in flt):
        v = int(m.group(3))
        alloc = (v + 7) // 8 * 8
        print(f"{name:70s} sgpr {m.group(2):>3s} vgpr {v:3d} spill {m.group(4)}  waves/SIMD {min(8, 512 // max(alloc, 1))}")
