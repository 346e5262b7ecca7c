#!/usr/bin/env python3
"""Static instruction mix of the device kernels (cross-compiles csrc to ISA with the library's flags).
usage: tools/isa_stats.py <kernel-name substring> [-DFLAG=...] [--dump file]   e.g. tools/isa_stats.py 'pose_accumulate_kernel<true, true, 2>'"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from badslam_amd import build as b   # noqa: E402

args = sys.argv[1:]
dump = None
if "--dump" in args:
    i = args.index("--dump")
    dump = args[i + 1]
    del args[i:i + 2]
flt = [a for a in args if not a.startswith("-D")]
defs = [a for a in args if a.startswith("-D")]
flags = [f for f in b.FLAGS if f not in ("-fPIC", "-shared")]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run([b.HIPCC] + flags + defs + ["-S", "--cuda-device-only", "-o", out] + b.sources(), check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void bslam::", "")
    if flt and not any(f in name for f in flt):
        continue
    body = m.group(2)
    ins = [ln.strip().split()[0] for ln in body.splitlines() if ln.startswith("\t") and not ln.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print(f"{name}\n   {len(ins)} instructions: VALU {valu} (v_mov {c['v_mov_b32_e32']}, v_cndmask {c['v_cndmask_b32_e32'] + c['v_cndmask_b32_e64']}, "
          f"dpp {body.count(' row_') + body.count('quad_perm')}, v_readlane/firstlane {c['v_readlane_b32'] + c['v_readfirstlane_b32']}), "
          f"SALU {sum(v for k, v in c.items() if k.startswith('s_') and not k.startswith(('s_waitcnt', 's_nop', 's_barrier', 's_cbranch', 's_branch')))}, "
          f"ds_bpermute {c['ds_bpermute_b32']}, ds_read/write {sum(v for k, v in c.items() if k.startswith(('ds_read', 'ds_write')))}, "
          f"global_load {sum(v for k, v in c.items() if k.startswith('global_load'))}, global_store {sum(v for k, v in c.items() if k.startswith('global_store'))}, "
          f"s_load {sum(v for k, v in c.items() if k.startswith('s_load'))}, s_waitcnt {c['s_waitcnt']}, s_nop {c['s_nop']}, s_barrier {c['s_barrier']}, branches {sum(v for k, v in c.items() if k.startswith(('s_cbranch', 's_branch')))}")
    if dump:
        open(dump, "w").write(body)
