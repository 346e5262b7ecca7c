#!/usr/bin/env python3
"""Kernel tuning helper: builds libbadslam_hip.so variants with extra -D flags into
tools/variants/ (here, cross-compiled) and benches each on the GPU box.

  python tools/variants.py build name1:-DA=1,-DB=2 name2:-DA=3 ...
  python tools/variants.py bench [bench.py args...]      (on the GPU box)
"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "variants")
sys.path.insert(0, ROOT)


def build(specs):
    from badslam_amd import build as b
    os.makedirs(OUT, exist_ok=True)
    for f in glob.glob(os.path.join(OUT, "*.so")):
        os.remove(f)
    procs = []
    for spec in specs:
        name, _, defs = spec.partition(":")
        flags = [d for d in defs.split(",") if d]
        cmd = [b.HIPCC] + b.FLAGS + flags + b.sources() + ["-o", os.path.join(OUT, f"libbadslam_hip_{name}.so")]
        procs.append((name, subprocess.Popen(cmd)))
    for name, p in procs:
        if p.wait() != 0:
            raise SystemExit(f"variant {name} failed to build")


def bench(args):
    for so in sorted(glob.glob(os.path.join(OUT, "*.so"))):
        env = dict(os.environ, BSLAM_HIP_LIB=so)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline", "0", "--pcg", "0", "--secondary", "0"] + args, env=env,
                           capture_output=True, text=True)
        name = os.path.basename(so)[len("libbadslam_hip_"):-3]
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
            roof = j["roofline"]
            print(f"{name:24s} ms/step {j['ms_per_step']:9.3f}  pose_acc {roof['avg_launch_us']:9.1f} us (frac {roof['frac']:.3f})  geometry {roof['geometry_kernel']['us_per_step']:9.1f} us/step "
                  f"(frac {roof['geometry_kernel']['frac']:.3f})  activation {roof['activation_kernel']['avg_launch_us']:7.1f} us", flush=True)
        except Exception:
            print(name, "FAILED", r.stdout[-500:], r.stderr[-1500:], flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        bench(sys.argv[2:])
