#!/usr/bin/env python3
"""A/B builds of the library: python profiles/build_variant.py NAME file.hip [-DFLAG ...] -> variants/libroam_NAME.so
(the named translation unit recompiled with the extra flags, every other object taken from the regular build)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from radarslampy_amd import build as B
name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()
os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
obj = os.path.join(ROOT, "variants", f"{src[:-4]}_{name}.o")
cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + B.EXTRA.get(src, []) + extra + ["-c", os.path.join(B.CSRC, src), "-o", obj]
subprocess.run(cmd, check=True)
objs = [obj if f == src else os.path.join(B.CSRC, f[:-4] + ".o") for f in B._sources()]
out = os.path.join(ROOT, "variants", f"libroam_{name}.so")
subprocess.run(["/opt/rocm/bin/hipcc", f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", out] + objs + ["-ldl", "-lz"], check=True)
print(out)
