#!/usr/bin/env python3
"""Development: registers, spilled bytes and LDS of the library's kernels whose mangled name contains every argument."""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd", "bkcore", "libbkcore.so")
LLVM = "/opt/rocm/lib/llvm/bin"
d = tempfile.mkdtemp()
shutil.copy(LIB, os.path.join(d, "libbkcore.so"))
subprocess.run([LLVM + "/llvm-objdump", "--offloading", "libbkcore.so"], cwd=d, check=True, capture_output=True)
for f in sorted(os.listdir(d)):
    if "amdgcn" not in f:
        continue
    txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", f], cwd=d, check=True, capture_output=True, text=True).stdout
    name, res = None, {}
    def flush():
        if name and all(p in name for p in sys.argv[1:]):
            print(f.split(".")[1], name[:90], res)
    for line in txt.splitlines():
        m = re.match(r"\s+\.name:\s+(\S+)", line)
        if m:
            flush(); name, res = m.group(1), {}
            continue
        m = re.match(r"\s+\.(private_segment_fixed_size|vgpr_count|group_segment_fixed_size):\s+(\d+)", line)
        if m and name:
            res[m.group(1)[:7]] = int(m.group(2))
    flush()
shutil.rmtree(d)
