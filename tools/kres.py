#!/usr/bin/env python3
"""tools/kres.py FILE.hip [name-regex]: registers / LDS / occupancy of each kernel in one source file of 3dgs-native_amd/csrc, one
line per kernel (hipcc -Rpass-analysis=kernel-resource-usage, object to /tmp; nothing in the tree is touched).
KRES_FLAGS="-DGSR_CENSUS" adds compiler flags."""
import os
import re
import subprocess
import sys

src = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3dgs-native_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
       "-I../../include", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"]
if "blend_" in src:
    cmd.insert(1, "-fno-slp-vectorize")
cmd[1:1] = os.environ.get("KRES_FLAGS", "").split()
out = subprocess.run(cmd, cwd=here, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    k, _, v = m.group(1).partition(": ")
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = cur.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        rows[cur] = {}
    elif cur:
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    if pat.search(name):
        print("%-72s vgpr %3s agpr %2s sgpr %3s scratch %3s lds %6s occ %s" % (
            name[:72], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"),
            r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
