#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
   python3 tools/kernel_resources.py bark_amd/csrc/chol.hip [extra hipcc flags...]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
       "-Rpass-analysis=kernel-resource-usage", *sys.argv[2:], "-x", "hip", "-c", src, "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z \[\]/]+?): (\S+) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
for r in rows:
    nm = re.sub(r"bark::\(anonymous namespace\)::", "", r["name"])
    nm = re.sub(r"\(.*", "", nm)
    print(f"{nm[:70]:70s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('LDS Size [bytes/block]','?'):>7s}")
