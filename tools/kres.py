#!/usr/bin/env python3
"""Per-kernel resource table from hipcc -Rpass-analysis=kernel-resource-usage (stdin): name, SGPRs, VGPRs, AGPRs, spills, scratch."""
import re
import subprocess
import sys
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"s2sr::\(anonymous namespace\)::|void |\(s2sr::ConvParams\)", "", cur)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    if sys.argv[1:] and not any(a in k for a in sys.argv[1:]):
        continue
    print(f"{k:70s} sgpr {v.get('TotalSGPRs', 0):4d} vgpr {v.get('VGPRs', 0):4d} agpr {v.get('AGPRs', 0):4d} sspill {v.get('SGPRs Spill', 0):4d} vspill {v.get('VGPRs Spill', 0):3d} scratch {v.get('ScratchSize [bytes/lane]', 0):4d}")
