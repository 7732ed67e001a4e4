#!/bin/bash
# VGPR / spill / occupancy / LDS of every kernel in a .hip file (cross-compiles, no GPU needed)
# usage: tools/kernel_resources.sh hifiles-solver_amd/csrc/fused_hex.hip [filter] [extra hipcc flags]
f=$1; pat=${2:-.}; shift; shift
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
python3 -c '
import re,sys,subprocess
cur=None; rows={}
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)",line)
    if m:
        cur=subprocess.run(["c++filt",m.group(1)],capture_output=True,text=True).stdout.strip().split("(")[0]; rows[cur]={}; continue
    m=re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]+\])?: (\d+)",line)
    if m and cur: rows[cur][m.group(1).strip()]=int(m.group(2))
for k,v in rows.items():
    print("%-60s VGPR %3d AGPR %3d spill %3d occ %d LDS %6d scratch %d"%(k.replace("void hfx::",""),v.get("VGPRs",-1),v.get("AGPRs",-1),v.get("VGPRs Spill",-1),v.get("Occupancy",-1),v.get("LDS Size",-1),v.get("ScratchSize",-1)))
' | grep -E "$pat"
