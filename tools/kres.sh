#!/bin/bash
# compact per-kernel resource summary: tools/kres.sh file.hip [more.hip ...]
cd "$(dirname "$0")/../ssrs_amd/csrc"
for f in "$@"; do
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -c "$f" -o "${f%.hip}.o" -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
name=None; d={}
for line in sys.stdin:
    if "error" in line or "warning:" in line: print(line.rstrip())
    m=re.search(r"remark:\s+(.*?):\s+(\S+) \[-Rpass", line)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2)
    if k=="Function Name": name=v; d={}
    else: d[k]=v
    if k.startswith("LDS Size"):
        print(name[:60], "vgpr",d.get("VGPRs"),"agpr",d.get("AGPRs"),"spill",d.get("VGPRs Spill"),"scratch",d.get("ScratchSize [bytes/lane]"),"occ",d.get("Occupancy [waves/SIMD]"),"lds",v)
'
done
