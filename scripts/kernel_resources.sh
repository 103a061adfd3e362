#!/bin/bash
# VGPR / SGPR / spill / LDS per kernel of one translation unit: scripts/kernel_resources.sh dlpd_corr.hip [extra flags]
ROOT=$(cd $(dirname $0)/.. && pwd)
SRC=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -I $ROOT/deeplocalproteindocking_amd/csrc -I $ROOT/include \
  -Rpass-analysis=kernel-resource-usage "$@" -c $ROOT/deeplocalproteindocking_amd/csrc/$SRC -o /tmp/kr_$$.o 2>&1 | python3 -c "
import re,sys,subprocess
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r'remark:\s+(.*?)\s*\[-Rpass', l)
    if not m: continue
    t=m.group(1)
    if t.startswith('Function Name:'):
        cur={'name':t.split(':',1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in t:
        k,v=t.split(':',1); cur[k.strip()]=v.strip()
names=subprocess.run(['c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.splitlines()
for r,n in zip(rows,names):
    n=n.split('(')[0].replace('void ','')
    print('%-46s vgpr %-4s agpr %-3s sgpr %-4s spill %-4s scratch %-5s lds %-7s occ %s'%(n[:46],r.get('VGPRs'),r.get('AGPRs'),r.get('TotalSGPRs'),r.get('VGPRs Spill'),r.get('ScratchSize [bytes/lane]'),r.get('LDS Size [bytes/block]'),r.get('Occupancy [waves/SIMD]')))
"
rm -f /tmp/kr_$$.o
