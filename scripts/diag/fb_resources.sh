#!/bin/bash
# resource usage of every instance of conv_bwd_fused.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c /root/repo/depthinspace_amd/csrc/conv_bwd_fused.hip -I/root/repo/depthinspace_amd/csrc -o /tmp/fb.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re
cur=None;rows=[]
for l in sys.stdin:
    if 'error' in l: print(l)
    m=re.search(r'Name: (\S+)',l)
    if m: cur={'name':m.group(1)[23:-8]}; rows.append(cur)
    for k in ['AGPRs','VGPRs Spill']:
        m=re.search(k+r': (\d+)',l)
        if m and cur is not None: cur[k]=m.group(1)
for r in rows: print(r['name'], r.get('AGPRs'), r.get('VGPRs Spill'))
"
