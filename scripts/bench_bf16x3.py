#!/usr/bin/env python
"""Launch time of every conv_bf16x3_kernel variant (activation / statistics / accumulate) at 6x512x432x32, interleaved
repetitions so that clock drift hits all of them alike.   TAG=name python scripts/bench_bf16x3.py"""
import sys, os; sys.path.insert(0,'/root/repo')
import torch
from depthinspace_amd import lib
dev='cuda'
n,h,w=6,512,432
x=torch.randn(n,h,w,32,device=dev); wt=torch.randn(32,32,3,3,device=dev)*0.05; b=torch.randn(32,device=dev)
pk=torch.empty(9*3*4*32*8,dtype=torch.int16,device=dev)
lib.call('dis_conv2d_pack_weights_bf16x3',wt,pk,32,32,3,0)
y=torch.zeros(n,h,w,32,device=dev)
st=torch.zeros(2*n,dtype=torch.float64,device=dev)
def t(act,stats,bias):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): lib.call('dis_conv2d_fwd_bf16x3',x,pk,bias,y,stats,n,h,w,32,32,3,1,1,act)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/50*1e3
gy=torch.randn(n,h,w,32,device=dev); gw=torch.empty_like(wt); gb=torch.empty(32,device=dev)
ws=torch.empty(lib.fn('dis_conv2d_wgrad_workspace')(32,32,3,1),device=dev)
def tw():
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): lib.call('dis_conv2d_wgrad_bf16x3',x,gy,gw,gb,ws,n,h,w,32,32,32,3,1,1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/50*1e3
cfg=[('selu+stats',1,st),('selu',1,None),('none',0,None),('none+stats',0,st),('accum',0x100,None)]
for _ in range(100): lib.call('dis_conv2d_fwd_bf16x3',x,pk,b,y,st,n,h,w,32,32,3,1,1,1)
res={k:[] for k,_,_ in cfg}
for rep in range(5):
    for k,a,s in cfg:
        y.zero_()
        res[k].append(t(a,s,b))
res['wgrad (3 launches)']=[tw() for _ in range(5)]
for k in res: print(os.environ.get('TAG',''), k, ' '.join(f'{v:.1f}' for v in res[k]))
