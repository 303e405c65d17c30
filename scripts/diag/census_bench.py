#!/usr/bin/env python
"""Times the census window loss kernels at the DIS-SF bs=8 size (32 x 1 x 512 x 432): four single-estimate launches against the
multi-estimate launch, forward and backward.  DIS_HIP_LIB selects the library build (variants of csrc/pixel_ops.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from depthinspace_amd import lib

n, h, w, S = 32, 512, 432, 4
g = torch.Generator().manual_seed(0)
ta = torch.randn(n, 1, h, w, generator=g).cuda()
es = torch.randn(S, n, 1, h, w, generator=g).cuda()
go = torch.rand(S, n, 1, h, w, generator=g).cuda()
out = torch.empty_like(es)
ge = torch.empty_like(es)


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def single_fwd():
    for k in range(S):
        lib.call('dis_photometric_fwd', es[k], ta, out[k], n, 1, h, w, 9, 3, 0.5)


def single_bwd():
    for k in range(S):
        lib.call('dis_photometric_bwd', es[k], ta, go[k], ge[k], n, 1, h, w, 9, 3, 0.5)


def multi1_fwd():
    for k in range(S):
        lib.call('dis_photometric_fwd_multi', es[k], ta, out[k], 1, n, h, w, 9, 3, 0.5)


def multi1_bwd():
    for k in range(S):
        lib.call('dis_photometric_bwd_multi', es[k], ta, go[k], ge[k], 1, n, h, w, 9, 3, 0.5)


print('4 x multi with ONE estimate: fwd %.3f ms, bwd %.3f ms' % (timeit(multi1_fwd), timeit(multi1_bwd)))
print(os.environ.get('DIS_HIP_LIB', 'default'),
      'fwd 4 x single %.3f ms, multi %.3f ms;  bwd 4 x single %.3f ms, multi %.3f ms' % (
          timeit(single_fwd), timeit(lambda: lib.call('dis_photometric_fwd_multi', es, ta, out, S, n, h, w, 9, 3, 0.5)),
          timeit(single_bwd), timeit(lambda: lib.call('dis_photometric_bwd_multi', es, ta, go, ge, S, n, h, w, 9, 3, 0.5))))
