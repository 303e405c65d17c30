"""CPU: the tie-breaking rule of the HIP neighbour selection (depthinspace_amd/csrc/nth_select.h, host build in
oracle/_build/libdis_host.so) returns exactly what torch.topk(k=9 of 36, largest=False, sorted=False) returns - ids AND
order - on random, heavily tied, masked-fill and quantised keys; its depth-limit branch equals libstdc++'s __heap_select."""
import ctypes
import numpy as np
import pytest
import torch

from tests import bitexact as B


def _std(keys, k=9):
    keys = np.ascontiguousarray(keys, np.float32)
    rows, n = keys.shape
    out = np.empty((rows, k), np.int32)
    B.host_lib().stdsel_rows(ctypes.c_void_p(keys.ctypes.data), ctypes.c_long(rows), ctypes.c_int(n), ctypes.c_int(k),
                             ctypes.c_void_p(out.ctypes.data))
    return out


@pytest.mark.parametrize('kind', ['random', 'few_distinct', 'masked_fill', 'quantised', 'nan'])
def test_nth_select_is_torch_topk(kind):
    rng = np.random.RandomState(3)
    rows = 50000
    keys = rng.rand(rows, 36)
    if kind == 'few_distinct':
        keys = rng.randint(0, 4, (rows, 36))
    elif kind == 'masked_fill':
        keys = np.where(rng.rand(rows, 36) < 0.7, np.finfo(np.float32).max, keys)
    elif kind == 'quantised':
        keys = np.round(keys * 6) / 6
    elif kind == 'nan':
        keys = np.where(rng.rand(rows, 36) < 0.1, np.nan, np.round(keys * 8) / 8)
    keys = keys.astype(np.float32)
    ours = B.topk9(keys)
    assert np.array_equal(ours, _std(keys))
    _, ti = torch.topk(torch.from_numpy(keys).unsqueeze(-1), 9, dim=1, largest=False, sorted=False)
    assert np.array_equal(ours, ti[..., 0].numpy())
    bad = B.host_lib().heapsel_mismatches(ctypes.c_void_p(keys.ctypes.data), ctypes.c_long(rows), 36, 9)
    assert bad == 0
