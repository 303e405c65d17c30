"""GPU: dis_allreduce_* (the C ABI's gradient exchange) on the one device a test box has: the library
finds RCCL by itself, a one-rank communicator reduces in place (sum and mean of one rank = the input), handles are checked, and the
trainer's wrapper (trainer.AbiComm, what FlatAdam uses under DIS_ALLREDUCE=abi) drives the same entry points.  The N > 1 semantics
(sum over ranks, mean for the DP step) are RCCL's; the bucketing / ordering logic around them is covered on CPU with gloo
(tests/test_distributed.py)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_reduces_in_place():
    from depthinspace_amd import lib as L
    from depthinspace_amd.trainer import AbiComm
    comm = AbiComm(rank=0, world_size=1)
    assert len(comm.unique_id) == 128 and any(comm.unique_id)
    x = torch.randn(1 << 20, device='cuda')
    ref = x.clone()
    comm.all_reduce(x)
    comm.all_reduce(x, average=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):       # on the caller's stream, asynchronous
        comm.all_reduce(x[: 12345])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    # argument checks: count 0 is a no-op, a negative count / a foreign handle / NULL are refused
    f = L.fn('dis_allreduce_sum_f32')
    assert f(comm.handle, x.data_ptr(), 0, 0, None) == 0
    assert f(comm.handle, x.data_ptr(), -1, 0, None) == -1
    assert f(None, x.data_ptr(), 4, 0, None) == -3
    fake = ctypes.create_string_buffer(64)
    assert f(ctypes.cast(fake, ctypes.c_void_p), x.data_ptr(), 4, 0, None) == -1
    assert L.fn('dis_allreduce_init')(None, None, 1, 0) == -3
    h = ctypes.c_void_p()
    idb = ctypes.create_string_buffer(comm.unique_id, 128)
    assert L.fn('dis_allreduce_init')(ctypes.cast(ctypes.pointer(h), ctypes.c_void_p), ctypes.cast(idb, ctypes.c_void_p), 2, 2) == -1
    comm.close()
    assert comm.handle is None
