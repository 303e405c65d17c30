"""Operator boundary of the reference: `model.ext_functions.photometric_loss`
(/root/reference/model/ext_functions.py:115-154), routed to libdis_hip.so instead of
connecting_the_dots' `ext_cpu` / `ext_cuda` pybind modules.

Same call signature, same type strings, same autograd contract (gradient wrt `es` only).
The four dead wrappers of the reference (nn, crosscheck, proj_nn, xcorrvol; no call sites) are not provided.
"""
from .. import ops


def photometric_loss(es, ta, block_size, type='mse', eps=0.1):
    type = type.lower()
    if type == 'mse':
        type = 0
    elif type == 'sad':
        type = 1
    elif type == 'census_mse':
        type = 2
    elif type == 'census_sad':
        type = 3
    else:
        raise Exception('invalid loss type')
    return ops.photometric(es, ta, block_size, type, eps)
