import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


class conv_split(object):
    """context manager: run a block with the 3x3 conv kernels on the given operand split ('bf16x3': three bf16 terms, six
    products; 'f16x2': two fp16 terms, three products - the default), then restore the previous mode"""

    def __init__(self, mode):
        self.mode = {'bf16x3': 0, 'f16x2': 1}[mode]

    def __enter__(self):
        from depthinspace_amd import lib
        self.prev = lib.fn('dis_get_conv_split')()
        assert lib.fn('dis_set_conv_split')(self.mode) == 0

    def __exit__(self, *a):
        from depthinspace_amd import lib
        lib.fn('dis_set_conv_split')(self.prev)
        return False
