import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The package has no CPU / eager / vendor path of its own (far_amd/_vendor.py): the torch compositions the CPU tests and the
    # vendor comparison legs need are test infrastructure and are installed here, for the test session only.
    from far_amd import _vendor
    from tests import vendor_ops
    _vendor.install(vendor_ops)


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
