import os
import sys

import pytest
import torch  # noqa: F401  first, so its bundled HIP runtime is the one libspsp.so shares (see supersampler_amd/__init__.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_report_header(config):
    try:
        import supersampler_amd as sp
        i = sp.library_info()
        return "libspsp: %s [%s]%s" % (i["path"], i["version"], "  ** SPSP_LIB override **" if i["overridden_by_SPSP_LIB"] else "")
    except Exception as e:  # noqa: BLE001 -- the header must not stop collection (the ABI test reports a missing library)
        return "libspsp: not loadable (%s)" % e
