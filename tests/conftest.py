import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size CPU oracle runs (enable with PB_FULL=1)")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("PB_FULL") == "1":
        return
    skip = pytest.mark.skip(reason="full-size CPU oracle run; set PB_FULL=1")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)
