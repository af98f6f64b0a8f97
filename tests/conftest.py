import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: opt-in long GPU checks (minutes of CPU oracle): run with -m \"gpu and slow\"")


def pytest_collection_modifyitems(config, items):
    """`slow` tests are opt-in: they run only when the marker expression names them (-m "gpu and slow")"""
    if "slow" in (config.getoption("markexpr") or ""):
        return
    skip = pytest.mark.skip(reason='opt-in: run with -m "gpu and slow"')
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def parity_log():
    """append measured parity figures (json lines) to gpurun_out/parity_report.jsonl: the stated
    tolerances in the tests are kept at ~2x the values recorded here on the MI355X box"""
    import json
    path = os.path.join(REPO, "gpurun_out", "parity_report.jsonl")
    os.makedirs(os.path.dirname(path), exist_ok=True)

    def log(test, **figures):
        with open(path, "a") as f:
            f.write(json.dumps({"test": test, **figures}) + "\n")
    return log
