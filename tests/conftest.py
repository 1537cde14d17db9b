import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    # every parity error the tests measured (worst per test) -> gpurun_out/parity_report.json (scratch; copied to profiles/ by hand)
    try:
        from tests.helpers import dump_records
        dump_records(os.path.join(REPO, "gpurun_out", "parity_report.json"))
    except Exception:  # reporting must never fail a run
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR
