import hashlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# deterministic blinding (ug_test_set_blinding) exists only in processes that ask for it before the library loads
os.environ["ULTRAGROTH_TEST_HOOKS"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def zkey():
    return open(os.path.join(GOLDEN, "circuit_final.zkey"), "rb").read()


@pytest.fixture(scope="session")
def wtns():
    return open(os.path.join(GOLDEN, "witness.wtns"), "rb").read()


@pytest.fixture(scope="session")
def vkey():
    import json
    return json.load(open(os.path.join(GOLDEN, "verification_key.json")))


def fixed_rs():
    """Deterministic blinding of SURVEY.md Appendix A: r = LE(sha256("r")[:31]), s = LE(sha256("s")[:31])."""
    r = hashlib.sha256(b"r").digest()[:31]
    s = hashlib.sha256(b"s").digest()[:31]
    return r, s


@pytest.fixture(scope="session")
def device():
    # torch bundles its own HIP runtime: when a test uses both, torch must initialise first (as bench.py does),
    # otherwise torch finds the runtime already loaded by libultragroth_hip.so and reports no GPUs
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    import ultragroth_amd as ug
    d = ug.Device(0)
    yield d
    d.close()
