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


def _cpu_share():
    """CPU threads this process may really use: the affinity mask capped by the cgroup CPU quota (as bench.py's cpu_share)"""
    cores = len(os.sched_getaffinity(0))
    for quota_file, period_file in (("/sys/fs/cgroup/cpu.max", None),
                                    ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if period_file is None:
                quota, period = open(quota_file).read().split()
            else:
                quota, period = open(quota_file).read().strip(), open(period_file).read().strip()
            if quota not in ("max", "-1"):
                cores = max(1, min(cores, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
    return cores


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """the oracle's OpenMP loops with as many threads as this process may use and no more: a GPU box shows every core of the
    host, and its default -- one thread per visible core under a 16-core quota -- made the full-size H polynomials 3x slower"""
    try:
        import oracle as O
        O.lib.ugo_set_num_threads(max(1, min(_cpu_share(), 32)))
    except Exception:
        pass
    yield


@pytest.fixture(scope="session")
def device():
    # (torch bundles its own HIP runtime; the loader makes sure the process ends up with ONE copy whichever of the two is
    # imported first: ultragroth_amd/_lib.py _one_hip_runtime, tests/test_abi.py)
    import ultragroth_amd as ug
    d = ug.Device(0)
    yield d
    d.close()
