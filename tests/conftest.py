import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_addoption(parser):
    parser.addoption("--syn3r-lib", default=None,
                     help="developer A/B: run the suite against another BUILD of libsyn3r_hip.so (tools/build_variant.sh)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if config.getoption("--syn3r-lib"):
        from syn3r_amd import _lib
        _lib.set_library_path(config.getoption("--syn3r-lib"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible (gpu tests must not silently pass)")
    return torch.device("cuda", 0)


def record_measurement(name: str, **values) -> None:
    """Append one JSON line with a test's MEASURED errors to gpurun_out/test_measurements.jsonl (scratch, merged back from the
    GPU box): the tolerances written in the tests are set from these records (<= 2x the measurement)."""
    import json
    out = ROOT / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        with open(out / "test_measurements.jsonl", "a") as f:
            f.write(json.dumps({"test": name, **values}) + "\n")
    except OSError:
        pass


@pytest.fixture
def measurements():
    """`record_measurement` as a fixture (tests need not import this module as `tests.conftest`, which depends on the rootdir
    being on sys.path)."""
    return record_measurement
