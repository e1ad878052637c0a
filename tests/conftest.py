import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_fastq(path):
    """Minimal strict 4-line FASTQ reader for test fixtures (plain or .gz)."""
    import gzip

    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rb") as f:
        lines = f.read().split(b"\n")
    return [lines[i].rstrip(b"\r") for i in range(1, len(lines), 4)]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
