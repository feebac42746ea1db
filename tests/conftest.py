import gzip
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.join(ROOT, "tests")
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_fasta_gz(path):
    reads = []
    with gzip.open(path, "rb") as f:
        for line in f:
            if not line.startswith(b">"):
                reads.append(line.strip())
    return reads


@pytest.fixture(scope="session")
def example_reads():
    return read_fasta_gz(os.path.join(GOLDEN, "reads-0.00.fa.gz"))
