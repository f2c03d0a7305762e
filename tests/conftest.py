import json
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str):
    with open(GOLDEN / name) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def mb_cases():
    return load_golden("mb_cases.json") + load_golden("mb_cases_v2.json")


@pytest.fixture(scope="session")
def sb_cases():
    return load_golden("sb_cases.json") + load_golden("sb_cases_v2.json")


@pytest.fixture(scope="session")
def jd_cases():
    return load_golden("jd_cases.json") + load_golden("jd_cases_v2.json")


@pytest.fixture(scope="session")
def jdn_cases():
    return load_golden("jdn_cases.json")


@pytest.fixture(scope="session")
def kernel_vectors():
    return load_golden("kernel_vectors.json")
