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


def rows_crc(rows) -> int:
    """Digest of token rows as tests/golden/gen_golden.py::rows_crc forms it (the runaway cases of mb_cases_v3.json store
    their forwards as digests): crc32 over the little-endian int64 image, shape included."""
    import zlib

    import numpy as np
    a = np.asarray(rows, dtype="<i8")
    return zlib.crc32(a.tobytes(), zlib.crc32(np.asarray(a.shape, dtype="<i8").tobytes()))


def forward_matches(got: dict, want: dict, greedy: bool = True) -> bool:
    """One forward of a multiblock call (kv_len, out rows[, greedy rows]) against its golden record, full or digest."""
    if got["kv_len"] != want["kv_len"]:
        return False
    if "out_crc" in want:
        return ((len(got["out"]), len(got["out"][0])) == (want["B"], want["T"]) and rows_crc(got["out"]) == want["out_crc"]
                and (not greedy or rows_crc(got["greedy"]) == want["greedy_crc"]))
    return got["out"] == want["out"] and (not greedy or got["greedy"] == want["greedy"])


def kv_matches(kv_tokens, call: dict) -> bool:
    return rows_crc(kv_tokens) == call["kv_tokens_crc"] if "kv_tokens_crc" in call else kv_tokens == call["kv_tokens"]


@pytest.fixture(scope="session")
def mb_cases():
    return load_golden("mb_cases.json") + load_golden("mb_cases_v2.json") + load_golden("mb_cases_v3.json")


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
