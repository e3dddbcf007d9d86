"""Shared pytest plumbing: markers, golden-fixture loader, fp32 ulp comparison."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as fh:
        return json.load(fh)


@pytest.fixture
def golden():
    return load_golden


def assert_float_eq(got, expected, ulps=4, what=""):
    """gtest's EXPECT_FLOAT_EQ: within `ulps` units in the last place in fp32."""
    got = np.asarray(got, np.float32).ravel()
    expected = np.asarray(expected, np.float32).ravel()
    assert got.shape == expected.shape, (what, got.shape, expected.shape)

    def key(x):  # monotone integer image of the fp32 line
        i = x.view(np.int32).astype(np.int64)
        return np.where(i < 0, -(i & 0x7FFFFFFF), i)

    d = np.abs(key(got) - key(expected))
    bad = np.nonzero(d > ulps)[0]
    assert bad.size == 0, "%s: %d elements differ by > %d ulp, first idx %d got %r exp %r" % (
        what, bad.size, ulps, bad[0], got[bad[0]], expected[bad[0]])


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
