"""Frozen vectors (tests/golden/, produced by the oracle restatement -- NOT by the reference, which
cannot be built here): the oracle must keep reproducing them (CPU), and so must the HIP path (GPU)."""
import pytest

import _golden


def test_there_are_vectors():
    assert len(_golden.NAMES) >= 6


@pytest.mark.parametrize("name", _golden.NAMES)
def test_oracle_reproduces_golden(oracle, name):
    left, right, model, band, flags, d = _golden.load(name)
    _golden.check(oracle.dp_align(left, right, model, band, flags=flags), d, name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", _golden.NAMES)
def test_gpu_reproduces_golden(pg, name):
    left, right, model, band, flags, d = _golden.load(name)
    _golden.check(pg.align(left, right, model, band, flags=flags), d, name)
