"""The randomised GPU-vs-oracle campaign inside the driver-run suite (round 2 kept it in tools/soak.py,
builder-run only): 600 random cases in 24 blocks + 6 large ones (2 000 - 7 000 genomes: many ref tiles of
the default-shape kernel), seeds disjoint from the builder's campaigns.  tests/soak_case.py draws a
case: shapes, k lists, sketch sizes, bbits, cluster tables, [EXT] switches, every output mode."""
import numpy as np
import pytest

from soak_case import reset_options, soak_case

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_options():
    yield
    reset_options()


@pytest.mark.parametrize("block", range(24))
def test_soak_block_of_25_random_cases(block):
    rng = np.random.Generator(np.random.PCG64(9_000_000 + block))
    bad = []
    for case in range(25):
        desc, msgs = soak_case(rng)
        if msgs:
            bad.append("block %d case %d %s: %s" % (block, case, desc, "; ".join(msgs)))
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("seed", range(6))
def test_soak_large_case(seed):
    rng = np.random.Generator(np.random.PCG64(9_100_000 + seed))
    desc, msgs = soak_case(rng, big=True)
    assert not msgs, "%s: %s" % (desc, "; ".join(msgs))
