"""A fixed slice of the seeded parity fuzz (tools/parity_fuzz.py: random channels, sample sizes, presets, block sizes,
option overrides and signal kinds - full-scale noise, square waves, near-identical channels, wasted bits - through the
device engine, the oracle and the device decoder).  The sweep that found the frames beyond the verbatim size
(tests/cases.py: noise24-wrapped-estimate) ran over tens of thousands of seeds; these run with every test pass."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", list(range(0, 240, 40)))
def test_seeded_random_configurations_match_the_oracle(gpu_engine, first):
    import parity_fuzz
    ran = 0
    for seed in range(first, first + 40):
        r = parity_fuzz.run_seed(seed)
        if r is None:
            continue
        assert r[0], r[1]
        ran += 1
    assert ran >= 25


# Seeds on which a development build once differed from the oracle: blocks shorter than the 4096 samples a search
# workgroup covers (2304, 1152, 192, 16 samples: waves with lanes beyond the block's end, down to a single lane with
# samples), where the candidate loop's votes across the wave must still see every candidate.
@pytest.mark.parametrize("seed", [300077, 300186, 300284, 301107, 301932, 302579, 302919, 303187, 303692, 304242, 304850, 305270, 305523])
def test_seeds_that_once_differed(gpu_engine, seed):
    import parity_fuzz
    r = parity_fuzz.run_seed(seed)
    assert r is not None and r[0], r
