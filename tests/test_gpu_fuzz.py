"""A slice of the builder-run fuzzers inside the driver-run suite (tools/fuzz_parity.py, tools/fuzz_fused.py run the same
functions over hundreds to thousands of seeds; their logs are in profiles/)."""
import pytest

import fuzz_cases as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", range(100, 150, 10))
def test_fuzz_parity_slice_hip_path_vs_oracle(first):
    """50 random configurations (ten per case): integer stages bit for bit, images and all gradients within 1e-4 of the CPU
    oracle; a one-pixel alpha / transmittance threshold flip is the one tolerated difference (at most one per ten)."""
    flips = 0
    for seed in range(first, first + 10):
        status, errs, info = F.parity_one(seed)
        assert status != "MISMATCH", (seed, info, errs)
        flips += status == "flip"
    assert flips <= 1


@pytest.mark.parametrize("first", range(0, 20, 10))
def test_fuzz_fused_slice_one_call_path_vs_staged_calls(first):
    """20 random scenes (uniform and clustered, tied depths, lists beyond the fused kernel's LDS sort): scg_forward against the
    staged calls bit for bit."""
    longest = 0
    for seed in range(first, first + 10):
        longest = max(longest, F.fused_one(seed))
    assert longest > 1536                     # the slice reaches the rare-size sort kernels
