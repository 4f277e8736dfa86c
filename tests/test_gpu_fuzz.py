"""A slice of the extended randomised run inside the suite (tools/fuzz_gpu.py runs tens of thousands of seeds of the same
generator outside it): 300 scenarios — every index mode, chunk-group size and input format, the bit-sliced regime (narrow
tables, wide rows), the tiled search on 32- and 64-bit keys, with and without the chunk-size hook — each against the CPU
checker's .bv bits, log numbers, chunk / k-mer counts and probe count (tests/fuzz_cases.py)."""
import pytest

from fuzz_cases import fuzz_one

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", range(20000, 20300, 50))
def test_fuzz_scenarios_match_cpu_checker(first):
    bad, hooked = [], 0
    for seed in range(first, first + 50):
        ok, hook, what = fuzz_one(seed)
        hooked += hook
        if not ok:
            bad.append(what)
    assert not bad, bad
    assert hooked >= 1            # (about one scenario in fourteen is chunked through the hook)
