"""Extended randomised parity run (tens of thousands of scenarios, outside the test-suite; the suite runs the first few hundred
seeds of the same generator, tests/test_gpu_fuzz.py): job-level GPU results against the CPU checker, tests/fuzz_cases.py.
  python tools/fuzz_gpu.py [first_seed] [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_cases import fuzz_one  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    bad = n_hook = 0
    for seed in range(first, first + count):
        ok, hook, what = fuzz_one(seed)
        n_hook += hook
        if not ok:
            bad += 1
            print("MISMATCH", what, flush=True)
        if (seed - first) % 100 == 99:
            print(f"... {seed - first + 1} scenarios, {bad} mismatches", flush=True)
    print(f"fuzz: {count} scenarios from seed {first} ({n_hook} of them with the chunk-size hook, the CPU checker chunked alike), {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
