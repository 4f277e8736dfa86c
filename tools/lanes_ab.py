"""A/B of an option (default: the two-lane index phase, index_lanes=2,1) on the BASELINE configs[1] job:
  [AB_OPTION=part_packed AB_VALUES=1,0] python tools/lanes_ab.py [reads] [k] [read length]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import commet_amd  # noqa: E402
from commet_amd import synth  # noqa: E402


OPTION = os.environ.get("AB_OPTION", "index_lanes")
VALUES = [int(x) for x in os.environ.get("AB_VALUES", "2,1").split(",")]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    b0, o0 = synth.synth_set(0, n, L)
    b1, o1 = synth.synth_set(1, n, L)
    with commet_amd.Context(k=k, t=2) as ctx:
        s0 = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
        s1 = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
        ref = None
        for rep in range(3):
            for lanes in VALUES:
                ctx.set_option(OPTION, lanes)
                t0 = time.perf_counter()
                tags, st, inf = ctx.index_and_search(s0, [s1])
                wall = (time.perf_counter() - t0) * 1e3
                ref = ref if ref is not None else tags[0].tobytes()
                assert ref == tags[0].tobytes(), "lanes change results"
                print(f"{OPTION}={lanes}: chunks {inf['n_chunks']} index {inf['index_ms']:.2f} ms search {inf['search_ms']:.2f} ms "
                      f"call {wall:.2f} ms shared {st[0]['shared']}", flush=True)


if __name__ == "__main__":
    main()
