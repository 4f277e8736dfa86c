"""Worker of tests/test_distributed_cpu.py: commet_amd.matrix.run under torch.distributed.run (gloo, no GPU) with the
CPU checker as the engine.  argv: sets.txt out_dir k t [rank that fails]"""
import json
import os
import sys
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from commet_amd import matrix  # noqa: E402
from oracle_engine import OracleEngine  # noqa: E402


def main():
    sets, out, k, t = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    if len(sys.argv) > 5:
        OracleEngine.fail_on_rank = int(sys.argv[5])
    try:
        res = matrix.run(sets, out, k=k, t=t, verbose=False, engine_factory=OracleEngine,
                         fatal_hook=lambda msg: open(os.path.join(out, f"last_words_rank{os.environ.get('RANK', '0')}.txt"), "w").write(msg))
    except BaseException:
        traceback.print_exc()
        sys.stderr.flush()
        if os.environ.get("COMMET_TEST_SOFT_FAIL"):      # a caller that catches (bench.py's matrix leg): leaves in its own time;
            import time                                  # the other ranks must have been told by matrix.run itself: give them
            time.sleep(3)                                # the time to say so before the launcher ends them
            sys.exit(5)
        os._exit(1)                 # what commet_amd.matrix.main does
    if res is not None:
        res.pop("rank0_profile")
        for r in res["per_rank"]:
            r.setdefault("backend", None)
        json.dump(res, open(os.path.join(out, "result.json"), "w"))


if __name__ == "__main__":
    main()
