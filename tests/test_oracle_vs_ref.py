"""The CPU restatement (oracle/) against the compiled reference (oracle/_ref),
live, on randomised scenarios.  Runs only where the reference was built."""
import os

import pytest

import oracle_binding as ob
import util
from scenarios import Scenario, compare_runs


@pytest.mark.parametrize("seed", range(40))
def test_oracle_matches_reference(tmp_path, ref_index_and_search, seed):
    scn = Scenario(str(tmp_path / "scn"), seed)
    out_r, log_r = str(tmp_path / "out_ref"), str(tmp_path / "log_ref")
    out_o, log_o = str(tmp_path / "out_orc"), str(tmp_path / "log_orc")
    p = util.run([ref_index_and_search, "-i", scn.index_cfg, "-s", scn.search_cfg, "-o", out_r, "-l", log_r,
                  "-k", str(scn.k), "-t", str(scn.t)], check=False)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    rc, res, chunks, kmers = ob.index_and_search(scn.index_cfg, scn.search_cfg, out_o, log_o, scn.k, scn.t)
    assert rc == 0
    compare_runs(out_r, log_r, out_o, log_o, scn)
