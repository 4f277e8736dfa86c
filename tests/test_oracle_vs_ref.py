"""The CPU restatement (oracle/) against the compiled reference (oracle/_ref),
live, on randomised scenarios.  Runs only where the reference was built."""
import pytest

from scenarios import Scenario, compare_runs, run_oracle, run_tool


@pytest.mark.parametrize("seed", range(40))
def test_oracle_matches_reference(tmp_path, ref_index_and_search, seed):
    scn = Scenario(str(tmp_path / "scn"), seed)
    out_r, log_r = str(tmp_path / "out_ref"), str(tmp_path / "log_ref")
    out_o, log_o = str(tmp_path / "out_orc"), str(tmp_path / "log_orc")
    p = run_tool(ref_index_and_search, scn, out_r, log_r)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    rc, res, chunks, kmers = run_oracle(scn, out_o, log_o)
    assert rc == 0
    compare_runs(out_r, log_r, out_o, log_o, scn)


@pytest.mark.parametrize("seed", range(500, 530))
def test_oracle_matches_reference_fastq_and_gzip(tmp_path, ref_index_and_search, seed):
    """FASTQ and gzip inputs, format chosen per file"""
    scn = Scenario(str(tmp_path / "scn"), seed, formats=("fa", "fq", "fa.gz", "fq.gz"), crlf=False)
    out_r, log_r = str(tmp_path / "out_ref"), str(tmp_path / "log_ref")
    out_o, log_o = str(tmp_path / "out_orc"), str(tmp_path / "log_orc")
    p = run_tool(ref_index_and_search, scn, out_r, log_r)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    rc, res, chunks, kmers = run_oracle(scn, out_o, log_o)
    assert rc == 0
    compare_runs(out_r, log_r, out_o, log_o, scn)
