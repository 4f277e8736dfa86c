"""The chunk loop on resident sets (commet_index_and_search) against the CPU
checker run on the same FASTA files: .bv bytes and the [indexed, searched,
shared] numbers must be identical."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import util
from scenarios import Scenario, run_oracle

pytestmark = pytest.mark.gpu


def _load_set(commet, ctx, files, sdir):
    batches = [util.to_batch(util.parse_reads(os.path.join(sdir, fa))) for fa, _, _, _ in files]
    rs = commet.ReadSet.from_files(ctx, batches)
    sel = np.concatenate([s for _, _, _, s in files]) if files else np.zeros(0, bool)
    has_bv = any(bv for _, bv, _, _ in files)
    return rs, (util.bits_from_bools(sel) if has_bv else None)


@pytest.mark.parametrize("seed,index_mode", [(s, 0) for s in range(60)] + [(s, 2) for s in range(60, 80)])
def test_job_matches_oracle(tmp_path, seed, index_mode):
    import commet_amd as commet
    scn = Scenario(str(tmp_path / "scn"), seed, k=None if index_mode == 0 else [20, 21, 24, 25][seed % 4])
    out_o, log_o = str(tmp_path / "out"), str(tmp_path / "log")
    rc, res, chunks, kmers = run_oracle(scn, out_o, log_o)
    assert rc == 0
    with commet.Context(k=scn.k, t=scn.t) as ctx:
        irs, isel = _load_set(commet, ctx, scn.sets[scn.index_name], scn.dir)
        srs, ssel = [], []
        for nme in sorted(scn.search_names):          # std::map order
            r, s = _load_set(commet, ctx, scn.sets[nme], scn.dir)
            srs.append(r)
            ssel.append(s)
        ctx.set_option("count_probes", 1)
        ctx.set_option("index_mode", index_mode)
        ctx.set_option("chunk_group", [4, 1, 2, 3][seed % 4])     # chunk filters searched per pass (1 = reference order)
        tags, stats, info = ctx.index_and_search(irs, srs, isel, ssel)
        assert info["probes"] == sum(r["probes"] for r in res)        # P_ref: the reference's own probe count
        assert info["n_chunks"] == chunks
        assert info["kmers_indexed"] == kmers
        by_name = {r["name"]: r for r in res}
        for nme, tg, st in zip(sorted(scn.search_names), tags, stats):
            o = by_name[nme]
            assert (st["indexed"], st["searched"], st["shared"]) == (o["indexed"], o["searched"], o["shared"]), nme
            pos = 0
            for fa, _, reads, _ in scn.sets[nme]:
                _, n, bits = util.read_bv(os.path.join(out_o, os.path.basename(fa) + "_in_" + scn.index_name + ".bv"))
                exp = util.bools_from_bits(bits, n)
                got = util.bools_from_bits(tg, pos + n)[pos:pos + n]
                assert np.array_equal(got, exp), (nme, fa)
                pos += n
        for r in [irs] + srs:
            r.close()
