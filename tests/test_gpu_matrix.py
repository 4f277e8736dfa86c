"""The resident N x N driver (commet_amd/matrix.py) against what the reference's own
driver produced: Commet.py + the reference binaries on ABCDE_bench (tests/golden/abcde/commet_py),
and against the CPU checker run through Commet.py's job sequence on synthetic sets."""
import gzip
import os
import sys

import numpy as np
import pytest

import oracle_binding as ob
from conftest import ROOT

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module", autouse=True)
def _tools():
    from commet_amd import build
    build.build_lib()
    build.build_tools()


@pytest.fixture()
def abcde(tmp_path):
    os.makedirs(tmp_path / "ABCDE_bench")
    for f, copies in (("A", "A"), ("B", "BD"), ("C", "CE")):
        data = gzip.open(os.path.join(GOLD, "abcde", f + ".fa.gz")).read()
        for c in copies:
            open(tmp_path / "ABCDE_bench" / (c + ".fa"), "wb").write(data)
    return tmp_path


@pytest.mark.parametrize("label", ["three_sets", "five_sets"])
def test_matrix_driver_reproduces_commet_py(abcde, label, monkeypatch):
    from commet_amd import matrix
    gold = os.path.join(GOLD, "abcde", "commet_py", label)
    monkeypatch.chdir(abcde)
    open("sets.txt", "w").write(open(os.path.join(gold, "sets.txt")).read())
    res = matrix.run("sets.txt", "out/", k=32, t=2, verbose=False)
    assert res["world"] == 1
    for f in sorted(os.listdir(gold)):
        if f.endswith((".csv", ".bv")):
            got = open(os.path.join("out", f), "rb").read()
            exp = open(os.path.join(gold, f), "rb").read()
            assert got == exp, f
    n = len(res["names"])
    assert sum(1 for f in os.listdir("out") if "_in_" in f and f.endswith(".bv")) == \
        sum(1 for f in os.listdir(gold) if "_in_" in f and f.endswith(".bv"))
    assert res["matrix"][0][0] == 12000 and len(res["matrix"]) == n


def test_matrix_driver_matches_oracle_on_synthetic_sets(tmp_path, monkeypatch):
    """4 sets (one of two files, one with a filter that empties a file), k=20: every .bv equals the CPU
    checker driven through Commet.py's job order with the same filter bvs."""
    from commet_amd import matrix, synth
    sys.path.insert(0, GOLD)
    from make_golden import commet_jobs
    import util
    monkeypatch.chdir(tmp_path)
    k, t, n, L = 20, 2, 6000, 80
    names = ["s0", "s1", "s2", "s3"]
    files = [["s0.fa"], ["s1a.fa", "s1b.fa"], ["s2.fa"], ["s3.fa"]]
    for s, fl in enumerate(files):
        b, o = synth.synth_set(s, n, L, copy_frac=0.3)
        if len(fl) == 1:
            synth.write_fasta(fl[0], b, o)
        else:
            h = n // 2
            synth.write_fasta(fl[0], b[: h * L], o[: h + 1])
            synth.write_fasta(fl[1], b[h * L:], o[h:] - o[h])
    rng = np.random.default_rng(1)
    bvs = []
    for s, fl in enumerate(files):
        row = []
        for j, f in enumerate(fl):
            cnt = len(util.parse_fasta(f))
            sel = rng.random(cnt) < 0.9
            if (s, j) == (1, 1):
                sel[:] = False                                   # a file with no selected read
            util.write_bv(f + ".bv", "filter of " + f, sel)
            row.append(f + ".bv")
        bvs.append(row)
    open("sets.txt", "w").write("".join(
        f"{names[s]}: " + "; ".join(f"{f},{b}" for f, b in zip(files[s], bvs[s])) + "\n" for s in range(4)))
    res = matrix.run("sets.txt", "out/", k=k, t=t, verbose=False)

    def cfg(si, restrict_to=None, out="orc"):
        parts = []
        for f, b in zip(files[si], bvs[si]):
            parts.append(f + "," + (b if restrict_to is None else f"{out}/{f}_in_{names[restrict_to]}.bv"))
        return names[si] + ":" + ";".join(parts)

    os.makedirs("orc")
    for kind, idx, searches, restr in commet_jobs(names):
        open("i.txt", "w").write(cfg(idx, restr) + "\n")
        open("s.txt", "w").write("".join(cfg(s) + "\n" for s in searches))
        rc, *_ = ob.index_and_search("i.txt", "s.txt", "orc", "orc", k, t)
        assert rc == 0
    checked = 0
    for f in sorted(os.listdir("orc")):
        if f.endswith(".bv"):
            assert open(os.path.join("out", f), "rb").read() == open(os.path.join("orc", f), "rb").read(), f
            checked += 1
    assert checked == 5 * 3            # 5 files, each searched in the 3 other sets
    # matrix rows = bit counts of those files
    for a in range(4):
        for b in range(4):
            if a != b:
                tot = sum(util.bools_from_bits(util.read_bv(f"orc/{f}_in_{names[b]}.bv")[2], util.read_bv(f"orc/{f}_in_{names[b]}.bv")[1]).sum()
                          for f in files[a])
                assert res["matrix"][a][b] == int(tot)


@pytest.mark.parametrize("world,handover,launcher", [(2, "ipc", "torchrun"), (4, "ipc", "spawn"), (2, "image", "spawn"), (2, "canary-killed", "spawn")])
def test_matrix_driver_ranks_share_the_pairs(abcde, monkeypatch, world, handover, launcher):
    """N > 1: `world` processes — started by torch.distributed.run (the driver's form) or by `python -m commet_amd.matrix
    --gpus N` itself (plain child processes); no torch in any rank — take contiguous runs of the pair list, every set is parsed by
    one rank only and reaches the others device to device (the default: commet_readset_export / _import, HIP IPC handles of the
    owner's buffers, the first real set through a fresh canary process first) or as a packed image (COMMET_MATRIX_IPC=0:
    commet_readset_save / _load; "canary-killed": the canary is given no time, so every taker asks the owners for images in
    mid-run); the ranks write their .bv files side by side and rank 0 assembles the matrices.  All ranks use GPU 0 here
    (COMMET_FORCE_DEVICE); outputs must equal Commet.py's."""
    import subprocess
    gold = os.path.join(GOLD, "abcde", "commet_py", "five_sets")
    monkeypatch.chdir(abcde)
    open("sets.txt", "w").write(open(os.path.join(gold, "sets.txt")).read())
    env = dict(os.environ, COMMET_FORCE_DEVICE="0", PYTHONPATH=ROOT, COMMET_MATRIX_REPORT="report.json")
    env.pop("COMMET_MATRIX_IPC", None)
    if handover == "image":
        env["COMMET_MATRIX_IPC"] = "0"
    if handover == "canary-killed":
        env["COMMET_IPC_CANARY_S"] = "0.001"
    args = ["-m", "commet_amd.matrix", "sets.txt", "-k", "32", "-t", "2", "-o", "out2/"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(29600 + (os.getpid() + world) % 300)] + args
    else:
        cmd = [sys.executable] + args + ["--gpus", str(world)]
        env = {k_: v for k_, v in env.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    for f in sorted(os.listdir(gold)):
        if f.endswith((".csv", ".bv")):
            assert open(os.path.join("out2", f), "rb").read() == open(os.path.join(gold, f), "rb").read(), f
    import json
    rep = json.load(open("report.json"))
    takers = [r for r in rep["per_rank"] if r["sets_loaded"] > 0]
    assert takers and {r["handover"] for r in takers} == {"ipc" if handover == "ipc" else "image"}   # (the probe between the ranks passed)
    if handover == "ipc":
        assert {r["ipc_canary"] for r in takers} == {"passed"}
    if handover == "canary-killed":
        assert all(r["ipc_canary"].startswith("failed (no answer") for r in takers)
        assert sum(r["save_s"] > 0 for r in rep["per_rank"]) > 0                                    # owners wrote images on request
    assert all(r["backend"] == "tcp" and r["torch_loaded"] is False for r in rep["per_rank"])
    assert sum(r["sets_parsed"] for r in rep["per_rank"]) == 5 and sum(r["sets_loaded"] for r in rep["per_rank"]) > 0
    os.remove("report.json")


def test_exported_set_is_imported_by_another_process(tmp_path):
    """commet_readset_export / _import across two processes on one device: the importer's copy gives the same per-file
    read counts, k-mer counts (at ITS k) and job results as a set it parses itself; ragged reads (the offsets travel too)"""
    import subprocess
    import commet_amd
    import util
    rng = np.random.default_rng(21)
    reads = util.random_reads(rng, 5000, 5, 260, n_rate=0.02)
    b, o = util.to_batch(reads)
    np.save(tmp_path / "b.npy", b)
    np.save(tmp_path / "o.npy", o)
    child = f'''
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
import commet_amd
blob = open({str(tmp_path / "set.blob")!r}, "rb").read()
b, o = np.load({str(tmp_path / "b.npy")!r}), np.load({str(tmp_path / "o.npy")!r})
with commet_amd.Context(k=20, t=2) as ctx:
    got = commet_amd.ReadSet.import_(ctx, blob)
    own = commet_amd.ReadSet.from_files(ctx, [(b[: int(o[3000])], o[:3001]), (b[int(o[3000]):], o[3000:] - o[3000])])
    assert got.file_reads() == own.file_reads() == [3000, 2000], got.file_reads()
    assert np.array_equal(got.kmer_counts(), own.kmer_counts())
    q = commet_amd.ReadSet.from_files(ctx, [(b[: int(o[1500])], o[:1501])])
    r1, r2 = ctx.index_and_search(got, [q]), ctx.index_and_search(own, [q])
    assert np.array_equal(r1[0][0], r2[0][0]) and r1[1][0]["shared"] == r2[1][0]["shared"] > 1000
    r3, r4 = ctx.index_and_search(q, [got]), ctx.index_and_search(q, [own])
    assert np.array_equal(r3[0][0], r4[0][0])
    from commet_amd import matrix
    eng = matrix.HipEngine.__new__(matrix.HipEngine)          # the driver's probe check: the packed images are the same bytes
    assert eng.same_set(got, own) and not eng.same_set(got, q)
print("imported ok")
'''
    with commet_amd.Context(k=31, t=2) as ctx:                            # (the exporter's k plays no role)
        rs = commet_amd.ReadSet.from_files(ctx, [(b[: int(o[3000])], o[:3001]), (b[int(o[3000]):], o[3000:] - o[3000])])
        open(tmp_path / "set.blob", "wb").write(rs.export())
        p = subprocess.run([sys.executable, "-c", child], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert p.returncode == 0 and b"imported ok" in p.stdout, p.stdout.decode()[-2000:]
        with pytest.raises(commet_amd.CommetError):
            commet_amd.ReadSet.import_(ctx, b"not a descriptor")


def test_packed_image_round_trip(tmp_path):
    """commet_readset_save / _load: a set parsed from three files (ragged reads, N, lowercase, FASTQ + gzip) and its
    packed image loaded into a context with ANOTHER k give the same per-file read counts, k-mer counts and job results"""
    import commet_amd
    import util
    rng = np.random.default_rng(9)
    reads = [util.random_reads(rng, n, 5, 260, n_rate=0.02) for n in (700, 1, 1300)]
    files = []
    for i, (r, fmt) in enumerate(zip(reads, ("fa", "fq.gz", "fa"))):
        p = str(tmp_path / f"s{i}.{fmt}")
        util.write_reads(p, r, fmt, rng=rng, multiline=(i == 2))
        files.append(p)
    q = util.related_reads(rng, [x for r in reads for x in r], 1500, 20, 200, share=0.6)
    qb, qo = util.to_batch(q)
    with commet_amd.Context(k=25, t=2) as c1:
        a = commet_amd.ReadSet.from_fasta(c1, files)
        a.save(str(tmp_path / "a.pk"))
        counts, kc25 = a.file_reads(), a.kmer_counts()
        b = commet_amd.ReadSet.load(c1, str(tmp_path / "a.pk"))
        assert b.file_reads() == counts and np.array_equal(b.kmer_counts(), kc25)
        qs = commet_amd.ReadSet.from_files(c1, [(qb, qo)])
        ra, rb = c1.index_and_search(a, [qs]), c1.index_and_search(b, [qs])
        assert np.array_equal(ra[0][0], rb[0][0]) and ra[1][0]["shared"] == rb[1][0]["shared"] > 100
        rq = c1.index_and_search(qs, [b])                      # the loaded set as the search set
        assert np.array_equal(rq[0][0], c1.index_and_search(qs, [a])[0][0])
    with commet_amd.Context(k=16, t=1) as c2:                   # the image does not depend on k
        a = commet_amd.ReadSet.from_fasta(c2, files)
        b = commet_amd.ReadSet.load(c2, str(tmp_path / "a.pk"))
        assert np.array_equal(a.kmer_counts(), b.kmer_counts()) and not np.array_equal(b.kmer_counts(), kc25)
        qs = commet_amd.ReadSet.from_files(c2, [(qb, qo)])
        assert np.array_equal(c2.index_and_search(a, [qs])[0][0], c2.index_and_search(b, [qs])[0][0])
    with commet_amd.Context(k=16, t=1) as c3, pytest.raises(commet_amd.CommetError):
        commet_amd.ReadSet.load(c3, files[0])                  # not a packed image
