"""A CPU engine for commet_amd.matrix.run built on the CPU checker (oracle/), so that the multi-rank HOST logic of the
N x N driver (pair runs, one parse per set, packed images handed between ranks, gathers, failure handling) can run
under several processes without a GPU.  TEST INFRASTRUCTURE ONLY: the package itself has one engine, HipEngine."""
import os
import pickle
import shutil
import tempfile

import numpy as np

import oracle_binding as ob
import util


class OracleEngine:
    fail_on_rank = None          # tests: the rank whose parse() raises

    def __init__(self, k, t, local_rank):
        self.k, self.t, self.rank = k, t, int(os.environ.get("RANK", "0"))
        self.work = tempfile.mkdtemp(prefix="commet_orc_")
        self.njobs = 0

    def parse(self, files):
        if OracleEngine.fail_on_rank is not None and self.rank == OracleEngine.fail_on_rank:
            raise RuntimeError("injected failure while parsing")
        return dict(files=[os.path.abspath(f) for f in files], counts=[len(util.parse_reads(f)) for f in files])

    def save(self, rs, path):
        with open(path + ".tmp", "wb") as fh:
            pickle.dump(rs, fh)
        os.rename(path + ".tmp", path)

    def load(self, path):
        with open(path, "rb") as fh:
            return pickle.load(fh)

    # the file-less hand-over of the driver (HipEngine: HIP IPC handles of device buffers; here: the descriptor itself)
    ipc = True                   # tests: False = this engine has no export (the driver then takes packed images)

    def parse_probe(self):
        if not OracleEngine.ipc:
            raise RuntimeError("no device-to-device hand-over in this engine")
        return dict(files=[], counts=[])

    def export_set(self, rs):
        if not OracleEngine.ipc:
            raise RuntimeError("no device-to-device hand-over in this engine")
        return pickle.dumps(rs)

    def import_set(self, blob):
        if os.environ.get("COMMET_TEST_IMPORT_HANG") == str(self.rank) and pickle.loads(blob)["files"]:   # (not the probe set)
            import time
            time.sleep(600)                      # a HIP call that never returns
        return pickle.loads(blob)

    # the driver's canary (a fresh child process that imports the first real set before a rank does): tests choose its fate
    canary = os.environ.get("COMMET_TEST_CANARY")     # None: no canary command; "ok" / "fail" / "hang"

    def __getattr__(self, name):
        if name == "canary_argv" and OracleEngine.canary:
            import sys
            code = {"ok": "import sys; sys.exit(0)", "fail": "import sys; sys.exit(1)", "hang": "import time; time.sleep(600)"}[OracleEngine.canary]
            return lambda scratch, candidates: [sys.executable, "-c", code]
        raise AttributeError(name)

    def file_reads(self, rs):
        return list(rs["counts"])

    def release(self, rs):
        pass

    def _cfg(self, name, rs, sel, d):
        parts, pos = [], 0
        for j, (f, c) in enumerate(zip(rs["files"], rs["counts"])):
            if sel is None:
                parts.append(f)
            else:
                bools = np.unpackbits(np.asarray(sel, dtype=np.uint8), bitorder="little")[pos:pos + c].astype(bool)
                bv = os.path.join(d, f"{name}_{j}.bv")
                util.write_bv(bv, "sel", bools)
                parts.append(f + "," + bv)
            pos += c
        return name + ":" + ";".join(parts)

    def index_and_search(self, index, searches, isel, ssels):
        self.njobs += 1
        d = os.path.join(self.work, f"job{self.njobs}")
        os.makedirs(d)
        ssels = ssels or [None] * len(searches)
        # distinct basenames per search set: the tool names its outputs after them
        links = []
        for q, rs in enumerate(searches):
            fl = []
            for j, f in enumerate(rs["files"]):
                ln = os.path.join(d, f"q{q:03d}_{j:03d}.fa")
                os.symlink(f, ln)
                fl.append(ln)
            links.append(dict(files=fl, counts=rs["counts"]))
        open(os.path.join(d, "i.txt"), "w").write(self._cfg("I", index, isel, d) + "\n")
        open(os.path.join(d, "s.txt"), "w").write("".join(self._cfg(f"Q{q:03d}", rs, ssels[q], d) + "\n" for q, rs in enumerate(links)))
        rc, res, chunks, kmers = ob.index_and_search(os.path.join(d, "i.txt"), os.path.join(d, "s.txt"), os.path.join(d, "out"),
                                                     os.path.join(d, "log"), self.k, self.t)
        if rc != 0:
            raise RuntimeError("CPU checker failed")
        by = {r["name"]: r for r in res}
        tags, stats = [], []
        for q, rs in enumerate(links):
            bools = []
            for f, c in zip(rs["files"], rs["counts"]):
                _, n, bits = util.read_bv(os.path.join(d, "out", os.path.basename(f) + "_in_I.bv"))
                assert n == c
                bools.append(util.bools_from_bits(bits, n))
            tags.append(util.bits_from_bools(np.concatenate(bools)))
            r = by[f"Q{q:03d}"]
            stats.append(dict(indexed=r["indexed"], searched=r["searched"], shared=r["shared"], search_ms=0.0))
        shutil.rmtree(d, ignore_errors=True)
        info = dict(n_chunks=chunks, kmers_indexed=kmers, total_ms=0.0, index_ms=0.0, search_ms=0.0)
        return tags, stats, info

    def synchronize(self):
        pass

    def close(self):
        shutil.rmtree(self.work, ignore_errors=True)

    def mismatch_error(self, msg):
        return RuntimeError(msg)
