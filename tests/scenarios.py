"""Randomised index_and_search scenarios covering the reference's corner cases
(SURVEY §7 step 0): N / lowercase / IUPAC / multi-line FASTA / CRLF / multi-file
sets / input filter bvs incl. all-zero ones / reads shorter than k / small k
forcing many chunks / t = 1..4 / reverse-complement-only hits."""
import os

import numpy as np

import util


class Scenario:
    def __init__(self, d, seed, k=None, t=None, n_scale=1.0, allow_bv=True, allow_zero_bv=True, crlf=None):
        rng = np.random.default_rng(seed)
        self.dir = d
        self.k = int(k if k is not None else rng.choice([8, 10, 12, 13, 16, 20, 25]))
        self.t = int(t if t is not None else rng.integers(1, 5))
        os.makedirs(d, exist_ok=True)
        lo, hi = (5, 90) if rng.random() < 0.7 else (60, 60)
        multiline = rng.random() < 0.4
        crlf = (rng.random() < 0.15) if crlf is None else crlf
        self.sets = {}      # name -> list of (fasta path, bv path or None, reads, select bools)
        pool = []
        n_sets = int(rng.integers(2, 5))
        names = [f"S{i}" for i in range(n_sets)]
        for si, name in enumerate(names):
            files = []
            for fi in range(int(rng.integers(1, 4))):
                n = max(1, int(rng.integers(3, 120) * n_scale))
                reads = util.related_reads(rng, pool, n, lo, hi, share=0.5 if si else 0.0,
                                           n_rate=float(rng.choice([0, 0.01, 0.05])))
                # no empty sequences: the reference's behaviour on them is undefined
                reads = [r if len(r) else b"A" for r in reads]
                pool.extend(reads[: max(1, n // 2)])
                fa = os.path.join(d, f"{name}_f{fi}.fa")
                util.write_fasta(fa, reads, rng=rng, multiline=multiline, crlf=crlf)
                bv = None
                sel = np.ones(n, dtype=bool)
                if allow_bv and rng.random() < 0.5:
                    mode = rng.random()
                    if allow_zero_bv and mode < 0.2:
                        sel = np.zeros(n, dtype=bool)
                    elif mode < 0.4:
                        sel = np.ones(n, dtype=bool)
                    else:
                        sel = rng.random(n) < rng.uniform(0.2, 0.95)
                    bv = fa + ".bv"
                    util.write_bv(bv, f"filter of {fa}", sel)
                files.append((fa, bv, reads, sel))
            self.sets[name] = files
        self.index_name = names[0]
        self.search_names = names[1:]
        self.index_cfg = os.path.join(d, "index.txt")
        self.search_cfg = os.path.join(d, "search.txt")
        with open(self.index_cfg, "w") as fh:
            fh.write(self._line(self.index_name) + "\n")
        with open(self.search_cfg, "w") as fh:
            for nme in self.search_names:
                fh.write(self._line(nme) + "\n")

    def _line(self, name):
        parts = []
        for fa, bv, _, _ in self.sets[name]:
            parts.append(fa + ("," + bv if bv else ""))
        return name + ":" + ";".join(parts)

    def expected_outputs(self):
        """(bv basename, log basename) the tool writes"""
        bvs = []
        for nme in self.search_names:
            for fa, _, _, _ in self.sets[nme]:
                bvs.append(os.path.basename(fa) + "_in_" + self.index_name + ".bv")
        logs = [f"{nme}_in_{self.index_name}.log" for nme in self.search_names]
        return bvs, logs


def compare_runs(out_a, log_a, out_b, log_b, scn):
    bvs, logs = scn.expected_outputs()
    for b in bvs:
        da = open(os.path.join(out_a, b), "rb").read()
        db = open(os.path.join(out_b, b), "rb").read()
        assert da == db, f"{b} differs (k={scn.k} t={scn.t} dir={scn.dir})"
    for l in logs:
        la = util.last_log_line(os.path.join(log_a, l))
        lb = util.last_log_line(os.path.join(log_b, l))
        assert la == lb, f"{l}: {la!r} != {lb!r} (k={scn.k} t={scn.t} dir={scn.dir})"
