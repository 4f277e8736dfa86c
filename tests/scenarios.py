"""Randomised index_and_search scenarios covering the reference's corner cases
(SURVEY §7 step 0): N / lowercase / IUPAC / multi-line FASTA / CRLF / multi-file
sets / input filter bvs incl. all-zero ones / reads shorter than k / small k
forcing many chunks / t = 1..4 / reverse-complement-only hits."""
import os

import numpy as np

import util


class Scenario:
    def __init__(self, d, seed, k=None, t=None, n_scale=1.0, allow_bv=True, allow_zero_bv=True, crlf=None, formats=("fa",)):
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
                fmt = formats[int(rng.integers(0, len(formats)))] if len(formats) > 1 else formats[0]
                fa = f"{name}_f{fi}.{fmt}"       # relative: tools run with cwd = scenario dir
                util.write_reads(os.path.join(d, fa), reads, fmt, rng=rng, multiline=multiline, crlf=crlf)
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
                    util.write_bv(os.path.join(d, bv), f"filter of {fa}", sel)
                files.append((fa, bv, reads, sel))
            self.sets[name] = files
        self.index_name = names[0]
        self.search_names = names[1:]
        self.index_cfg = "index.txt"
        self.search_cfg = "search.txt"
        with open(os.path.join(d, self.index_cfg), "w") as fh:
            fh.write(self._line(self.index_name) + "\n")
        with open(os.path.join(d, self.search_cfg), "w") as fh:
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


def run_tool(tool, scn, out, log, extra_env=None, extra_args=()):
    """runs an index_and_search-compatible CLI inside the scenario dir"""
    env = dict(os.environ)
    if extra_env:
        env.update(extra_env)
    import subprocess
    return subprocess.run([tool, "-i", scn.index_cfg, "-s", scn.search_cfg, "-o", out, "-l", log,
                           "-k", str(scn.k), "-t", str(scn.t)] + list(extra_args), cwd=scn.dir, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def run_oracle(scn, out, log, max_kmer=0):
    """the CPU checker in-process (paths resolved against the scenario dir); max_kmer != 0: chunked like a library context
    with that `max_kmer` option"""
    import oracle_binding as ob
    cwd = os.getcwd()
    os.chdir(scn.dir)
    try:
        return ob.index_and_search(scn.index_cfg, scn.search_cfg, out, log, scn.k, scn.t, max_kmer=max_kmer)
    finally:
        os.chdir(cwd)


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


class GoldenScenario:
    """A committed scenario of tests/golden/scenarios (inputs + the reference's outputs)."""

    def __init__(self, name):
        import json
        base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenarios")
        meta = json.load(open(os.path.join(base, "index.json")))[name]
        self.dir = os.path.join(base, name)
        self.k, self.t = meta["k"], meta["t"]
        self.index_name = meta["index"]
        self.search_names = meta["search"]
        self.index_cfg, self.search_cfg = "index.txt", "search.txt"
        self.expected_dir = os.path.join(self.dir, "expected")
        self.log_lines = json.load(open(os.path.join(self.expected_dir, "log_lines.json")))
        self.sets = {}
        for cfg in (self.index_cfg, self.search_cfg):
            for line in open(os.path.join(self.dir, cfg)).read().split("\n"):
                if not line:
                    continue
                tag, rest = line.split(":", 1)
                files = []
                for item in rest.split(";"):
                    parts = item.strip(" ").split(",")
                    fa = parts[0].strip(" ")
                    bv = parts[1].strip(" ") if len(parts) > 1 else None
                    reads = util.parse_reads(os.path.join(self.dir, fa))
                    if bv:
                        _, n, bits = util.read_bv(os.path.join(self.dir, bv))
                        sel = util.bools_from_bits(bits, n)
                    else:
                        sel = np.ones(len(reads), dtype=bool)
                    files.append((fa, bv, reads, sel))
                self.sets[tag] = files

    @staticmethod
    def names():
        import json
        base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenarios")
        return sorted(json.load(open(os.path.join(base, "index.json"))).keys())

    def expected_outputs(self):
        return Scenario.expected_outputs(self)

    def check_against_golden(self, out_dir, log_dir):
        bvs, logs = self.expected_outputs()
        for b in bvs:
            got = open(os.path.join(out_dir, b), "rb").read()
            exp = open(os.path.join(self.expected_dir, b), "rb").read()
            assert got == exp, f"{b} differs from the reference's output ({self.dir})"
        for l in logs:
            got = util.last_log_line(os.path.join(log_dir, l))
            assert got == self.log_lines[l], f"{l}: {got!r} != {self.log_lines[l]!r}"
