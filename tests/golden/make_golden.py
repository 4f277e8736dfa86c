#!/usr/bin/env python3
"""Generates the committed golden vectors from the REFERENCE itself.

Run in the build container only (needs /root/reference and oracle/_ref, i.e.
`make -C oracle ref`).  Everything written here is DATA: tiny inputs and the
reference's outputs for them; no reference source text is stored.

  scenarios/sNN/      inputs of a randomised scenario (tests/scenarios.py, seed NN)
                      + expected/: the .bv files and the 4th log line the
                      reference's index_and_search produced for them
  keys_kat.json       (read, k) -> HashKey values, forward (add) and reverse
                      (rv_add), from a driver around the reference's hash_key.h
  abcde/              the reference's own smoke dataset (ABCDE_bench/*.fa, B==D and
                      C==E are byte-identical so only A,B,C are stored, gzipped) and
                      the reference's outputs for the 3-set config it ships
                      (sets_config.txt) and the 5-set config of BASELINE config[0],
                      k=32 t=2, run through the job sequence of Commet.py:570-574.
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.environ.get("COMMET_REFERENCE", "/root/reference")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "index_and_search")

import numpy as np  # noqa: E402

from scenarios import Scenario, run_tool  # noqa: E402

SCENARIO_SEEDS = list(range(1000, 1024))
MIXED_FORMAT_SEEDS = set(range(1016, 1024))     # FASTQ / gzip inputs
FULL_MODE_SEEDS = set(range(1000, 1024, 2))     # also run with -f


def make_scenarios():
    base = os.path.join(HERE, "scenarios")
    shutil.rmtree(base, ignore_errors=True)
    meta = {}
    for seed in SCENARIO_SEEDS:
        d = os.path.join(base, f"s{seed}")
        scn = Scenario(d, seed, n_scale=0.6, **({"formats": ("fa", "fq", "fa.gz", "fq.gz"), "crlf": False}
                                               if seed in MIXED_FORMAT_SEEDS else {}))
        p = run_tool(REF_BIN, scn, "expected", "expected_log")
        assert p.returncode == 0, p.stderr
        lines = {}
        for f in sorted(os.listdir(os.path.join(d, "expected_log"))):
            lines[f] = open(os.path.join(d, "expected_log", f)).read().strip().split("\n")[-1]
        shutil.rmtree(os.path.join(d, "expected_log"))
        with open(os.path.join(d, "expected", "log_lines.json"), "w") as fh:
            json.dump(lines, fh, indent=1, sort_keys=True)
        if seed in FULL_MODE_SEEDS:      # the -f (full comparison) mode of the reference, SURVEY 8f-4
            p = run_tool(REF_BIN, scn, "expected_full", "expected_full_log", extra_args=["-f"])
            assert p.returncode == 0, p.stderr
            lines = {}
            for f in sorted(os.listdir(os.path.join(d, "expected_full_log"))):
                lines[f] = open(os.path.join(d, "expected_full_log", f)).read().strip().split("\n")[3:]
            shutil.rmtree(os.path.join(d, "expected_full_log"))
            with open(os.path.join(d, "expected_full", "log_lines.json"), "w") as fh:
                json.dump(lines, fh, indent=1, sort_keys=True)
        meta[f"s{seed}"] = dict(seed=seed, k=scn.k, t=scn.t, index=scn.index_name, search=scn.search_names)
    with open(os.path.join(base, "index.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)


KAT_DRIVER = r"""
#include <cstdio>
#include <cstring>
#include <string>
#include "hash_key.h"
#include "alphabet.h"
int main(int argc, char **argv) {
    int k = atoi(argv[1]); int rev = atoi(argv[2]); std::string s = argv[3];
    HashKey h(k); Alphabet *al = Alphabet::getInstance();
    for (size_t i = 0; i < s.size(); i++) {
        if (!al->is_in(s[i])) { h.clear(); continue; }
        char c = s[i];
        int sz = rev ? h.rv_add(c) : h.add(c);
        if (sz >= k) printf("%zu %lu %lu %lu %lu\n", i, h.keya(), h.keyb(), h.keyc(), h.keyd());
    }
    return 0;
}
"""


def make_keys_kat():
    tmp = tempfile.mkdtemp()
    src = os.path.join(tmp, "kat.cpp")
    open(src, "w").write(KAT_DRIVER)
    exe = os.path.join(tmp, "kat")
    subprocess.run(["g++", "-O1", "-w", "-I", os.path.join(REF, "include"), "-o", exe, src], check=True)
    rng = np.random.default_rng(7)
    cases = []
    alphabet = np.frombuffer(b"ACGTacgtNRY", dtype=np.uint8)
    probs = np.array([.2, .2, .2, .2, .04, .04, .04, .04, .02, .01, .01])
    for k in (1, 2, 5, 8, 13, 20, 21, 31, 32, 33, 34, 40):
        for L in (0, 1, k - 1 if k > 1 else 1, k, k + 1, 60, 150):
            seq = alphabet[rng.choice(len(alphabet), size=L, p=probs)].tobytes().decode()
            for rev in (0, 1):
                out = subprocess.run([exe, str(k), str(rev), seq], check=True, stdout=subprocess.PIPE).stdout.decode()
                rows = [[int(x) for x in line.split()] for line in out.strip().split("\n") if line]
                cases.append(dict(k=k, reverse=rev, seq=seq, rows=rows))
    with open(os.path.join(HERE, "keys_kat.json"), "w") as fh:
        json.dump(cases, fh)
    shutil.rmtree(tmp)


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def commet_jobs(names):
    """the N^2-1 index_and_search invocations of Commet.py (compare_all_against, Commet.py:186-240)
    as (kind, index set, [search sets], restriction of the index set or None)"""
    jobs = []
    n = len(names)
    for ref in range(n - 1):
        jobs.append(("J1", ref, list(range(ref + 1, n)), None))
        for i in range(ref + 1, n):
            jobs.append(("J2", i, [ref], ref))     # index S_i restricted by <F>_in_<S_ref>.bv, search S_ref
            jobs.append(("J3", ref, [i], i))       # index S_ref restricted by <G>_in_<S_i>.bv, search S_i
    return jobs


def run_commet_matrix(workdir, sets, k, t):
    """sets: list of (name, [fasta paths relative to workdir]).  Drives oracle/_ref/index_and_search through
    Commet.py's job sequence with unfiltered inputs (all-ones filter bvs are implied by omitting them)."""
    names = [s[0] for s in sets]
    out = "out"
    os.makedirs(os.path.join(workdir, out), exist_ok=True)

    def cfg_line(si, restrict_to=None):
        name, files = sets[si]
        parts = []
        for f in files:
            if restrict_to is None:
                parts.append(f)
            else:
                parts.append(f + "," + out + "/" + os.path.basename(f) + "_in_" + names[restrict_to] + ".bv")
        return name + ":" + ";".join(parts)

    for n_job, (kind, idx, searches, restr) in enumerate(commet_jobs(names)):
        icfg = f"job{n_job}_index.txt"
        scfg = f"job{n_job}_search.txt"
        open(os.path.join(workdir, icfg), "w").write(cfg_line(idx, restr) + "\n")
        open(os.path.join(workdir, scfg), "w").write("".join(cfg_line(s) + "\n" for s in searches))
        subprocess.run([REF_BIN, "-i", icfg, "-s", scfg, "-o", out, "-l", out, "-k", str(k), "-t", str(t)],
                       cwd=workdir, check=True, stdout=subprocess.DEVNULL)
    return out


def make_abcde():
    dst = os.path.join(HERE, "abcde")
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    for f in "ABC":
        with open(os.path.join(REF, "ABCDE_bench", f + ".fa"), "rb") as fi, \
                gzip.GzipFile(os.path.join(dst, f + ".fa.gz"), "wb", compresslevel=9, mtime=0) as fo:
            fo.write(fi.read())
    result = {}
    for label, sets in (
        ("three_sets", [("set1", ["ABCDE_bench/A.fa"]), ("set2", ["ABCDE_bench/B.fa", "ABCDE_bench/C.fa"]),
                        ("set3", ["ABCDE_bench/D.fa"])]),
        ("five_sets", [(x, [f"ABCDE_bench/{x}.fa"]) for x in "ABCDE"]),
    ):
        work = tempfile.mkdtemp()
        os.symlink(os.path.join(REF, "ABCDE_bench"), os.path.join(work, "ABCDE_bench"))
        out = run_commet_matrix(work, sets, 32, 2)
        od = os.path.join(dst, label)
        os.makedirs(od)
        entry = {}
        for f in sorted(os.listdir(os.path.join(work, out))):
            if f.endswith(".bv"):
                shutil.copy(os.path.join(work, out, f), os.path.join(od, f))
                entry[f] = sha(os.path.join(od, f))
        result[label] = dict(sets=sets, k=32, t=2, sha256=entry)
        shutil.rmtree(work)
    with open(os.path.join(dst, "expected.json"), "w") as fh:
        json.dump(result, fh, indent=1, sort_keys=True)


def make_commet_py():
    """Runs the reference's own driver (Commet.py, python3) with the reference binaries on ABCDE_bench and keeps
    what it produces: the three matrix CSVs, the filter .bv files and every *_in_*.bv (k=32, t=2, defaults)."""
    dst = os.path.join(HERE, "abcde", "commet_py")
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    for label, lines in (("three_sets", ["set1: ABCDE_bench/A.fa", "set2: ABCDE_bench/B.fa; ABCDE_bench/C.fa",
                                         "set3: ABCDE_bench/D.fa"]),
                         ("five_sets", [f"{x}: ABCDE_bench/{x}.fa" for x in "ABCDE"])):
        work = tempfile.mkdtemp()
        os.symlink(os.path.join(REF, "ABCDE_bench"), os.path.join(work, "ABCDE_bench"))
        for f in ("dendro.R", "heatmap.r"):
            os.symlink(os.path.join(REF, f), os.path.join(work, f))
        open(os.path.join(work, "sets.txt"), "w").write("\n".join(lines) + "\n")
        subprocess.run([sys.executable, os.path.join(REF, "Commet.py"), "sets.txt", "-b",
                        os.path.join(ROOT, "oracle", "_ref") + "/", "-o", "out/", "-k", "32", "-t", "2"],
                       cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        od = os.path.join(dst, label)
        os.makedirs(od)
        shutil.copy(os.path.join(work, "sets.txt"), od)
        for f in sorted(os.listdir(os.path.join(work, "out"))):
            if f.endswith(".csv") or f.endswith(".bv"):
                shutil.copy(os.path.join(work, "out", f), od)
        shutil.rmtree(work)


if __name__ == "__main__":
    what = sys.argv[1:] or ["scenarios", "keys", "abcde", "commetpy"]
    if "commetpy" in what:
        make_commet_py()
    if "scenarios" in what:
        make_scenarios()
    if "keys" in what:
        make_keys_kat()
    if "abcde" in what:
        make_abcde()
