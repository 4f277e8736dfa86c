#!/usr/bin/env python3
"""doc_refs.py — keeps the `symbol` (`file:line`) references of DESIGN.md true.

A reference is a symbol in backticks followed by a parenthesised `path:line` in backticks.  --check: every referenced line must hold
the symbol's last component (tests/test_design_refs.py runs this); --fix: rewrite the line numbers to where the symbol is DEFINED now
(the first line that holds it together with one of `(`, `struct`, `class`, `constexpr`, `def `, `inline`, `template`, else its first mention)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASES = ("", "commet_amd/csrc")          # a path is taken from the repo root, else from the library's source directory (DESIGN.md says so)
REF = re.compile(r"`([A-Za-z_][\w:<>, ]*?)`\s*\(`([\w/.]+\.(?:hpp|h|hip|cpp|py|c)):(\d+)`\)")


def resolve(path):
    for b in BASES:
        full = os.path.join(ROOT, b, path)
        if os.path.exists(full):
            return full
    return None


def _name(sym):
    return sym.split("::")[-1].split("<")[0].strip()


def find_line(path, sym):
    lines = open(resolve(path), errors="replace").read().split("\n")
    name = _name(sym)
    word = re.compile(r"\b" + re.escape(name) + r"\b")
    defs = [i for i, ln in enumerate(lines, 1) if word.search(ln) and re.search(r"(\bstruct\b|\bclass\b|\bconstexpr\b|\bdef \b|\binline\b|__global__|^\w[\w:<>\*& ]* \*?" + re.escape(name) + r"\()", ln)
            and not ln.lstrip().startswith(("//", "*", "#"))]
    if defs:
        return defs[0]
    # a kernel's name usually sits on the line after `template <...>` / `__global__ ... void name(`
    alls = [i for i, ln in enumerate(lines, 1) if word.search(ln) and not ln.lstrip().startswith(("//", "*"))]
    return alls[0] if alls else None


def check(doc):
    text = open(os.path.join(ROOT, doc)).read()
    bad = []
    for m in REF.finditer(text):
        sym, path, line = m.group(1), m.group(2), int(m.group(3))
        full = resolve(path)
        if full is None:
            bad.append(f"{path}: no such file ({sym})")
            continue
        lines = open(full, errors="replace").read().split("\n")
        if not (1 <= line <= len(lines)) or not re.search(r"\b" + re.escape(_name(sym)) + r"\b", lines[line - 1]):
            bad.append(f"{path}:{line} does not hold `{sym}`")
    return bad, len(REF.findall(text))


def fix(doc):
    p = os.path.join(ROOT, doc)
    text = open(p).read()

    def sub(m):
        sym, path = m.group(1), m.group(2)
        if resolve(path) is None:
            return m.group(0)
        ln = find_line(path, sym)
        return m.group(0) if ln is None else m.group(0).replace(f"{path}:{m.group(3)}", f"{path}:{ln}")

    open(p, "w").write(REF.sub(sub, text))


if __name__ == "__main__":
    docs = [a for a in sys.argv[1:] if not a.startswith("--")] or ["DESIGN.md"]
    for d in docs:
        if "--fix" in sys.argv:
            fix(d)
        bad, n = check(d)
        print(f"{d}: {n} references, {len(bad)} stale")
        for b in bad:
            print("  " + b)
    sys.exit(0)
