"""DESIGN.md's `symbol` (`file:line`) references must point at lines that hold the symbol (tools/doc_refs.py --fix rewrites them)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_md_references_hold_their_symbols():
    spec = importlib.util.spec_from_file_location("doc_refs", os.path.join(ROOT, "tools", "doc_refs.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    bad, n = m.check("DESIGN.md")
    assert n >= 40, "DESIGN.md lost its file:line references"
    assert not bad, "stale references in DESIGN.md (python tools/doc_refs.py --fix): " + "; ".join(bad)
    assert os.path.getsize(os.path.join(ROOT, "DESIGN.md")) <= 15 * 1024, "DESIGN.md is the mechanism only: measurements go to MEASUREMENTS.md"
