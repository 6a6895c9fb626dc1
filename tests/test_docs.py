"""The documents against the files they cite (DESIGN.md quoted numbers its cited profiles did not hold)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_numbers_block_is_what_the_committed_profiles_say():
    """DESIGN.md section 0's numbers block must be exactly what tools/design_numbers.py prints from profiles/ for the newest round."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_numbers.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"<!-- numbers:begin -->\n(.*?)\n<!-- numbers:end -->", text, flags=re.S)
    assert m, "DESIGN.md has no numbers block"
    assert m.group(1).strip() == out.stdout.strip(), "run: python tools/design_numbers.py --write"


def test_readme_measured_paragraph_is_generated_from_the_committed_profiles():
    """README.md's "Measured" paragraph was typed by hand until round 5 and went stale: it is a generated block now."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_numbers.py"), "--readme"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    text = open(os.path.join(ROOT, "README.md")).read()
    m = re.search(r"<!-- measured:begin -->\n(.*?)\n<!-- measured:end -->", text, flags=re.S)
    assert m, "README.md has no measured block"
    assert m.group(1).strip() == out.stdout.strip(), "run: python tools/design_numbers.py --write"


def test_every_tool_and_profile_the_current_documents_cite_exists():
    """README / DESIGN / INTEGRATION / tools/README name scripts and profile files; a renamed or deleted one must not stay cited.  (The notebooks
    under docs/ and the profiles themselves are history and cite scripts by the names they had.)"""
    missing = []
    for doc in ("README.md", "DESIGN.md", "INTEGRATION.md", os.path.join("tools", "README.md")):
        text = open(os.path.join(ROOT, doc)).read()
        for path in set(re.findall(r"`((?:tools|profiles|tests|docs|oracle|include|armour_amd)/[A-Za-z0-9_./-]+\.[a-z0-9]+)", text)):
            if "*" in path or path.endswith((".so", ".o")) or "/lib/" in path or "/bin/" in path or "_ref/" in path:
                continue
            if not os.path.exists(os.path.join(ROOT, path)):
                missing.append((doc, path))
    assert not missing, missing
