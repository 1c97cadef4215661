#!/usr/bin/env python3
"""Word list for tests/test_policy_surface.py: every identifier the reference's OWN sources contain (include/coloc/*.hpp,
src/*.cpp; comments and string literals removed) -- the vocabulary a drop-in header may use on openMVG:: / cv:: / Eigen
objects without inventing API (the reference writes most third-party names unqualified, behind using-directives).
Output: tests/golden/reference_member_names.json (a sorted list of identifiers -- data, not source text).  Needs
/root/reference; the test runs from the committed list."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("COLOC_REFERENCE", "/root/reference")


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r'"(\\.|[^"\\])*"', '""', text)


def member_names(text):
    return set(re.findall(r"(?:\.|->|::)\s*([A-Za-z_]\w*)", strip_comments(text)))


def main():
    names = set()
    files = sorted(glob.glob(os.path.join(REF, "include", "coloc", "*.hpp")) + glob.glob(os.path.join(REF, "include", "coloc", "*.h"))
                   + glob.glob(os.path.join(REF, "src", "*.cpp")))
    for f in files:
        names |= set(re.findall(r"[A-Za-z_]\w*", strip_comments(open(f, errors="replace").read())))
    out = {"source": "identifiers of %d files of include/coloc and src (comments and strings removed)" % len(files), "names": sorted(names)}
    dst = os.path.join(ROOT, "tests", "golden", "reference_member_names.json")
    json.dump(out, open(dst, "w"), indent=0)
    print("wrote", dst, len(names), "names")


if __name__ == "__main__":
    main()
