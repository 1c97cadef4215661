#!/usr/bin/env python3
"""Second word list for tests/test_policy_surface.py -- the CONVERSE of reference_member_names.json: every member the reference's
callers use on the policy objects (`detector.` / `matcher.` / `localizer.` / `robustMatcher.` / `filter.` / `covIntOptimizer.` /
`logger.` in include/coloc/{coloc,colocInterface,InterfaceDisk,InterfaceROS}.hpp, comments and strings removed).  A drop-in header must
declare each of them (or INTEGRATION.md 4c says why not).  Output: tests/golden/reference_policy_calls.json ({object: [members]} --
identifiers, not source text).  Needs /root/reference; the test runs from the committed list."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("COLOC_REFERENCE", "/root/reference")
OBJECTS = ["detector", "matcher", "localizer", "robustMatcher", "filter", "covIntOptimizer", "logger"]
FILES = ["coloc.hpp", "colocInterface.hpp", "InterfaceDisk.hpp", "InterfaceROS.hpp"]


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r'"(\\.|[^"\\])*"', '""', text)


def policy_calls(ref=REF):
    calls = {o: set() for o in OBJECTS}
    for f in FILES:
        text = strip_comments(open(os.path.join(ref, "include", "coloc", f), errors="replace").read())
        for m in re.finditer(r"\b(%s)\s*(?:\.|->)\s*([A-Za-z_]\w*)" % "|".join(OBJECTS), text):
            calls[m.group(1)].add(m.group(2))
    return {o: sorted(v) for o, v in calls.items()}


def main():
    out = {"source": "members used on the policy objects in include/coloc/{%s} (comments and strings removed)" % ",".join(FILES),
           "calls": policy_calls()}
    dst = os.path.join(ROOT, "tests", "golden", "reference_policy_calls.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst, {k: len(v) for k, v in out["calls"].items()})


if __name__ == "__main__":
    main()
