#!/usr/bin/env python3
"""Fixture generator (authoring container only): the literal TDFA tables of the reference's three checked-in
TDFA matchers (emitted by internal/compiler/tdfa.go:584-794) -> tests/golden/tdfa_tables.json (numbers only)."""
import json, os, re
REF = "/root/reference"
FILES = {"URLCapture": "benchmarks/curated/URLCapture.go", "TDFASemVer": "benchmarks/curated/TDFASemVer.go",
         "IPv4Pattern": "tests/integration/streaming/testdata/ipv4_pattern.go"}
PATTERNS = {"URLCapture": r"(?P<protocol>https?)://(?P<host>[\w\.-]+)(?::(?P<port>\d+))?(?P<path>/[\w\./]*)?",
            "TDFASemVer": r"(?P<major>\d+)\.(?P<minor>\d+)\.(?P<patch>\d+)(?:-(?P<prerelease>[\w.-]+))?(?:\+(?P<build>[\w.-]+))?",
            "IPv4Pattern": r"(\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3})"}

def lit(src, var):
    m = re.search(r"^var %s\s*=\s*[^{]*(\{.*?\})\s*(?://.*)?$" % re.escape(var), src, re.M)
    if not m:
        return None
    t = m.group(1).replace("{", "[").replace("}", "]").replace("true", "True").replace("false", "False")
    return eval(t)

out = {}
for name, rel in FILES.items():
    src = open(os.path.join(REF, rel)).read()
    d = {"pattern": PATTERNS[name]}
    for key, var in (("transitions", "transitions"), ("tag_action_count", "tagActionCount"), ("tag_action_tags", "tagActionTags"),
                     ("tag_action_offsets", "tagActionOffsets"), ("accept", "acceptStates"), ("accept_eot", "acceptStatesEOT"),
                     ("accept_action_count", "acceptActionCount"), ("accept_action_tags", "acceptActionTags"),
                     ("accept_action_offsets", "acceptActionOffsets")):
        d[key] = lit(src, var + name)
    m = re.search(r"if start == 0 \{\s*state = (\d+)(.*?)\} else \{\s*state = (\d+)", src, re.S)
    d["start_begin"], d["start_any"] = int(m.group(1)), int(m.group(3))
    out[name] = d
    print(name, "states", len(d["transitions"]), "start", d["start_begin"], d["start_any"])
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tdfa_tables.json"), "w"))
