#!/usr/bin/env python3
"""Fixture generator (authoring container only; reads /root/reference).  Extracts the LITERAL known-answer vectors of
the reference's replacement-template tests (replace/template_test.go: TestParse, TestValidateAndResolve,
TestValidateAndResolve_ResolvesNames) into tests/golden/replace_kats.json -- inputs and expected values only."""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "replace_kats.json")
STR = r'"((?:[^"\\]|\\.)*)"'
TYPES = {"SegmentLiteral": "lit", "SegmentFullMatch": "full", "SegmentCaptureIndex": "idx", "SegmentCaptureName": "name"}


def unq(s):
    return json.loads('"' + s + '"')


def blocks(body):
    """top-level `{ ... }` struct literals of a test table"""
    out, depth, start = [], 0, None
    i, n = -1, len(body)
    while i + 1 < n:
        i += 1
        ch = body[i]
        if ch == '"':                      # skip Go string literals: their braces are data
            i += 1
            while body[i] != '"':
                i += 2 if body[i] == "\\" else 1
            continue
        if ch == "{":
            if depth == 0:
                start = i
            depth += 1
        elif ch == "}":
            depth -= 1
            if depth == 0:
                out.append(body[start:i + 1])
    return out


def table(src, func):
    m = re.search(r"^func %s\(t \*testing\.T\) \{$" % func, src, re.M)
    body = src[m.end():]
    t0 = body.index("tests := []struct")
    i = body.index("}{", t0) + 1          # the `{` that opens the table literal
    return blocks(body[i:])[0][1:-1]


def segs(txt):
    out = []
    for b in blocks(txt):
        ty = TYPES[re.search(r"Type:\s*(Segment\w+)", b).group(1)]
        s = {"type": ty}
        m = re.search(r"Literal:\s*" + STR, b)
        if m:
            s["literal"] = unq(m.group(1))
        m = re.search(r"CaptureIndex:\s*(\d+)", b)
        if m:
            s["index"] = int(m.group(1))
        m = re.search(r"CaptureName:\s*" + STR, b)
        if m:
            s["name"] = unq(m.group(1))
        out.append(s)
    return out


src = open(REF + "/replace/template_test.go").read()
k = {"parse": [], "validate": []}
for b in blocks(table(src, "TestParse")):
    case = {"name": unq(re.search(r"name:\s*" + STR, b).group(1)), "template": unq(re.search(r"template:\s*" + STR, b).group(1)),
            "want_err": bool(re.search(r"wantErr:\s*true", b))}
    m = re.search(r"wantSegs:\s*\[\]Segment\{", b)
    if m:
        inner = b[m.end() - 1:]
        case["segments"] = segs(blocks(inner)[0][1:-1])
    k["parse"].append(case)
for b in blocks(table(src, "TestValidateAndResolve")):
    names = {}
    m = re.search(r"captureNames:\s*map\[string\]int\{([^}]*)\}", b)
    if m:
        for nm, ix in re.findall(STR + r":\s*(\d+)", m.group(1)):
            names[unq(nm)] = int(ix)
    k["validate"].append({"name": unq(re.search(r"name:\s*" + STR, b).group(1)), "template": unq(re.search(r"template:\s*" + STR, b).group(1)),
                          "capture_names": names, "num_captures": int(re.search(r"numCaptures:\s*(\d+)", b).group(1)),
                          "want_err": bool(re.search(r"wantErr:\s*true", b))})
json.dump(k, open(OUT, "w"), indent=1, ensure_ascii=False)
print(len(k["parse"]), "parse cases,", len(k["validate"]), "validate cases ->", OUT)
