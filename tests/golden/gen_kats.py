#!/usr/bin/env python3
"""Fixture generator (authoring container only; reads /root/reference).  Extracts the LITERAL known-answer
vectors the reference's own tests hold for the hot path into tests/golden/kats.json -- inputs and expected
values only (no source text):

  min/max match length      internal/compiler/analysis_match_len_test.go:8-187
  DefaultMaxLeftover/MinBuf internal/compiler/analysis_match_len_test.go:189-258
  nested-quantifier flags   internal/compiler/analysis_test.go
  repeating-capture KATs    internal/compiler/repeating_test.go:45-98
  stream.Config defaults    stream/stream_test.go:18-134
  streaming offsets         tests/integration/streaming/streaming_test.go:190-316 (scenario parameters)
  curated test inputs       scripts/curated/cases.go:23-242
"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kats.json")


def go_unquote(s: str) -> str:
    if s.startswith("`"):
        return s[1:-1]
    body = s[1:-1]
    return json.loads('"' + body.replace("\\'", "'") + '"') if "\\x" not in body else bytes(body, "utf-8").decode("unicode_escape")


def func_body(src: str, name: str) -> str:
    m = re.search(r"^func %s\(.*?\) \{$" % re.escape(name), src, re.M)
    start = m.end()
    end = src.index("\n}\n", start)
    return src[start:end]


STR = r'("(?:[^"\\]|\\.)*"|`[^`]*`)'

k = {}
src = open(REF + "/internal/compiler/analysis_match_len_test.go").read()
k["min_match_len"] = [[go_unquote(p), int(v)] for p, v in re.findall(r"\{%s, (-?\d+)\}" % STR, func_body(src, "TestMinMatchLen"))]
k["max_match_len"] = [[go_unquote(p), int(v)] for p, v in re.findall(r"\{%s, (-?\d+)\}" % STR, func_body(src, "TestMaxMatchLen"))]
k["analyze_match_length"] = [[go_unquote(p), int(a), int(b)] for p, a, b in
                             re.findall(r"\{%s, (-?\d+), (-?\d+)\}" % STR, func_body(src, "TestAnalyzeMatchLength"))]


def go_int(expr: str) -> int:
    return int(eval(expr.replace("<<", "<<"), {}))


body = func_body(src, "TestDefaultMaxLeftover")
k["default_max_leftover"] = [[int(a), go_int(b)] for a, b in re.findall(r"MaxMatchLen: (-?\d+)\},\s*want:\s*([^,/]+),", body)]
body = func_body(src, "TestMinBufferSize")
k["min_buffer_size"] = [[int(a), go_int(b)] for a, b in re.findall(r"MaxMatchLen: (-?\d+)\},\s*want:\s*([^,/]+),", body)]

src = open(REF + "/internal/compiler/repeating_test.go").read()
body = func_body(src, "TestRepeatingCapturesBehavior")
kats = []
for m in re.finditer(r"pattern:\s*%s,\s*input:\s*%s,\s*wantFull:\s*%s,\s*wantCaptures:\s*\[\]string\{([^}]*)\}" % (STR, STR, STR), body):
    caps = [go_unquote(x) for x in re.findall(STR, m.group(4))]
    kats.append({"pattern": go_unquote(m.group(1)), "input": go_unquote(m.group(2)), "full": go_unquote(m.group(3)), "captures": caps})
k["repeating_captures"] = kats

src = open(REF + "/internal/compiler/analysis_test.go").read()
k["nested_quantifiers"] = [[go_unquote(p), b == "true"] for p, b in
                           re.findall(r"\{%s, (true|false), %s\}" % (STR, STR), func_body(src, "TestDetectNestedQuantifiers"))
                           ] if False else [[go_unquote(m[0]), m[1] == "true"] for m in
                                            re.findall(r"\{%s, (true|false), %s\}" % (STR, STR), func_body(src, "TestDetectNestedQuantifiers"))]
k["analyze_complexity"] = [[go_unquote(m[0]), m[1] == "true", m[2] == "true"] for m in
                           re.findall(r"\{%s, (true|false), (true|false), %s\}" % (STR, STR), func_body(src, "TestAnalyzeComplexity"))]

# stream.Config: literal cases of Validate / ApplyDefaults (stream/stream_test.go:18-134)
src = open(REF + "/stream/stream_test.go").read()


def cfg_fields(txt):
    d = {"BufferSize": 0, "MaxLeftover": 0}
    for a, b in re.findall(r"(BufferSize|MaxLeftover):\s*([^,}]+)", txt):
        d[a] = go_int(b)
    return d


body = func_body(src, "TestConfigValidate")
k["config_validate"] = [{"cfg": cfg_fields(m[0]), "minBuffer": go_int(m[1]), "wantErr": m[2] == "true"} for m in
                        re.findall(r"cfg:\s*Config\{([^}]*)\},\s*minBuffer:\s*([^,]+),\s*wantErr:\s*(true|false)", body)]
body = func_body(src, "TestConfigApplyDefaults")
k["config_apply_defaults"] = [
    {"cfg": cfg_fields(m[0]), "minBuffer": go_int(m[1]), "defaultLeftover": go_int(m[2]), "wantBufferSize": go_int(m[3]),
     "wantMaxLeftover": go_int(m[4])}
    for m in re.findall(r"cfg:\s*Config\{([^}]*)\},\s*minBuffer:\s*([^,]+),\s*defaultLeftover:\s*([^,]+),(?:\s*//[^\n]*)?\s*"
                        r"wantBufferSize:\s*([^,]+),\s*wantMaxLeftover:\s*([^,]+),", body)]

# streaming boundary scenario (tests/integration/streaming/streaming_test.go:190-280): 100 KiB of 'x' with dates
# planted at literal offsets, streamed with 64 KiB buffers; expected StreamOffsets == the planted offsets.
src = open(REF + "/tests/integration/streaming/streaming_test.go").read()
body = func_body(src, "TestStreamingLargeInputBoundary")
pos = re.search(r"datePositions := \[\]int\{(.*?)\n\t\}", body, re.S).group(1)
pos = [int(x) for x in re.findall(r"^\s*(\d+),", pos, re.M)]
dates = re.findall(r'"(\d{4}-\d{2}-\d{2})"', re.search(r"dates := \[\]string\{(.*?)\}", body, re.S).group(1))
k["streaming_boundary"] = {"total_size": 100 * 1024, "fill": "x", "positions": pos, "dates": dates, "buffer_size": 64 * 1024,
                           "pattern": r"(\d{4}-\d{2}-\d{2})"}

# curated cases: pattern + inputs
src = open(REF + "/scripts/curated/cases.go").read()
cases = []
for m in re.finditer(r"Name:\s*%s,.*?Pattern:\s*%s,.*?Inputs?:\s*\[\]string\{(.*?)\n\t\t\}," % (STR, STR), src, re.S):
    inputs = [go_unquote(x) for x in re.findall(STR, m.group(3))]
    cases.append({"name": go_unquote(m.group(1)), "pattern": go_unquote(m.group(2)), "inputs": inputs})
k["curated_cases"] = cases

json.dump(k, open(OUT, "w"), indent=1, ensure_ascii=False)
print({a: (len(b) if hasattr(b, "__len__") else b) for a, b in k.items()})
