#!/usr/bin/env python3
"""Fixture generator (authoring container only): the reference's e2e corpus -- patterns, inputs and engine/feature
labels of /root/reference/tests/e2e/testdata.json (data, not code) -> tests/golden/e2e_corpus.json."""
import json, os
d = json.load(open("/root/reference/tests/e2e/testdata.json"))
out = [{"pattern": e["pattern"], "inputs": e["inputs"], "engine_labels": e["engine_labels"], "feature_labels": e["feature_labels"]} for e in d]
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "e2e_corpus.json"), "w"), ensure_ascii=False, indent=0)
print(len(out), "patterns")
