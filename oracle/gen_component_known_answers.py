#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Extracts the known-answer vectors the reference's own component unit test holds
(/root/reference/src/aslp-nnet/nnet-component-test.cc:53-206: LengthNorm, ConvolutionalComponent identity and 3x3 with a hand-computed
in-diff, MaxPoolingComponent) as DATA: the component description strings and the text matrices of every test function, parsed out of
the string literals of that file, written to tests/golden/component_known_answers.json.  Nothing of the test's code is kept.

Usage (development container only; the reference tree does not travel): python3 oracle/gen_component_known_answers.py
"""
import json
import os
import re
import sys

REF = "/root/reference/src/aslp-nnet/nnet-component-test.cc"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "component_known_answers.json")


def literal(src, pos):
    """The C string literal(s) starting at src[pos] == '"' (adjacent literals concatenated, backslash-newline continuations dropped)."""
    out = []
    while pos < len(src) and src[pos] == '"':
        pos += 1
        while src[pos] != '"':
            if src[pos] == "\\":
                if src[pos + 1] == "\n":
                    pos += 2
                    continue
                out.append({"n": "\n", "t": "\t"}.get(src[pos + 1], src[pos + 1]))
                pos += 2
                continue
            out.append(src[pos])
            pos += 1
        pos += 1
        m = re.match(r"\s*", src[pos:])
        pos += m.end()
    return "".join(out), pos


def matrix(text):
    """Kaldi text matrix "[ a b ; c d ]" (rows also end at newlines) -> list of rows."""
    body = text.strip()
    assert body.startswith("[") and body.rstrip().endswith("]"), text
    body = body[1:body.rindex("]")]
    rows = [[float(x) for x in r.split()] for r in re.split(r"[;\n]", body)]
    return [r for r in rows if r]


def main():
    src = open(REF).read()
    tests = {}
    for m in re.finditer(r"void (UnitTest\w+)\(\)\s*\{", src):
        name = m.group(1)
        end = src.find("\n  }\n", m.end())
        body = src[m.end():end]
        rec = {"matrices": {}}
        c = re.search(r"(ReadComponentFromString|Component::Init)\(\s*\"", body)
        if c:
            text, _ = literal(body, c.end() - 1)
            rec["component"] = " ".join(text.split())
            rec["component_format"] = "nnet-file" if c.group(1) == "ReadComponentFromString" else "proto-line"
        for mm in re.finditer(r"ReadCuMatrixFromString\(\s*\"", body):
            text, pos = literal(body, mm.end() - 1)
            var = re.match(r"\s*,\s*&(\w+)", body[pos:]).group(1)
            rec["matrices"][var] = matrix(text)
        tests[name] = rec
    assert set(tests) >= {"UnitTestLengthNorm", "UnitTestConvolutionalComponentUnity", "UnitTestConvolutionalComponent3x3", "UnitTestMaxPoolingComponent"}, tests.keys()
    doc = {"source": "src/aslp-nnet/nnet-component-test.cc:53-206 (string literals only)", "tests": tests}
    with open(OUT, "w") as f:
        json.dump(doc, f, sort_keys=True, separators=(",", ":"))
    print("wrote", OUT, {k: sorted(v["matrices"]) for k, v in tests.items()})


if __name__ == "__main__":
    sys.exit(main())
