#!/usr/bin/env python3
"""Dev-time harvester: copies the known-answer DATA (numbers only) that the reference's own unit
tests hold for the hot path into tests/golden/reference_kats.json.

Run in the build container only (needs /root/reference); the GPU box uses the committed JSON.
Sources: src/fdct.rs:249-274 (INPUT1/OUTPUT1/INPUT2/OUTPUT2), src/image_buffer.rs:326-421
(assert_rgb_to_ycbcr triples), src/encoder.rs:1302-1321 (sampling-factor table).
"""
import json
import os
import re

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def ints(s):
    return [int(v) for v in re.findall(r"-?\d+", s)]


def main():
    fd = open(os.path.join(REF, "fdct.rs")).read()
    vec = {}
    for name in ("INPUT1", "OUTPUT1", "INPUT2", "OUTPUT2"):
        m = re.search(r"const %s: \[i16; 64\] = \[(.*?)\];" % name, fd, re.S)
        vec[name] = ints(m.group(1))
        assert len(vec[name]) == 64
    ib = open(os.path.join(REF, "image_buffer.rs")).read()
    triples = [[ints(a), ints(b)] for a, b in
               re.findall(r"assert_rgb_to_ycbcr\(\[([^\]]*)\], \[([^\]]*)\]\)", ib)]
    assert len(triples) == 93
    en = open(os.path.join(REF, "encoder.rs")).read()
    sf = [[n, int(h), int(v)] for n, h, v in
          re.findall(r"SamplingFactor::(\w+)\.get_sampling_factors\(\), \((\d), (\d)\)", en)]
    assert len(sf) == 16
    enum_vals = {}
    for n, expr in re.findall(r"^\s+([FR]_\w+) = ([^,]+),", en, re.M):
        enum_vals[n] = eval(expr)
    out = {
        "_provenance": "harvested by tests/golden/harvest_reference_kats.py from the reference's unit tests",
        "fdct": [{"input": vec["INPUT1"], "output": vec["OUTPUT1"]},
                 {"input": vec["INPUT2"], "output": vec["OUTPUT2"]}],
        "rgb_to_ycbcr": triples,
        "sampling_factors": sf,
        "sampling_factor_enum": enum_vals,
    }
    with open(os.path.join(HERE, "reference_kats.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("fdct vectors: 2, colour triples:", len(triples), "sampling rows:", len(sf))


if __name__ == "__main__":
    main()
