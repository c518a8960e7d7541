"""Access to the committed golden fixtures (tests/golden/, produced by make_golden.py from the
unmodified reference binary)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def load_case(name):
    m = manifest()[name]
    post = np.fromfile(os.path.join(GOLDEN, name + ".post"), dtype="<f4").reshape(-1, 40)
    with open(os.path.join(GOLDEN, name + ".list")) as f:
        lines = [ln.rstrip("\n") for ln in f]
    return m, post, lines


def encode_cases():
    with open(os.path.join(GOLDEN, "encode_cases.json")) as f:
        return json.load(f)


def as_strings(msgs):
    return ["".join("1" if b else "0" for b in row) for row in msgs]


def sync_kw(m):
    return dict(sync_marker=m.get("sync_marker", ""), sync_period=m.get("sync_period", 0))
