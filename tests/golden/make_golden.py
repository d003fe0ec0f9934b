"""Regenerates tests/golden/oracle_golden.npz -- small fixtures that pin the oracle's outputs.

The reference (Rust) cannot be built in this image and ships no tests, so these vectors are produced by the C oracle
after it was checked against the known answers recorded in SURVEY.md s4 (coefficient-stream SHA-256s, bit counts,
huff_simple0 pixels).  They are data: JPEG bytes in, coefficient hashes / RGB out.   Run:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import oracle_binding as orc  # noqa: E402

SYNTH = [(16, 8, "444", 75), (64, 48, "444", 50), (64, 48, "422", 75), (64, 36, "420", 75), (24, 40, "420", 90),
         (40, 24, "gray", 75), (33, 17, "422", 60), (48, 64, "440", 75), (56, 40, "420", 30)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ge.build()
    mjx = ge.load_package()
    out = {}
    for name in ["huff_simple0.jpg", "lena-bw.jpeg", "lena.jpeg", "2x2-chroma.jpeg"]:
        data = open(os.path.join(ROOT, "tests", "data", name), "rb").read()
        for lay, tag in ((orc.LAYOUT_REF, "ref"), (orc.LAYOUT_STD, "std")):
            d = orc.decode(data, layout=lay)
            out["file/%s/%s/rgb_sha" % (name, tag)] = np.array(sha(d.rgb))
            out["file/%s/%s/coef_sha" % (name, tag)] = np.array(sha(np.concatenate([c.astype("<i2").ravel() for c in d.coefs])))
    for i, (w, h, sub, q) in enumerate(SYNTH):
        data = mjx.synth_jpeg(w, h, sub, q, seed=100 + i)
        key = "synth/%d" % i
        out[key + "/jpeg"] = np.frombuffer(data, np.uint8)
        out[key + "/meta"] = np.array([w, h, q])
        out[key + "/sub"] = np.array(sub)
        d = orc.decode(data, layout=orc.LAYOUT_STD)
        out[key + "/std_rgb"] = d.rgb
        out[key + "/coefs"] = orc.interleave(d)
        try:
            out[key + "/ref_rgb"] = orc.decode(data, layout=orc.LAYOUT_REF).rgb
        except orc.OracleError as e:        # the reference panics on this geometry
            out[key + "/ref_panic"] = np.array(e.code)
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print("wrote", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "oracle_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
