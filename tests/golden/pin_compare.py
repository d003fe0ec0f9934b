"""Half of tests/golden/pin_with_cargo.sh: compares one P3 picture written by the reference binary (src/main.rs:35-39) with
  a. the committed answers (tests/golden/ref_emul_golden.json: rgb_sha256, width, height),
  b. the oracle run now on the same file (ORC_LAYOUT_REF, cosf per term) -- byte for byte,
  c. optionally a P3 / P6 picture from `mjx_cli --ref-compat` -- within 1 LSB (the T2b tolerance).
usage: pin_compare.py <reference.ppm> <file.jpeg> <key in ref_emul_golden.json> [<gpu.ppm>]; exit 0 = all equal."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def read_ppm(path):
    """P3 exactly as src/main.rs:35-39 writes it (and P6, which mjx_cli --p6 writes) -> uint8 array [H, W, 3]."""
    raw = open(path, "rb").read()
    if raw[:2] == b"P3":
        tok = raw.split()
        w, h, mx = int(tok[1]), int(tok[2]), int(tok[3])
        assert mx == 255, mx
        a = np.array(tok[4:4 + w * h * 3], dtype=np.int64)
        assert a.size == w * h * 3, (a.size, w, h)
        return a.astype(np.uint8).reshape(h, w, 3)
    assert raw[:2] == b"P6", raw[:2]
    hdr = raw.split(b"\n", 3)
    w, h = map(int, hdr[1].split())
    return np.frombuffer(hdr[3], np.uint8, w * h * 3).reshape(h, w, 3)


def compare(ref_ppm, jpeg, key, gpu_ppm=None, out=print):
    import oracle_binding as orc
    ok = True
    ref = read_ppm(ref_ppm)
    sha = hashlib.sha256(ref.tobytes()).hexdigest()
    gold = json.load(open(os.path.join(HERE, "ref_emul_golden.json"))).get(key)
    if gold is None:
        out("%s: no committed answer under that key" % key)
        ok = False
    else:
        same = sha == gold["rgb_sha256"] and ref.shape[1] == int(gold["width"]) and ref.shape[0] == int(gold["height"])
        out("%s: reference %dx%d sha256 %s  committed %s  -> %s" % (key, ref.shape[1], ref.shape[0], sha[:16], gold["rgb_sha256"][:16], "EQUAL" if same else "DIFFERENT"))
        ok = ok and same
    o = orc.decode(open(jpeg, "rb").read(), layout=orc.LAYOUT_REF, faithful_cos=True, faithful_huff=True)
    same = o.rgb.shape == ref.shape and bool(np.array_equal(o.rgb, ref))
    out("%s: oracle (ORC_LAYOUT_REF, cosf per term) vs reference -> %s" % (key, "EQUAL byte for byte" if same else
        "DIFFERENT (max |d| %d, %d bytes)" % (int(np.abs(o.rgb.astype(int) - ref.astype(int)).max()), int((o.rgb != ref).sum())) if o.rgb.shape == ref.shape else "DIFFERENT shape"))
    ok = ok and same
    if gpu_ppm:
        g = read_ppm(gpu_ppm)
        d = int(np.abs(g.astype(int) - ref.astype(int)).max()) if g.shape == ref.shape else 999
        out("%s: mjx_cli --ref-compat vs reference: max |d| = %d (tolerance 1)" % (key, d))
        ok = ok and d <= 1
    return ok


if __name__ == "__main__":
    sys.exit(0 if compare(*sys.argv[1:5]) else 1)
