"""Randomised parity sweep on a GPU box: python tools/fuzz_parity.py [seed] [images] [dump-dir] [ref]

Encodes random pictures with Pillow (libjpeg) -- random size 1..900 x 1..700, grey / 4:4:4 / 4:2:2 / 4:2:0, quality 1..100,
standard or optimised Huffman tables, with or without restart intervals, interleaved or one scan per component,
smooth / noisy / mixed content -- decodes them in
batches of random chunking (host- or device-side de-stuffing) through the C ABI and checks every image against the CPU oracle (test infrastructure):
coefficients bit-exact (T0), RGB within 1 LSB (T2a).  With `ref` as the fourth argument the bug-for-bug REF_COMPAT layout is swept instead (no restart intervals there):
pictures on which the reference panics must be reported as such, all others must match the oracle's reference layout.
tests/test_gpu_parity.py holds the fixed cases; this is the wide net.
"""
import io, os, sys
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import __graft_entry__ as ge
import oracle_binding as orc
import make_multiscan

mjx = ge.load_package()
orc.lib()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
total = int(sys.argv[2]) if len(sys.argv) > 2 else 600
dump = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None   # directory that receives every batch before it runs
ref_layout = len(sys.argv) > 4 and sys.argv[4] == "ref"
rng = np.random.default_rng(seed)
# (FUZZ_THROUGHPUT_PLAN=1: the cut of a large batch -- 512-byte subsequences, scans below one workgroup cut shorter -- instead of the
# short cuts a small batch gets)
ctx = mjx.Context(0, throughput_plan=bool(int(os.environ.get("FUZZ_THROUGHPUT_PLAN", "0"))))


def picture(w, h, kind):
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(xx / rng.uniform(3, 90) + yy / rng.uniform(3, 90) + p) for p in (0.0, 2.0, 4.0)], -1)
    if kind == 0:
        img = base
    elif kind == 1:
        img = rng.integers(0, 256, (h, w, 3)).astype(np.float64)
    elif kind == 2:
        img = base + rng.normal(0, rng.uniform(2, 40), (h, w, 3))
    else:
        img = np.full((h, w, 3), float(rng.integers(0, 256)))
        img[: h // 2, : w // 2] = rng.integers(0, 256, 3)
    return np.clip(img, 0, 255).astype(np.uint8)


def encode():
    w = int(rng.integers(1, 900)) if rng.random() < 0.8 else int(rng.integers(1, 40))
    h = int(rng.integers(1, 700)) if rng.random() < 0.8 else int(rng.integers(1, 40))
    if rng.random() < 0.25:                # the package's own generator: also 4:4:0, Annex-K tables
        sub = ["444", "422", "420", "440", "gray"][int(rng.integers(0, 5))]
        q = int(rng.integers(1, 101))
        return mjx.synth_jpeg(w, h, sub, q, seed=int(rng.integers(0, 1 << 30))), (w, h, {"synth": sub, "quality": q})
    arr = picture(w, h, int(rng.integers(0, 4)))
    grey = rng.random() < 0.15
    im = Image.fromarray(arr[..., 0] if grey else arr, "L" if grey else "RGB")
    kw = dict(quality=int(rng.integers(1, 101)), optimize=bool(rng.random() < 0.5))
    if not grey:
        kw["subsampling"] = int(rng.integers(0, 3))
    r = 1.0 if ref_layout else rng.random()
    if r < 0.2:
        kw["restart_marker_blocks"] = int(rng.integers(1, 40))
    elif r < 0.3:
        kw["restart_marker_rows"] = int(rng.integers(1, 4))
    buf = io.BytesIO()
    try:
        im.save(buf, "JPEG", **kw)
    except OSError:                       # (Pillow gives up on some option combinations; draw again)
        return encode()
    data = buf.getvalue()
    if not grey and not ref_layout and "restart_marker_blocks" not in kw and "restart_marker_rows" not in kw and rng.random() < 0.2:
        # the same coefficients as one scan per component (tests/golden/make_multiscan.py), checked against the oracle's
        # decode of the interleaved file
        try:
            rst = int(rng.integers(1, 30)) if rng.random() < 0.4 else 0
            together = bool(rng.random() < 0.5)         # "0; 1 2;": the two chroma components (1x1 each) interleaved
            return make_multiscan.twin(data, rst, chroma_together=together), (w, h, dict(kw, multiscan=True, rst=rst, cbcr=together), data)
        except KeyError:                  # an optimised table lacks a symbol the per-scan DC prediction needs
            pass
    return data, (w, h, kw)


done = differ = panics = 0
while done < total:
    items = [encode() for _ in range(int(rng.integers(1, 48)))]
    stuffed = bool(rng.random() < 0.3)             # FF00 pairs left in: the upload compacts the scans on the GPU
    scans = [mjx.ParsedScan(d, device_destuff=stuffed) for d, _ in items]
    chunk = int(rng.integers(1, 64))
    if dump:                                       # the batch that is about to run, for a post-mortem
        os.makedirs(dump, exist_ok=True)
        for f in os.listdir(dump):
            os.remove(os.path.join(dump, f))
        for k, (d, what) in enumerate(items):
            open(os.path.join(dump, "%03d.jpg" % k), "wb").write(d)
        open(os.path.join(dump, "batch.txt"), "w").write("seed %d stuffed %d chunk_images %d\n" % (seed, int(stuffed), chunk) + "\n".join(str(w[:3]) for _, w in items))
    batch = mjx.Batch(ctx, scans, keep_coefs=True, chunk_images=chunk,
                      layout=mjx.LAYOUT_REF_COMPAT if ref_layout else mjx.LAYOUT_STANDARD)
    batch.decode(); batch.wait()
    # the same list without kept coefficients: multi-scan pictures are then read straight from their scans' streams (DevImage::planar,
    # round 5) and every picture must come out byte for byte as above
    plain = None
    if not ref_layout and any(len(what) == 4 for _, what in items):
        plain = mjx.Batch(ctx, scans, chunk_images=chunk)
        plain.decode(); plain.wait()
    for i, (d, what) in enumerate(items):
        multiscan = len(what) == 4
        if plain is not None:
            assert plain.status(i) == batch.status(i), ("status without keep_coefs", what[:3], seed)
            if batch.status(i) == mjx.OK:
                assert np.array_equal(plain.rgb(i), batch.rgb(i)), ("RGB without keep_coefs", what[:3], seed)
        try:
            ref = orc.decode(d, layout=orc.LAYOUT_REF if ref_layout else orc.LAYOUT_STD, ext_dri=True, ext_1bit=True,
                             ext_multiscan=multiscan)
        except orc.OracleError:
            assert ref_layout and batch.status(i) != mjx.OK, ("oracle refuses, device decodes", what, seed)
            panics += 1
            continue
        assert batch.status(i) == mjx.OK, (what, batch.status(i))
        assert np.array_equal(batch.coefs(i), orc.interleave(ref)), ("T0", what[:3], seed)
        if multiscan:                                  # ... and the picture of the interleaved file the twin was made from
            src = orc.decode(what[3], layout=orc.LAYOUT_STD, ext_1bit=True)
            assert np.array_equal(ref.rgb, src.rgb), ("twin", what[:3], seed)
        diff = np.abs(batch.rgb(i).astype(np.int16) - ref.rgb.astype(np.int16))
        assert diff.max() <= 1, ("T2", what, int(diff.max()), seed)
        differ += int((diff > 0).sum())
    batch.close()
    if plain is not None:
        plain.close()
    done += len(items)
print("fuzz ok: seed %d, %d images, %d samples off by one%s" % (seed, done, differ, ", %d reference panics reported" % panics if ref_layout else ""))
