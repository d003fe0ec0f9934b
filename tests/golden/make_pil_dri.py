"""Writes the restart-interval fixtures of tests/golden/pil with Pillow / libjpeg: every picture is
saved twice, with and without DRI/RSTn markers.  libjpeg quantises both the same way, so the two files of a pair must
decode to identical coefficients -- that is what pins the oracle's ext_dri extension (tests/test_oracle_golden.py).
Deterministic: fixed seed, fixed sizes.  Usage: python tests/golden/make_pil_dri.py"""
import io, os
import numpy as np
from PIL import Image

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pil")
CASES = [  # name, width, height, mode, subsampling, quality, restart kwargs
    ("dri_420_r5", 288, 192, "RGB", 2, 80, dict(restart_marker_blocks=5)),
    ("dri_444_r1", 100, 75, "RGB", 0, 60, dict(restart_marker_blocks=1)),
    ("dri_422_rows", 333, 222, "RGB", 1, 90, dict(restart_marker_rows=1)),
    ("dri_gray_r7", 200, 120, "L", 0, 70, dict(restart_marker_blocks=7)),
    ("dri_420_720p_rows", 1280, 720, "RGB", 2, 85, dict(restart_marker_rows=1)),
    ("dri_420_r300", 640, 480, "RGB", 2, 75, dict(restart_marker_blocks=300)),
]

def picture(w, h, mode, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    chans = []
    for c in range(3 if mode == "RGB" else 1):
        f = rng.uniform(0.01, 0.2, 4)
        p = 128 + 60 * np.sin(xx * f[0] + yy * f[1] + c) + 40 * np.cos(xx * f[2] - yy * f[3]) + rng.normal(0, 12, (h, w))
        chans.append(np.clip(p, 0, 255).astype(np.uint8))
    return Image.fromarray(np.dstack(chans) if mode == "RGB" else chans[0], mode)

for k, (name, w, h, mode, sub, q, kw) in enumerate(CASES):
    im = picture(w, h, mode, 100 + k)
    extra = {} if mode == "L" else {"subsampling": sub}
    for suffix, more in (("", kw), ("_plain", {})):
        b = io.BytesIO()
        im.save(b, "JPEG", quality=q, **extra, **more)
        open(os.path.join(OUT, name + suffix + ".jpg"), "wb").write(b.getvalue())
        print(name + suffix, len(b.getvalue()))
