"""Writes the four flat-content pictures of test_flat_content_is_exact_and_is_not_decoded_lane_by_lane (tests/test_gpu_parity2.py) as
fixtures, so that the test needs no Pillow on the box that runs it (round-5 review, weak #11).  Run once where Pillow is installed:
    python tests/golden/flat_r06/make_flat.py
(Pillow 12.2.0, libjpeg-turbo, standard tables wrote the committed files.)"""
import io, os
import numpy as np
from PIL import Image

here = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(3)


def jpeg(a, q=75, sub=2):
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, "JPEG", quality=q, subsampling=sub)
    return buf.getvalue()


pics = []
a = np.full((1080, 1920, 3), 255, np.uint8); a[:, :960] = 128
pics.append(jpeg(a))                                                   # two flat halves
a = np.full((1080, 1920, 3), 255, np.uint8)
for y in range(60, 1000, 60):
    a[y:y + 20, 100:1800] = rng.integers(0, 255, (20, 1700, 1))
pics.append(jpeg(a))                                                   # a white page with noisy lines
a = np.zeros((768, 1024, 3), np.uint8)
for by in range(0, 768, 128):
    for bx in range(0, 1024, 128):
        a[by:by + 128, bx:bx + 128] = rng.integers(0, 256, 3)
pics.append(jpeg(a, q=90, sub=0))                                      # flat tiles, 4:4:4
a = np.full((600, 800), 200, np.uint8); a[200:400, 300:500] = 30
pics.append(jpeg(a))                                                   # grey, a dark square on a flat ground
for i, d in enumerate(pics):
    open(os.path.join(here, "flat%d.jpg" % i), "wb").write(d)
    print("flat%d.jpg" % i, len(d), "bytes")
