"""Large, noisy, high-quality pictures (multi-workgroup scans that synchronise slowly) against the oracle: python tools/big_noisy_check.py"""
import io, os, sys
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge, oracle_binding as orc
mjx = ge.load_package(); orc.lib()
rng = np.random.default_rng(7)
ctx = mjx.Context(0)
def enc(w,h,q,ss,noise):
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100*np.sin(xx/37.0 + yy/91.0 + p) for p in (0,2,4)], -1)
    img = np.clip(base + rng.normal(0, noise, (h,w,3)), 0, 255).astype(np.uint8)
    b = io.BytesIO(); Image.fromarray(img).save(b, "JPEG", quality=q, subsampling=ss); return b.getvalue()
cases = [(2500,1800,97,2,25), (3840,2160,90,2,40), (1900,3000,100,0,10), (4096,2304,99,1,60), (3000,200,100,2,80)]
datas = [enc(*c) for c in cases]
print([len(d) for d in datas])
b = mjx.Batch(ctx, [mjx.ParsedScan(d) for d in datas], keep_coefs=True, chunk_images=2)
b.decode(); b.wait()
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(len(datas)) as ex:                         # (ctypes releases the GIL: one core per picture)
    refs = list(ex.map(lambda d: orc.decode(d, layout=orc.LAYOUT_STD), datas))
for i, (c, ref) in enumerate(zip(cases, refs)):
    assert b.status(i) == 0, (c, b.status(i))
    assert np.array_equal(b.coefs(i), orc.interleave(ref)), ("T0", c)
    diff = np.abs(b.rgb(i).astype(np.int16) - ref.rgb.astype(np.int16))
    assert diff.max() <= 1, (c, diff.max())
    print("ok", c, float((diff>0).mean()))
