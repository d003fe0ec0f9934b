"""Diagnostic: where a stage-B wave spends a tile (build with tools/build_variant.sh stampb -DMJX_STAMP_B, run with MJX_LIB=ab/libmjx_stampb.so).
MJX_STREAM_LINEAR=1 gives the same for the linear stream."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
mjx = ge.load_package()
ctx = mjx.Context(0)
q = int(sys.argv[1]) if len(sys.argv) > 1 else 75
sub = sys.argv[2] if len(sys.argv) > 2 else "420"          # (other layouts: the generic form; cycles per wave and tile are then per ITS tiles)
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160)
copies = int(sys.argv[5]) if len(sys.argv) > 5 else 16
datas = [mjx.synth_jpeg(W, H, sub, q, seed=s) for s in range(16)]
base = mjx.Batch(ctx, [mjx.ParsedScan(d) for d in datas])
b = base.tile(copies)
out = (ctypes.c_ulonglong * 8)()
f = mjx.lib().mjx_debug_stamps
f.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
b.decode(); b.wait()
f(out, 1)
b.decode(); b.wait()
f(out, 0)
v = list(out)
names = ["zero fill (+ its barrier)", "scatter (+ mask, further rounds)", "DC term + next tile's fetch issued", "barriers around the IDCT", "IDCT", "settle: wait for the next tile's words", "pixels", "last barrier"]
tot = sum(v[:8])
tile_mcus = {"420": int(os.environ.get("MJX_TILE420", "16")), "422": 32, "444": 64, "gray": 128, "440": 32}.get(sub, 32)
mw, mh = {"420": (16, 16), "422": (16, 8), "440": (8, 16), "444": (8, 8), "gray": (8, 8)}.get(sub, (16, 16))
mcus = ((W + mw - 1) // mw) * ((H + mh - 1) // mh)
tiles = (mcus + tile_mcus - 1) // tile_mcus * copies // 16
for n, x in zip(names, v):
    print("%-44s %6.1f %%  %8.0f cycles per wave and tile" % (n, 100.0 * x / tot, x / (256 * tiles * 4.0)))
print("total shader cycles per wave and tile", tot / (256 * tiles * 4.0), "tiles per picture", tiles)
