"""Diagnostic: where a stage-B wave spends a tile (build with tools/build_variant.sh stampb -DMJX_STAMP_B, run with MJX_LIB=ab/libmjx_stampb.so).
MJX_STREAM_LINEAR=1 gives the same for the linear stream."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
mjx = ge.load_package()
ctx = mjx.Context(0)
q = int(sys.argv[1]) if len(sys.argv) > 1 else 75
sub = sys.argv[2] if len(sys.argv) > 2 else "420"          # (other layouts: the generic form; cycles per wave and tile are then per ITS tiles)
datas = [mjx.synth_jpeg(3840, 2160, sub, q, seed=s) for s in range(16)]
base = mjx.Batch(ctx, [mjx.ParsedScan(d) for d in datas])
b = base.tile(16)
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
tile_mcus = {"420": 32, "422": 32, "444": 64, "gray": 128, "440": 32}.get(sub, 32)
mcus = {"420": 240 * 135, "422": 240 * 270, "440": 480 * 135, "444": 480 * 270, "gray": 480 * 270}.get(sub, 240 * 135)
tiles = (mcus + tile_mcus - 1) // tile_mcus
for n, x in zip(names, v):
    print("%-44s %6.1f %%  %8.0f cycles per wave and tile" % (n, 100.0 * x / tot, x / (256 * tiles * 4.0)))
print("total shader cycles per wave and tile", tot / (256 * tiles * 4.0), "tiles per picture", tiles)
