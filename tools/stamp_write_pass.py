"""Diagnostic: shares of a write-pass wave step (build with tools/build_variant.sh stamp -DMJX_STAMP, run with MJX_LIB=ab/libmjx_stamp.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
mjx = ge.load_package()
ctx = mjx.Context(0)
datas = [mjx.synth_jpeg(3840, 2160, "420", 75, seed=s) for s in range(16)]
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
names = ["restage check + restage", "slot address + table read", "second-level lookup region", "value + output push", "x update + block end", "refill", "entry-ring flush (when due)", "DC-ring flush (when due) + loop control"]
tot = sum(v[:8])
for n, x in zip(names, v):
    print("%-30s %6.1f %%" % (n, 100.0 * x / tot))
print("total shader cycles (all waves)", tot)
