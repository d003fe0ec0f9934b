"""CPU study for the single-decode entropy stage (round 5): tests/emul's emul_single_decode against the oracle, with the
population the prefix pass and the merge rounds are left with for different warm-up lengths.
    python tools/single_decode_stats.py [warm bits ...]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import oracle_binding as orc
mjx = ge.load_package()
lib = ctypes.CDLL(os.path.join(ROOT, "tests", "emul", "libhuff_emul.so"))
lib.emul_single_decode_cp.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint,
                                      ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]

def run(data, warm, head=32, sub_bits=0, mode=0, layout=0, cp_bits=0):
    cap = 400000
    out = np.zeros((cap, 64), np.int16)
    nb, st = ctypes.c_size_t(), (ctypes.c_int * 8)()
    rc = lib.emul_single_decode_cp(data, len(data), layout, mode, sub_bits, warm, head, cp_bits, out.ctypes.data, cap, ctypes.byref(nb), st)
    return rc, out[: nb.value].copy(), list(st)

if __name__ == "__main__":
    warms = [int(a) for a in sys.argv[1:]] or [0, 512, 1024, 2048]
    cases = [("4K q75", mjx.synth_jpeg(3840, 2160, "420", 75, 3)), ("4K q90", mjx.synth_jpeg(3840, 2160, "420", 90, 5)),
             ("1080p q75", mjx.synth_jpeg(1920, 1080, "420", 75, 7)), ("4K q50", mjx.synth_jpeg(3840, 2160, "420", 50, 9))]
    for name, d in cases:
        ref = orc.interleave(orc.decode(d, layout=orc.LAYOUT_STD))
        for w, cp in [(w, cp) for w in warms for cp in (256, 1024)]:
            rc, coefs, st = run(d, w, cp_bits=cp)
            ok = rc == 0 and np.array_equal(coefs, ref)
            print("%-10s cp %4d warm %4d: rc %d equal %s  nsub %d rounds %d  prefix lanes %d (%.1f %%) prefix symbols %d  merge symbols %d  worst e %d b %d bad %d"
                  % (name, cp, w, rc, ok, st[0], st[1], st[2], 100.0 * st[2] / st[0], st[3], st[4], st[5], st[7], st[6]))
