"""ctypes binding of the CPU oracle (oracle/libmjx_oracle.so).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYOUT_REF, LAYOUT_STD = 0, 1
OK, ERR_REF_PANIC, ERR_UNSUPPORTED, ERR_NO_SCAN, ERR_NOMEM = range(5)


class Opts(ctypes.Structure):
    _fields_ = [("strict_ref", ctypes.c_int), ("layout", ctypes.c_int), ("faithful_cos", ctypes.c_int),
                ("faithful_huff", ctypes.c_int), ("ext_1bit", ctypes.c_int), ("ext_dri", ctypes.c_int),
                ("ext_multiscan", ctypes.c_int)]


class Img(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int), ("height", ctypes.c_int), ("ncomp", ctypes.c_int), ("hs", ctypes.c_int * 3),
                ("vs", ctypes.c_int * 3), ("rgb", ctypes.POINTER(ctypes.c_uint8)),
                ("coef", ctypes.POINTER(ctypes.c_int16) * 3), ("nblocks", ctypes.c_size_t * 3),
                ("mcus_read", ctypes.c_size_t), ("bits_used", ctypes.c_size_t), ("msg", ctypes.c_char * 160)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "libmjx_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        l = ctypes.CDLL(path)
        l.orc_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(Opts), ctypes.POINTER(Img)]
        l.orc_decode.restype = ctypes.c_int
        l.orc_free_image.argtypes = [ctypes.POINTER(Img)]
        l.orc_idct_ref.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        l.orc_ycbcr_to_rgb.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
        l.orc_decode_many.restype = ctypes.c_uint64
        l.orc_decode_many.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_size_t,
                                      ctypes.POINTER(Opts), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        l.orc_decode_many_rgb.restype = ctypes.c_uint64
        l.orc_decode_many_rgb.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_size_t,
                                          ctypes.POINTER(Opts), ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                          ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        _lib = l
    return _lib


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        self.code = code
        super().__init__("oracle rc=%d: %s" % (code, msg))


class Decoded:
    pass


def decode(data, layout=LAYOUT_REF, strict_ref=False, faithful_cos=False, faithful_huff=False, ext_1bit=False, ext_dri=False,
           ext_multiscan=False):
    """-> object with rgb [H,W,3] u8, coefs (list per component of int16 [blocks,64]), hv (blocks per MCU per comp)."""
    o = Opts(int(strict_ref), int(layout), int(faithful_cos), int(faithful_huff), int(ext_1bit), int(ext_dri), int(ext_multiscan))
    im = Img()
    rc = lib().orc_decode(bytes(data), len(data), ctypes.byref(o), ctypes.byref(im))
    if rc != OK:
        raise OracleError(rc, im.msg.decode(errors="replace"))
    r = Decoded()
    r.width, r.height, r.ncomp = im.width, im.height, im.ncomp
    r.rgb = np.ctypeslib.as_array(im.rgb, (im.height, im.width, 3)).copy()
    r.coefs = [np.ctypeslib.as_array(im.coef[c], (im.nblocks[c], 64)).copy() for c in range(im.ncomp)]
    r.hv = [im.hs[c] * im.vs[c] for c in range(im.ncomp)]
    r.mcus, r.bits_used = im.mcus_read, im.bits_used
    lib().orc_free_image(ctypes.byref(im))
    return r


def interleave(dec):
    """Per-component T0 streams -> MCU-interleaved decode order [mcus * bpm, 64] (the device buffer's order)."""
    bpm = sum(dec.hv)
    out = np.empty((dec.mcus, bpm, 64), np.int16)
    off = 0
    for c, k in enumerate(dec.hv):
        out[:, off:off + k, :] = dec.coefs[c].reshape(dec.mcus, k, 64)
        off += k
    return out.reshape(-1, 64)


def idct_ref(block_natural, faithful_cos=False):
    a = np.ascontiguousarray(block_natural, np.float32).reshape(64)
    out = np.empty(64, np.float32)
    lib().orc_idct_ref(a.ctypes.data, out.ctypes.data, int(faithful_cos))
    return out.reshape(8, 8)


def ycbcr_to_rgb(y, cb, cr):
    out = np.zeros(3, np.uint8)
    lib().orc_ycbcr_to_rgb(float(y), float(cb), float(cr), out.ctypes.data)
    return out


def decode_many(datas, nthreads, layout=LAYOUT_REF, faithful=True, rgb_shapes=None):
    """Decodes the files on `nthreads` threads -> (pixels decoded, [status]) or, with rgb_shapes = [(height, width)] per
    file, (pixels, [status], [rgb array [H,W,3] u8])."""
    n = len(datas)
    arr = (ctypes.c_char_p * n)(*datas)
    lens = (ctypes.c_size_t * n)(*[len(d) for d in datas])
    st = (ctypes.c_int * n)()
    o = Opts(0, layout, int(faithful), int(faithful), 0, 0, 0)
    if rgb_shapes is None:
        px = lib().orc_decode_many(arr, lens, n, ctypes.byref(o), nthreads, st)
        return int(px), list(st)
    outs = [np.zeros((h, w, 3), np.uint8) for h, w in rgb_shapes]
    ptrs = (ctypes.c_void_p * n)(*[a.ctypes.data for a in outs])
    caps = (ctypes.c_size_t * n)(*[a.nbytes for a in outs])
    px = lib().orc_decode_many_rgb(arr, lens, n, ctypes.byref(o), nthreads, st, ptrs, caps)
    return int(px), list(st), outs


def f32_trunc(x):
    """decoder.rs:382-390 f32_to_u8 on one value"""
    l = lib()
    l.orc_f32_to_u8.restype = ctypes.c_uint8
    l.orc_f32_to_u8.argtypes = [ctypes.c_float]
    return int(l.orc_f32_to_u8(float(x)))
