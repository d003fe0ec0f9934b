"""jpeg-rust_amd -- MI355X-native baseline-JPEG decode path behind martinhath/jpeg-rust's surface.

This module is only the ctypes binding over the C ABI in ``include/mjx.h`` (``libmjx.so``: host JFIF parse in C++,
HIP kernels for gfx950) plus a thin mirror of the reference's Rust interface so tests read like the reference:

    reference (src/jpeg/mod.rs, src/jpeg/decoder.rs)          here
    ---------------------------------------------------------------------------------------------
    JPEGImage::parse(bytes) -> width()/height()/image_data()   JPEGImage.parse(bytes)
    JPEGDecoder::new(data).frame_header().scan_header()        JPEGDecoder(data).frame_header()...
        .dimensions(); huffman_*_tables(); quantization_table()
        .decode() -> (Vec<(u8,u8,u8)>, usize)                  .decode() -> (ndarray[H,W,3] u8, bytes_read)
    HuffmanTable::from_size_data_tables(sizes, data)           HuffmanTable.from_size_data_tables(sizes, data)

There is no CPU fallback: every decode goes through the HIP kernels and raises ``MjxError`` (MJX_ERR_DEVICE) when no
GPU is present.  The directory name contains a hyphen (it is the name the project contract fixes), so import it with
``__graft_entry__.load_package()`` or ``importlib``.
"""
import ctypes
import sys
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)

# ---- status codes (include/mjx.h) ---------------------------------------------------------------
OK, ERR_TRUNCATED, ERR_UNSUPPORTED_MARKER, ERR_DRI_UNSUPPORTED, ERR_BAD_HUFFMAN, ERR_REF_PANIC, ERR_DEVICE, \
    ERR_UNSUPPORTED_FORMAT, ERR_NO_SCAN, ERR_INVALID_ARG, ERR_NOMEM, ERR_MISSING_TABLE = range(12)
LAYOUT_STANDARD, LAYOUT_REF_COMPAT = 0, 1
STAGE_ENTROPY, STAGE_PIXELS, STAGE_ALL = 1, 2, 3
KERNEL_NAMES = ["gather", "huff_sync", "huff_fix", "huff_scan", "huff_write", "dc_scan", "idct_color", "upload", "huff_emit", "huff_prefix"]
SUBSAMPLING = {"444": 0, "422": 1, "420": 2, "gray": 3, "440": 4}


class MjxError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = int(code)
        msg = lib().mjx_strerror(self.code).decode() if _lib is not None else str(code)
        super().__init__("mjx error %d (%s)%s" % (self.code, msg, (": " + what) if what else ""))


class Opts(ctypes.Structure):
    _fields_ = [("strict_ref", ctypes.c_uint8), ("layout", ctypes.c_uint8), ("keep_coefs", ctypes.c_uint8),
                ("device_destuff", ctypes.c_uint8), ("chunk_images", ctypes.c_uint32)]


class Comp(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint8) for n in ("id", "h", "v", "tq", "td", "ta")]


class HuffTab(ctypes.Structure):
    _fields_ = [("bits", ctypes.c_uint8 * 16), ("vals", ctypes.c_uint8 * 256)]


class ScanPart(ctypes.Structure):
    _fields_ = [("scan", ctypes.POINTER(ctypes.c_uint8)), ("scan_len", ctypes.c_size_t), ("ncomp", ctypes.c_uint8),
                ("comp", ctypes.c_uint8 * 3), ("restart_interval", ctypes.c_uint16), ("n_restart", ctypes.c_uint32),
                ("restart_offsets", ctypes.POINTER(ctypes.c_uint32)), ("dc", HuffTab * 3), ("ac", HuffTab * 3)]


class ScanDesc(ctypes.Structure):
    _fields_ = [("scan", ctypes.POINTER(ctypes.c_uint8)), ("scan_len", ctypes.c_size_t),
                ("width", ctypes.c_uint16), ("height", ctypes.c_uint16), ("ncomp", ctypes.c_uint8),
                ("comp", Comp * 3), ("qt", (ctypes.c_uint16 * 64) * 4), ("qt_present", ctypes.c_uint8),
                ("dc", HuffTab * 4), ("ac", HuffTab * 4), ("dc_present", ctypes.c_uint8),
                ("ac_present", ctypes.c_uint8), ("scan_is_stuffed", ctypes.c_uint8), ("restart_interval", ctypes.c_uint16),
                ("n_restart", ctypes.c_uint32), ("restart_offsets", ctypes.POINTER(ctypes.c_uint32)),
                ("n_parts", ctypes.c_uint8), ("parts", ctypes.POINTER(ScanPart)), ("owner_", ctypes.c_void_p)]


class Image(ctypes.Structure):
    _fields_ = [("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("rgb", ctypes.POINTER(ctypes.c_uint8))]


# every symbol include/mjx.h declares: (restype, argtypes)
_P = ctypes.POINTER
_vp, _sz, _int = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
SYMBOLS = {
    "mjx_parse": (_int, [ctypes.c_char_p, _sz, _P(Opts), _P(ScanDesc)]),
    "mjx_free_scan": (None, [_P(ScanDesc)]),
    "mjx_validate": (_int, [_P(ScanDesc), _P(Opts)]),
    "mjx_decode": (_int, [ctypes.c_char_p, _sz, _P(Opts), _P(Image)]),
    "mjx_decode_batch": (_int, [_vp, _P(ctypes.c_char_p), _P(_sz), _sz, _P(Opts), ctypes.c_uint, _P(_P(ctypes.c_uint8)), _P(_int), _P(_vp)]),
    "mjx_free_image": (None, [_P(Image)]),
    "mjx_ctx_create": (_int, [_int, _P(_vp)]),
    "mjx_ctx_destroy": (None, [_vp]),
    "mjx_ctx_set_profiling": (_int, [_vp, _int]),
    "mjx_ctx_set_throughput_plan": (_int, [_vp, _int]),
    "mjx_ctx_numa_node": (_int, [_vp]),
    "mjx_host_processors": (ctypes.c_uint, []),
    "mjx_batch_create": (_int, [_vp, _P(ScanDesc), _sz, _P(Opts), _P(_vp), _P(_int)]),
    "mjx_batch_free": (None, [_vp]),
    "mjx_batch_tile": (_int, [_vp, _vp, _sz, _P(_vp)]),
    "mjx_batch_decode": (_int, [_vp, ctypes.c_uint]),
    "mjx_batch_wait": (_int, [_vp]),
    "mjx_batch_size": (_sz, [_vp]),
    "mjx_batch_status": (_int, [_vp, _sz]),
    "mjx_batch_image_info": (_int, [_vp, _sz] + [_P(ctypes.c_uint32)] * 4),
    "mjx_batch_rgb_device": (_int, [_vp, _sz, _P(_vp), _P(_sz)]),
    "mjx_batch_copy_rgb": (_int, [_vp, _sz, _vp]),
    "mjx_batch_copy_coefs": (_int, [_vp, _sz, _vp, _sz, _P(_sz)]),
    "mjx_batch_compare_rgb": (_int, [_vp, _P(_sz), _vp, _P(_sz), _sz, _P(ctypes.c_uint32), _P(ctypes.c_uint64)]),
    "mjx_batch_bytes": (_int, [_vp] + [_P(ctypes.c_uint64)] * 4),
    "mjx_batch_geometry": (_int, [_vp] + [_P(ctypes.c_uint64)] * 3),
    "mjx_batch_unconverged_runs": (_int, [_vp, _P(ctypes.c_uint64)]),
    "mjx_batch_kernel_ms": (_int, [_vp, _P(ctypes.c_double), _P(ctypes.c_uint64), _int]),
    "mjx_decode_scans": (_int, [_vp, _P(ScanDesc), _sz, _P(Opts), _P(_P(ctypes.c_uint8)), _P(_int), _P(_vp)]),
    "mjx_pool_create": (_int, [_P(_int), _sz, _P(_vp)]),
    "mjx_pool_destroy": (None, [_vp]),
    "mjx_pool_devices": (_sz, [_vp]),
    "mjx_pool_device": (_int, [_vp, _sz]),
    "mjx_pool_set_deal": (_int, [_vp, _int]),
    "mjx_pool_decode_batch": (_int, [_vp, _P(ctypes.c_char_p), _P(_sz), _sz, _P(Opts), ctypes.c_uint, _P(_int), _P(_P(ctypes.c_uint8)), _P(_int), _P(_vp)]),
    "mjx_pool_result_locate": (_int, [_vp, _sz, _P(_sz), _P(_vp), _P(_sz)]),
    "mjx_pool_result_host": (_int, [_vp, _sz, _P(ctypes.c_uint), _P(_int)]),
    "mjx_pool_result_slot_ms": (_int, [_vp, _sz, _P(ctypes.c_double)]),
    "mjx_pool_result_free": (None, [_vp]),
    "mjx_strerror": (ctypes.c_char_p, [_int]),
    "mjx_version": (ctypes.c_char_p, []),
}

_lib = None


def lib_path():
    """In-tree library; MJX_LIB overrides it (A/B comparisons of two builds on the same GPU box)."""
    return os.environ.get("MJX_LIB") or os.path.join(_PKG, "libmjx.so")


def lib():
    """Loads libmjx.so (built in-tree by build.py).  Fails loudly if it is missing: there is no fallback path."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError("libmjx.so is not built: run `python jpeg-rust_amd/build.py` (hipcc --offload-arch=gfx950)")
        l = ctypes.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(l, name)
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


def _check(rc, what=""):
    if rc != OK:
        raise MjxError(rc, what)


DESTUFF_AUTO, DESTUFF_DEVICE, DESTUFF_HOST = 0, 1, 2


def _opts(strict_ref=False, layout=LAYOUT_STANDARD, keep_coefs=False, chunk_images=0, device_destuff=False):
    """device_destuff: True = on the GPU, False = on the host, None = the library's choice (mjx.h: MJX_DESTUFF_*)."""
    dd = DESTUFF_AUTO if device_destuff is None else (DESTUFF_DEVICE if device_destuff else DESTUFF_HOST)
    return Opts(int(bool(strict_ref)), int(layout), int(bool(keep_coefs)), dd, int(chunk_images))


# ---- host parse ------------------------------------------------------------------------------------
class ParsedScan:
    """Owns one mjx_scan_desc filled by mjx_parse (reference: the state JPEGImage::parse hands to JPEGDecoder)."""

    def __init__(self, data, strict_ref=False, device_destuff=False):
        self.desc = ScanDesc()
        self._owned = False
        o = _opts(strict_ref=strict_ref, device_destuff=device_destuff)
        _check(lib().mjx_parse(bytes(data), len(data), ctypes.byref(o), ctypes.byref(self.desc)), "mjx_parse")
        self._owned = True

    def scan_bytes(self):
        return ctypes.string_at(self.desc.scan, self.desc.scan_len)

    def validate(self, layout=LAYOUT_STANDARD, strict_ref=False):
        """Status mjx_batch_create would give this image (host only)."""
        o = _opts(layout=layout, strict_ref=strict_ref)
        return int(lib().mjx_validate(ctypes.byref(self.desc), ctypes.byref(o)))

    def close(self):
        if self._owned:
            lib().mjx_free_scan(ctypes.byref(self.desc))
            self._owned = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- device context / batch ----------------------------------------------------------------------------
class Context:
    def __init__(self, device=0, profiling=False, throughput_plan=False):
        """throughput_plan: batches are always cut into 512-byte subsequences (for a small base that Batch.tile replicates)."""
        self.h = _vp()
        _check(lib().mjx_ctx_create(int(device), ctypes.byref(self.h)), "mjx_ctx_create(device=%d)" % device)
        if profiling:
            self.set_profiling(True)
        if throughput_plan:
            _check(lib().mjx_ctx_set_throughput_plan(self.h, 1))

    def set_profiling(self, on):
        _check(lib().mjx_ctx_set_profiling(self.h, int(bool(on))))

    def close(self):
        if self.h:
            lib().mjx_ctx_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        # not during interpreter shutdown: the HIP runtime may already be gone (its teardown aborts the process when a
        # device handle is released after it); the driver reclaims everything at exit anyway
        if sys is None or sys.is_finalizing():          # (at interpreter shutdown module globals may be gone already)
            return
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """Device-resident batch: inputs uploaded at construction, ``decode()`` only enqueues kernels."""

    def __init__(self, ctx, scans=None, strict_ref=False, layout=LAYOUT_STANDARD, keep_coefs=False, chunk_images=0,
                 _handle=None):
        self.ctx = ctx
        self.h = _vp()
        if _handle is not None:
            self.h = _handle
            return
        n = len(scans)
        arr = (ScanDesc * max(n, 1))()
        for i, s in enumerate(scans):
            ctypes.memmove(ctypes.byref(arr[i]), ctypes.byref(s.desc if isinstance(s, ParsedScan) else s),
                           ctypes.sizeof(ScanDesc))
        st = (_int * max(n, 1))()
        o = _opts(strict_ref, layout, keep_coefs, chunk_images)
        _check(lib().mjx_batch_create(ctx.h, arr, n, ctypes.byref(o), ctypes.byref(self.h), st), "mjx_batch_create")
        self.create_status = list(st)[:n]

    def tile(self, times):
        h = _vp()
        _check(lib().mjx_batch_tile(self.ctx.h, self.h, int(times), ctypes.byref(h)), "mjx_batch_tile")
        return Batch(self.ctx, _handle=h)

    def decode(self, stages=STAGE_ALL):
        _check(lib().mjx_batch_decode(self.h, int(stages)), "mjx_batch_decode")

    def wait(self):
        _check(lib().mjx_batch_wait(self.h), "mjx_batch_wait")

    def __len__(self):
        return int(lib().mjx_batch_size(self.h))

    def status(self, i):
        return int(lib().mjx_batch_status(self.h, i))

    def info(self, i):
        v = [ctypes.c_uint32() for _ in range(4)]
        lib().mjx_batch_image_info(self.h, i, *[ctypes.byref(x) for x in v])
        return dict(width=v[0].value, height=v[1].value, bpm=v[2].value, mcus=v[3].value)

    def rgb(self, i):
        inf = self.info(i)
        out = np.empty((inf["height"], inf["width"], 3), np.uint8)
        _check(lib().mjx_batch_copy_rgb(self.h, i, out.ctypes.data_as(_vp)), "mjx_batch_copy_rgb")
        return out

    def rgb_device(self, i):
        p, n = _vp(), _sz()
        _check(lib().mjx_batch_rgb_device(self.h, i, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def coefs(self, i):
        """T0 stream of image i: int16 [blocks, 64], MCU-interleaved decode order, zig-zag, DC predicted."""
        inf = self.info(i)
        nb = inf["bpm"] * inf["mcus"]
        out = np.empty((nb, 64), np.int16)
        got = _sz()
        _check(lib().mjx_batch_copy_coefs(self.h, i, out.ctypes.data_as(_vp), nb, ctypes.byref(got)), "copy_coefs")
        return out

    def compare_rgb(self, mine, other, theirs):
        """On-device comparison of picture mine[k] with picture theirs[k] of batch `other` (may be self):
        -> (max |difference| per pair as uint32 array, 0xffffffff = sizes differ / a picture failed; differing bytes per pair)."""
        n = len(mine)
        assert n == len(theirs)
        ia = (_sz * max(n, 1))(*[int(x) for x in mine])
        ib = (_sz * max(n, 1))(*[int(x) for x in theirs])
        mx = np.zeros(max(n, 1), np.uint32)
        cnt = np.zeros(max(n, 1), np.uint64)
        _check(lib().mjx_batch_compare_rgb(self.h, ia, other.h, ib, n, mx.ctypes.data_as(_P(ctypes.c_uint32)),
                                           cnt.ctypes.data_as(_P(ctypes.c_uint64))), "mjx_batch_compare_rgb")
        return mx[:n], cnt[:n]

    def bytes(self):
        v = [ctypes.c_uint64() for _ in range(4)]
        _check(lib().mjx_batch_bytes(self.h, *[ctypes.byref(x) for x in v]))
        return dict(scan=v[0].value, rgb=v[1].value, coef=v[2].value, pixels=v[3].value)

    def geometry(self):
        v = [ctypes.c_uint64() for _ in range(3)]
        _check(lib().mjx_batch_geometry(self.h, *[ctypes.byref(x) for x in v]))
        return dict(subsequences=v[0].value, blocks=v[1].value, chunks=v[2].value)

    def unconverged_runs(self):
        """Chunk runs since creation whose synchronisation had not converged in time (their pictures were skipped in that run; wait()
        repairs the last decode only): a throughput loop that enqueues several decodes before one wait checks this stays put."""
        v = ctypes.c_uint64()
        _check(lib().mjx_batch_unconverged_runs(self.h, ctypes.byref(v)))
        return v.value

    def kernel_ms(self, reset=False):
        ms = (ctypes.c_double * len(KERNEL_NAMES))()
        cnt = (ctypes.c_uint64 * len(KERNEL_NAMES))()
        _check(lib().mjx_batch_kernel_ms(self.h, ms, cnt, int(reset)))
        return {k: (ms[i], int(cnt[i])) for i, k in enumerate(KERNEL_NAMES)}

    def close(self):
        if self.h:
            lib().mjx_batch_free(self.h)
            self.h = _vp()

    def __del__(self):
        # not during interpreter shutdown: the HIP runtime may already be gone (its teardown aborts the process when a
        # device handle is released after it); the driver reclaims everything at exit anyway
        if sys is None or sys.is_finalizing():          # (at interpreter shutdown module globals may be gone already)
            return
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


# ---- mirror of the reference's interface ------------------------------------------------------------------
class HuffmanTable:
    """HuffmanTable::from_size_data_tables (src/jpeg/huffman.rs:37): keeps the DHT slices; codes are built in C++."""

    def __init__(self, size_data, data_table):
        self.size_data = bytes(size_data)
        self.data_table = bytes(data_table)
        if len(self.size_data) != 16:
            raise ValueError("size_data must hold 16 counts")

    @staticmethod
    def from_size_data_tables(size_data, data_table):
        return HuffmanTable(size_data, data_table)


class FrameComponentHeader:  # src/jpeg/mod.rs:104-113
    def __init__(self, component_id, horizontal_sampling_factor, vertical_sampling_factor, quantization_selector):
        self.component_id = component_id
        self.horizontal_sampling_factor = horizontal_sampling_factor
        self.vertical_sampling_factor = vertical_sampling_factor
        self.quantization_selector = quantization_selector


class FrameHeader:  # src/jpeg/mod.rs:90-101
    def __init__(self, sample_precision, num_lines, samples_per_line, frame_components):
        self.sample_precision = sample_precision
        self.num_lines = num_lines
        self.samples_per_line = samples_per_line
        self.image_components = len(frame_components)
        self.frame_components = list(frame_components)


class ScanComponentHeader:  # src/jpeg/mod.rs:132-139
    def __init__(self, component_id, dc_table_selector, ac_table_selector):
        self.component_id = component_id
        self.dc_table_selector = dc_table_selector
        self.ac_table_selector = ac_table_selector


class ScanHeader:  # src/jpeg/mod.rs:116-129
    def __init__(self, scan_components):
        self.num_components = len(scan_components)
        self.scan_components = list(scan_components)


class JPEGDecoder:
    """Builder with the reference's method names (src/jpeg/decoder.rs:55-162); decode() runs on the GPU."""

    def __init__(self, data):
        self.data = bytes(data)
        self._frame = None
        self._scan = None
        self._dims = (0, 0)
        self._ac, self._dc, self._qt = {}, {}, {}
        self.layout = LAYOUT_STANDARD

    def dimensions(self, dims):
        self._dims = (int(dims[0]), int(dims[1]))
        return self

    def frame_header(self, frame_header):
        self._frame = frame_header
        return self

    def scan_header(self, scan_header):
        self._scan = scan_header
        return self

    def huffman_ac_tables(self, ident, table):
        self._ac[int(ident)] = table

    def huffman_dc_tables(self, ident, table):
        self._dc[int(ident)] = table

    def quantization_table(self, ident, table):
        self._qt[int(ident)] = [int(v) for v in table]

    def _desc(self):
        d = ScanDesc()
        self._buf = (ctypes.c_uint8 * (len(self.data) + 32)).from_buffer_copy(self.data + b"\xaa" * 32)
        d.scan = ctypes.cast(self._buf, ctypes.POINTER(ctypes.c_uint8))
        d.scan_len = len(self.data)
        d.width, d.height = self._dims
        if self._frame is None or self._scan is None:
            raise MjxError(ERR_REF_PANIC, "frame_header/scan_header missing (reference: unwrap on None)")
        d.ncomp = self._scan.num_components
        if d.ncomp not in (1, 3):
            raise MjxError(ERR_UNSUPPORTED_FORMAT)
        for i, sc in enumerate(self._scan.scan_components):      # scan order, decoder.rs:141-150
            fc = [f for f in self._frame.frame_components if f.component_id == sc.component_id]
            if not fc:
                raise MjxError(ERR_REF_PANIC, "scan component not in frame")
            d.comp[i] = Comp(sc.component_id, fc[0].horizontal_sampling_factor, fc[0].vertical_sampling_factor,
                             fc[0].quantization_selector, sc.dc_table_selector, sc.ac_table_selector)
        for k, t in self._qt.items():
            for j in range(64):
                d.qt[k][j] = t[j]
            d.qt_present |= 1 << k
        for store, present, tabs in ((d.dc, "dc_present", self._dc), (d.ac, "ac_present", self._ac)):
            for k, t in tabs.items():
                for j in range(16):
                    store[k].bits[j] = t.size_data[j]
                for j, v in enumerate(t.data_table[:256]):
                    store[k].vals[j] = v
                setattr(d, present, getattr(d, present) | (1 << k))
        return d

    def decode(self, ctx=None):
        """-> (rgb ndarray [H, W, 3] uint8, bytes_read).  bytes_read is bookkeeping the reference's caller ignores
        (src/jpeg/mod.rs:415-417); it is reported as the scan length."""
        ctx = ctx or default_context()
        b = Batch(ctx, [self._desc()], layout=self.layout)
        try:
            if b.create_status[0] != OK:
                raise MjxError(b.create_status[0])
            b.decode()
            b.wait()
            if b.status(0) != OK:
                raise MjxError(b.status(0))
            return b.rgb(0), len(self.data)
        finally:
            b.close()


class JPEGImage:
    """JPEGImage::parse (src/jpeg/mod.rs:202) -> width() :467, height() :471, image_data() :475."""

    def __init__(self, width, height, rgb):
        self._w, self._h, self._rgb = width, height, rgb

    @staticmethod
    def parse(data, strict_ref=False, layout=LAYOUT_STANDARD, ctx=None):
        ctx = ctx or default_context()
        scan = ParsedScan(data, strict_ref=strict_ref)
        try:
            b = Batch(ctx, [scan], strict_ref=strict_ref, layout=layout)
            try:
                if b.create_status[0] != OK:
                    raise MjxError(b.create_status[0])
                b.decode()
                b.wait()
                if b.status(0) != OK:
                    raise MjxError(b.status(0))
                return JPEGImage(scan.desc.width, scan.desc.height, b.rgb(0))
            finally:
                b.close()
        finally:
            scan.close()

    def width(self):
        return self._w

    def height(self):
        return self._h

    def image_data(self):
        """ndarray [H, W, 3] uint8 (the reference returns Option<&Vec<(u8,u8,u8)>> of length W*H, row-major)."""
        return self._rgb


def decode_batch(ctx, datas, strict_ref=False, layout=LAYOUT_STANDARD, threads=0, device_destuff=None, keep_coefs=False):
    """mjx_decode_batch: parse (host threads) + GPU decode of a list of files -> (Batch, [status per file]).
    device_destuff: the host copies the entropy-coded bytes as they are; de-stuffing, restart markers and the scan's length
    are found on the GPU."""
    n = len(datas)
    arr = (ctypes.c_char_p * max(n, 1))(*[bytes(d) for d in datas])
    lens = (_sz * max(n, 1))(*[len(d) for d in datas])
    st = (_int * max(n, 1))()
    ptrs = (_P(ctypes.c_uint8) * max(n, 1))()
    h = _vp()
    o = _opts(strict_ref, layout, keep_coefs=keep_coefs, device_destuff=device_destuff)
    _check(lib().mjx_decode_batch(ctx.h, arr, lens, n, ctypes.byref(o), int(threads), ptrs, st, ctypes.byref(h)), "mjx_decode_batch")
    return Batch(ctx, _handle=h), list(st)[:n]


class Pool:
    """mjx_pool: one context, host thread and work queue per device slot; the files of a call are dealt to the slots by
    compressed bytes (largest first; i mod N for equal files) or round robin (set_deal).  threads_per_device = 0: the slots
    share the host's processors (max(2, P / 2N) parse threads each)."""

    def __init__(self, devices):
        self.h = _vp()
        arr = (_int * len(devices))(*[int(d) for d in devices])
        _check(lib().mjx_pool_create(arr, len(devices), ctypes.byref(self.h)), "mjx_pool_create(%s)" % list(devices))

    def __len__(self):
        return int(lib().mjx_pool_devices(self.h))

    def device(self, slot):
        return int(lib().mjx_pool_device(self.h, slot))

    def set_deal(self, round_robin):
        _check(lib().mjx_pool_set_deal(self.h, 1 if round_robin else 0), "mjx_pool_set_deal")

    def decode_batch(self, datas, strict_ref=False, layout=LAYOUT_STANDARD, threads_per_device=0, device_destuff=None):
        """-> PoolResult; .slot_of[i], .status[i], .rgb(i), .rc (the call's return code: a failed slot fails its own files only)"""
        n = len(datas)
        arr = (ctypes.c_char_p * max(n, 1))(*[bytes(d) for d in datas])
        lens = (_sz * max(n, 1))(*[len(d) for d in datas])
        st = (_int * max(n, 1))()
        slots = (_int * max(n, 1))()
        ptrs = (_P(ctypes.c_uint8) * max(n, 1))()
        h = _vp()
        o = _opts(strict_ref, layout, device_destuff=device_destuff)
        rc = lib().mjx_pool_decode_batch(self.h, arr, lens, n, ctypes.byref(o), int(threads_per_device), slots, ptrs, st, ctypes.byref(h))
        if not h:
            _check(rc, "mjx_pool_decode_batch")
        res = PoolResult(h, list(slots)[:n], list(st)[:n], [ctypes.cast(p, _vp).value for p in ptrs][:n])
        res.rc = int(rc)
        return res

    def close(self):
        if self.h:
            lib().mjx_pool_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        if sys is None or sys.is_finalizing():          # (at interpreter shutdown module globals may be gone already)
            return
        try:
            self.close()
        except Exception:
            pass


class PoolResult:
    def __init__(self, h, slot_of, status, ptrs):
        self.h, self.slot_of, self.status, self.ptrs = h, slot_of, status, ptrs

    def locate(self, i):
        slot, idx, b = _sz(), _sz(), _vp()
        _check(lib().mjx_pool_result_locate(self.h, i, ctypes.byref(slot), ctypes.byref(b), ctypes.byref(idx)))
        return slot.value, b, idx.value

    def host(self, slot):
        """-> (parse threads the slot's call ran with, NUMA node its host thread is bound to or -1)"""
        t, node = ctypes.c_uint(), _int()
        _check(lib().mjx_pool_result_host(self.h, slot, ctypes.byref(t), ctypes.byref(node)), "mjx_pool_result_host")
        return int(t.value), int(node.value)

    def slot_ms(self, slot):
        """-> wall clock (ms) of the slot's own mjx_decode_batch in this call (0.0: the slot had no file)"""
        ms = ctypes.c_double()
        _check(lib().mjx_pool_result_slot_ms(self.h, slot, ctypes.byref(ms)), "mjx_pool_result_slot_ms")
        return float(ms.value)

    def compare_rgb(self, mine, theirs):
        """On-device comparison of picture mine[k] with picture theirs[k] of this result (they may lie in different slots'
        batches on one device) -> (max |difference| per pair, differing bytes per pair), see Batch.compare_rgb."""
        n = len(mine)
        mx = np.zeros(max(n, 1), np.uint32)
        cnt = np.zeros(max(n, 1), np.uint64)
        groups = {}
        for k in range(n):
            _, ba, ia = self.locate(mine[k])
            _, bb, ib = self.locate(theirs[k])
            groups.setdefault((ba.value, bb.value), []).append((k, ia, ib))
        for (ha, hb), items in groups.items():
            m = len(items)
            ia = (_sz * m)(*[x[1] for x in items])
            ib = (_sz * m)(*[x[2] for x in items])
            gm = np.zeros(m, np.uint32)
            gc = np.zeros(m, np.uint64)
            _check(lib().mjx_batch_compare_rgb(_vp(ha), ia, _vp(hb), ib, m, gm.ctypes.data_as(_P(ctypes.c_uint32)),
                                               gc.ctypes.data_as(_P(ctypes.c_uint64))), "mjx_batch_compare_rgb")
            for j, x in enumerate(items):
                mx[x[0]], cnt[x[0]] = gm[j], gc[j]
        return mx[:n], cnt[:n]

    def rgb(self, i):
        _, b, idx = self.locate(i)
        v = [ctypes.c_uint32() for _ in range(4)]
        lib().mjx_batch_image_info(b, idx, *[ctypes.byref(x) for x in v])
        out = np.empty((v[1].value, v[0].value, 3), np.uint8)
        _check(lib().mjx_batch_copy_rgb(b, idx, out.ctypes.data_as(_vp)), "mjx_batch_copy_rgb")
        return out

    def close(self):
        if self.h:
            lib().mjx_pool_result_free(self.h)
            self.h = _vp()

    def __del__(self):
        if sys is None or sys.is_finalizing():          # (at interpreter shutdown module globals may be gone already)
            return
        try:
            self.close()
        except Exception:
            pass


def decode(data, strict_ref=False, layout=LAYOUT_STANDARD):
    """One-shot C entry point mjx_decode (parse + GPU decode + copy back) -> ndarray [H, W, 3] uint8."""
    img = Image()
    o = _opts(strict_ref, layout)
    _check(lib().mjx_decode(bytes(data), len(data), ctypes.byref(o), ctypes.byref(img)), "mjx_decode")
    try:
        return np.ctypeslib.as_array(img.rgb, (img.height, img.width, 3)).copy()
    finally:
        lib().mjx_free_image(ctypes.byref(img))


# ---- synthetic inputs ----------------------------------------------------------------
_synth = None


def synth_lib():
    global _synth
    if _synth is None:
        path = os.path.join(_PKG, "synth", "libmjx_synth.so")
        if not os.path.exists(path):
            raise ImportError("libmjx_synth.so is not built: run `python jpeg-rust_amd/build.py`")
        s = ctypes.CDLL(path)
        s.mjxs_synth_jpeg.restype = _sz
        s.mjxs_synth_jpeg.argtypes = [_int] * 4 + [ctypes.c_uint64, ctypes.c_float, ctypes.c_char_p, _sz]
        s.mjxs_synth_jpeg_ex.restype = _sz
        s.mjxs_synth_jpeg_ex.argtypes = [_int] * 5 + [ctypes.c_uint64, ctypes.c_float, ctypes.c_char_p, _sz]
        s.mjxs_encode_ex.restype = _sz
        s.mjxs_encode_ex.argtypes = [_vp, _int, _int, _int, _int, _int, ctypes.c_char_p, _sz]
        s.mjxs_encode.restype = _sz
        s.mjxs_encode.argtypes = [_vp, _int, _int, _int, _int, ctypes.c_char_p, _sz]
        s.mjxs_fill_rgb.restype = None
        s.mjxs_fill_rgb.argtypes = [_vp, _int, _int, ctypes.c_uint64, ctypes.c_float]
        _synth = s
    return _synth


def synth_jpeg(width, height, subsampling="420", quality=75, seed=0, noise_sigma=6.0, dqt16=False):
    """Deterministic baseline JPEG: plane waves + noise content, Annex-K tables, only markers the reference parses.
    dqt16: 16-bit quantisation tables (Pq = 1, src/jpeg/mod.rs:245-256), values not clamped at 255."""
    cap = width * height * 3 + 65536
    buf = ctypes.create_string_buffer(cap)
    n = synth_lib().mjxs_synth_jpeg_ex(width, height, SUBSAMPLING[subsampling], quality, 1 if dqt16 else 0, seed, noise_sigma, buf, cap)
    if n == 0:
        raise RuntimeError("synthetic encode failed")
    return buf.raw[:n]


def encode_rgb(rgb, subsampling="420", quality=75, dqt16=False):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w = rgb.shape[:2]
    cap = w * h * 3 + 65536
    buf = ctypes.create_string_buffer(cap)
    n = synth_lib().mjxs_encode_ex(rgb.ctypes.data_as(_vp), w, h, SUBSAMPLING[subsampling], quality, 1 if dqt16 else 0, buf, cap)
    if n == 0:
        raise RuntimeError("encode failed")
    return buf.raw[:n]


def synth_batch(n_unique, width, height, subsampling="420", quality=75, seed0=0, threads=None):
    threads = threads or min(32, os.cpu_count() or 1)
    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(lambda s: synth_jpeg(width, height, subsampling, quality, seed0 + s), range(n_unique)))
