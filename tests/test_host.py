"""CPU tests of the host side: C-ABI exports, JFIF parse (mirror of jpeg/mod.rs:202-465), decode tables + the per-lane
entropy routine through the CPU emulation of the GPU algorithm (tests/emul), the synthetic generator."""
import ctypes
import io
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = {  # name -> (W, H, [(id,h,v,tq,td,ta)], destuffed scan bytes incl. trailing FFD9)  -- SURVEY.md s4 table
    "huff_simple0.jpg": (16, 8, [(1, 1, 1, 0, 0, 0), (2, 1, 1, 1, 1, 1), (3, 1, 1, 1, 1, 1)], 10),
    "lena-bw.jpeg": (512, 512, [(1, 1, 1, 0, 0, 0)], 21496),
    "lena.jpeg": (512, 512, [(1, 2, 1, 0, 0, 0), (2, 1, 1, 1, 1, 1), (3, 1, 1, 1, 1, 1)], 90696),
    "2x2-chroma.jpeg": (750, 595, [(1, 2, 2, 0, 0, 0), (2, 1, 1, 1, 1, 1), (3, 1, 1, 1, 1, 1)], 145021),
}


def _read(name):
    return open(os.path.join(ROOT, "tests", "data", name), "rb").read()


def test_library_exports_every_symbol_of_the_header(mjx):
    header = open(os.path.join(ROOT, "include", "mjx.h")).read()
    declared = set(re.findall(r"\b(mjx_[a-z_0-9]+)\s*\(", header))
    assert declared == set(mjx.SYMBOLS), declared ^ set(mjx.SYMBOLS)
    raw = ctypes.CDLL(mjx.lib_path())
    for name in declared:
        assert hasattr(raw, name), name
    assert b"gfx950" in mjx.lib().mjx_version()
    assert mjx.lib().mjx_strerror(mjx.ERR_DRI_UNSUPPORTED)


@pytest.mark.parametrize("name", sorted(FIXTURES))
def test_parse_fixture(mjx, name):
    w, h, comps, scan_len = FIXTURES[name]
    s = mjx.ParsedScan(_read(name))
    d = s.desc
    assert (d.width, d.height, d.ncomp, d.scan_len) == (w, h, len(comps), scan_len)
    got = [(c.id, c.h, c.v, c.tq, c.td, c.ta) for c in list(d.comp)[:d.ncomp]]
    assert got == comps
    assert s.scan_bytes().endswith(b"\xff\xd9")
    assert b"\xff\x00" not in s.scan_bytes()[:-2] or name == "lena.jpeg"
    for c in comps:
        assert d.qt_present & (1 << c[3]) and d.dc_present & (1 << c[4]) and d.ac_present & (1 << c[5])
    s.close()


def test_parse_known_scan_prefix(mjx):
    s = mjx.ParsedScan(_read("huff_simple0.jpg"))
    assert s.scan_bytes().hex() == "fcffe2afeff3157fffd9"
    s.close()


def _code(mjx, data, strict):
    try:
        mjx.ParsedScan(data, strict_ref=strict).close()
        return mjx.OK
    except mjx.MjxError as e:
        return e.code


def test_parse_error_codes_mirror_the_reference_panics(mjx):
    good = _read("lena.jpeg")
    assert _code(mjx, _read("huff_simple0.jpg"), True) == mjx.ERR_UNSUPPORTED_MARKER     # APP12, mod.rs:445-447
    assert _code(mjx, _read("huff_simple0.jpg"), False) == mjx.OK                         # skipped
    sos = good.index(b"\xff\xda")
    dri = good[:sos] + b"\xff\xdd\x00\x04\x00\x08" + good[sos:]
    assert _code(mjx, dri, True) == mjx.ERR_DRI_UNSUPPORTED                                # mod.rs:424-428 panics
    assert _code(mjx, dri, False) == mjx.OK                                                # accepted
    assert _code(mjx, good[:sos], False) == mjx.ERR_NO_SCAN                               # image_data() == None
    assert _code(mjx, good[:100], False) == mjx.ERR_TRUNCATED
    assert _code(mjx, b"\x00\x01\x02\x03", True) == mjx.ERR_UNSUPPORTED_MARKER            # "Unhandled byte marker"
    sof2 = good.replace(b"\xff\xc0", b"\xff\xc2", 1)
    assert _code(mjx, sof2, True) == mjx.ERR_UNSUPPORTED_MARKER and _code(mjx, sof2, False) == mjx.ERR_UNSUPPORTED_FORMAT
    sof = good.index(b"\xff\xc0")
    bad_sampling = bytearray(good); bad_sampling[sof + 11] = 0x41                         # h = 4: assert at mod.rs:275
    assert _code(mjx, bytes(bad_sampling), True) == mjx.ERR_REF_PANIC
    assert _code(mjx, b"", False) == mjx.ERR_NO_SCAN


def test_parse_never_crashes_on_mutations(mjx):
    rng = np.random.default_rng(7)
    base = _read("lena-bw.jpeg")
    header_len = base.index(b"\xff\xda") + 12
    for k in range(300):
        b = bytearray(base[: header_len + 64])
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, header_len))] = int(rng.integers(0, 256))
        cut = int(rng.integers(1, len(b)))
        for data in (bytes(b), bytes(b[:cut])):
            for strict in (True, False):
                assert 0 <= _code(mjx, data, strict) <= mjx.ERR_MISSING_TABLE
    # files with restart intervals: mutations anywhere (markers inside the scan included), parse + plan must stay in bounds
    # ... and multi-scan files (markers between the scans, with and without restart intervals)
    bases = [open(os.path.join(os.path.dirname(__file__), "golden", "pil", n + ".jpg"), "rb").read()
             for n in ("dri_420_r5", "ms_420_q85_rst", "ms_422_q95", "ms_444_q40", "ms2_420_q85_rst", "ms2_444_q40")]
    for k in range(400):
        base = bases[k % len(bases)]
        b = bytearray(base)
        for _ in range(int(rng.integers(1, 8))):
            b[int(rng.integers(0, len(b)))] = int(rng.choice([0xff, 0xd0, 0xd7, 0x00, int(rng.integers(0, 256))]))
        cut = int(rng.integers(1, len(b)))
        for data in (bytes(b), bytes(b[:cut])):
            try:
                scan = mjx.ParsedScan(data)
            except mjx.MjxError as e:
                assert 0 < e.code <= mjx.ERR_MISSING_TABLE
                continue
            assert 0 <= scan.validate() <= mjx.ERR_MISSING_TABLE
            scan.close()


# ---- entropy algorithm on the CPU emulation ---------------------------------------------------------------
@pytest.fixture(scope="module")
def emul(mjx):
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "emul", "libhuff_emul.so"))
    lib.emul_decode_coefs.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]

    lib.emul_decode_coefs_sub.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p,
                                          ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]

    def run(data, layout, mode=0, sub_bits=0):
        cap = 400000
        out = np.zeros((cap, 64), np.int16)
        nb, st = ctypes.c_size_t(), (ctypes.c_int * 8)()
        rc = lib.emul_decode_coefs_sub(data, len(data), layout, mode, sub_bits, out.ctypes.data, cap, ctypes.byref(nb), st)
        return rc, out[: nb.value].copy(), list(st)
    return run


def test_subsequence_length_follows_the_scans_size(mjx, orc, emul):
    """Planning rule of round 4 (mjx_huff.h: kLongScanBits; mjx_plan.cpp: replan_subsequences): scans of at least 1.5 workgroups'
    worth of 1024-byte subsequences (0.79 MB) are cut into subsequences of 1024 .. 1280 bytes, shorter scans into 512 .. 640.
    Both sides of the rule through the emulated entropy stage -- which also walks the quad-interleaved stream of such a picture
    the way stage B does -- against the oracle's coefficients."""
    # (round 6: a shorter scan whose long subsequences fill one workgroup of 256 / 512 lanes to seven eighths takes them too --
    # 1920x1080 at quality 75 is 238 of them in a 256-lane workgroup; at quality 85 its 361 fill none that well and it keeps the short ones)
    for (w, h, q, noise), (lo, hi) in ((((3840, 2160, 75, 6.0)), (1024, 1280)), ((1920, 1080, 75, 6.0), (1024, 1280)),
                                       ((1920, 1080, 85, 6.0), (512, 640)), ((2048, 1536, 97, 30.0), (1024, 1280))):
        data = mjx.synth_jpeg(w, h, "420", q, seed=9, noise_sigma=noise)
        rc, coefs, st = emul(data, 0)
        assert rc == 0, (w, h, q)
        scan = mjx.ParsedScan(data)
        per_lane = scan.desc.scan_len / st[0]
        scan.close()
        assert lo * 0.99 <= per_lane <= hi, (w, h, q, per_lane, st[0])
        ref = orc.decode(data, layout=orc.LAYOUT_STD)
        assert np.array_equal(coefs, orc.interleave(ref)), (w, h, q)


@pytest.mark.parametrize("name", sorted(FIXTURES))
@pytest.mark.parametrize("layout", [0, 1])
def test_emulated_parallel_decode_equals_oracle_T0(mjx, orc, emul, name, layout):
    data = _read(name)
    rc, coefs, st = emul(data, layout)
    ref = orc.decode(data, layout=orc.LAYOUT_STD if layout == 0 else orc.LAYOUT_REF)
    assert rc == 0 and np.array_equal(coefs, orc.interleave(ref))


@pytest.mark.parametrize("w,h,sub,q", [(64, 48, "444", 75), (64, 48, "422", 30), (61, 45, "420", 95), (100, 60, "gray", 75),
                                       (48, 64, "440", 75), (750, 595, "420", 50), (1920, 1080, "420", 75),
                                      (1920, 1080, "420", 90), (2560, 1440, "420", 60)])
def test_emulated_decode_synthetic(mjx, orc, emul, w, h, sub, q):
    data = mjx.synth_jpeg(w, h, sub, q, seed=w + h)
    for wg in (0, 1):                        # merge rounds on a snapshot (concurrent lanes) / in place (one lane)
        rc, coefs, st = emul(data, 0, wg)
        ref = orc.decode(data, layout=orc.LAYOUT_STD)
        assert rc == 0 and np.array_equal(coefs, orc.interleave(ref)), (wg, st)


@pytest.mark.parametrize("sub_bits", [2048, 1024, 512, 256, 3000])
def test_emulated_decode_with_short_subsequences(mjx, orc, emul, sub_bits):
    """Batches too small to fill the device are re-cut into shorter subsequences (replan_subsequences, build_batch): the
    same kernel sequence must reach the same coefficients whatever the length (3000 is rounded down to a multiple of the
    checkpoint distance)."""
    for data in (_read("lena.jpeg"), _read("2x2-chroma.jpeg"), mjx.synth_jpeg(333, 217, "444", 85, seed=5),
                 mjx.synth_jpeg(1280, 720, "420", 92, seed=11), mjx.synth_jpeg(27, 546, "444", 99, seed=2)):
        ref = orc.decode(data, layout=orc.LAYOUT_STD)
        for mode in (0, 1):
            rc, coefs, st = emul(data, 0, mode, sub_bits)
            assert rc == 0 and np.array_equal(coefs, orc.interleave(ref)), (sub_bits, mode, st)


def test_emulated_decode_with_optimised_tables(mjx, orc, emul):
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    img = PIL.fromarray(rng.integers(0, 256, (72, 88, 3), dtype=np.uint8))
    for kw in (dict(optimize=True, subsampling=2, quality=85), dict(optimize=True, subsampling=0, quality=40),
               dict(optimize=False, subsampling=1, quality=100)):
        buf = io.BytesIO()
        img.save(buf, "JPEG", **kw)
        data = buf.getvalue()
        rc, coefs, st = emul(data, 0)
        try:
            ref = orc.decode(data, layout=orc.LAYOUT_STD)
        except orc.OracleError:
            continue                         # optimised tables may contain a 1-bit code the reference cannot decode (Q8)
        assert rc == 0 and np.array_equal(coefs, orc.interleave(ref))


def test_synthetic_generator_is_deterministic_and_standard(mjx, orc):
    a = mjx.synth_jpeg(96, 64, "420", 75, seed=9)
    assert a == mjx.synth_jpeg(96, 64, "420", 75, seed=9) and a != mjx.synth_jpeg(96, 64, "420", 75, seed=10)
    markers = set(re.findall(rb"\xff([\xc0-\xfe])", a[: a.index(b"\xff\xda") + 2]))
    assert markers <= {b"\xd8", b"\xe0", b"\xdb", b"\xc0", b"\xc4", b"\xda"}          # only what mod.rs:166-179 parses
    orc.decode(a, strict_ref=True, layout=orc.LAYOUT_STD)
    PIL = pytest.importorskip("PIL.Image")
    pil = np.array(PIL.open(io.BytesIO(a)).convert("RGB")).astype(float)
    std = orc.decode(a, layout=orc.LAYOUT_STD).rgb.astype(float)
    assert 10 * np.log10(255 ** 2 / np.mean((pil - std) ** 2)) > 35


def test_no_gpu_means_loud_failure_not_fallback(mjx):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mjx.MjxError) as e:
        mjx.Context(0)
    assert e.value.code == mjx.ERR_DEVICE
    with pytest.raises(mjx.MjxError) as e:
        mjx.decode(_read("lena.jpeg"))
    assert e.value.code == mjx.ERR_DEVICE


def test_host_processor_count_follows_affinity_and_quota(mjx):
    """mjx_host_processors (what mjx_decode_batch and the pool size their parse threads on) = the affinity mask capped by the
    cgroup CPU quota; a null context has no NUMA node."""
    n = int(mjx.lib().mjx_host_processors())
    assert 1 <= n <= len(os.sched_getaffinity(0))
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            assert n <= max(1, -(-int(q) // int(period)))
    except (OSError, ValueError):
        pass
    assert mjx.lib().mjx_ctx_numa_node(None) == -1


def test_ref_compat_panic_detection_matches_the_oracle(mjx, orc):
    """geometries on which the reference's placement code indexes out of bounds must be reported as
    MJX_ERR_REF_PANIC by the host plan (REF_COMPAT layout), everything else must be accepted."""
    checked = panics = 0
    for sub in ("420", "422", "440", "444", "gray"):
        for w in (8, 16, 17, 24, 31, 32, 40, 60, 64):
            for h in (8, 9, 16, 24, 36, 44, 48, 90):
                data = mjx.synth_jpeg(w, h, sub, 60, seed=w * 100 + h)
                try:
                    orc.decode(data, layout=orc.LAYOUT_REF)
                    ref_ok = True
                except orc.OracleError as e:
                    assert e.code == orc.ERR_REF_PANIC
                    ref_ok = False
                s = mjx.ParsedScan(data)
                rc = s.validate(mjx.LAYOUT_REF_COMPAT)
                assert s.validate(mjx.LAYOUT_STANDARD) == mjx.OK
                s.close()
                assert (rc == mjx.OK) == ref_ok, (sub, w, h, rc)
                assert rc in (mjx.OK, mjx.ERR_REF_PANIC)
                checked += 1
                panics += not ref_ok
    assert checked == 360 and panics > 10


PIL_DIR = os.path.join(ROOT, "tests", "golden", "pil")
# files written by libjpeg (tests/golden/pil): name -> (needs the 1-bit-code extension of the oracle)
PIL_FIXTURES = {"opt_420_q85.jpg": True, "opt_444_q40.jpg": True, "opt_422_q95.jpg": False, "std_420_q100.jpg": False,
                "opt_gray_q70.jpg": False, "opt_420_q10.jpg": True, "std_420_big.jpg": False,
                "tiny_gray_3x7_q7.jpg": False}      # one byte of entropy data: found by tools/fuzz_parity.py


@pytest.mark.parametrize("name", sorted(PIL_FIXTURES))
def test_libjpeg_written_files_on_the_emulation(mjx, orc, emul, name):
    data = open(os.path.join(PIL_DIR, name), "rb").read()
    needs_ext = PIL_FIXTURES[name]
    if needs_ext:                       # optimised tables with a 1-bit code: the reference itself cannot decode them
        with pytest.raises(orc.OracleError):
            orc.decode(data, layout=orc.LAYOUT_STD)
    ref = orc.decode(data, layout=orc.LAYOUT_STD, ext_1bit=needs_ext)
    rc, coefs, st = emul(data, 0)
    assert rc == 0 and np.array_equal(coefs, orc.interleave(ref))


def test_scan_shorter_than_the_reference_preload(mjx, orc):
    """huffman.rs:127-128 preloads four bytes and panics on a shorter scan; the bug-compatible modes report that, the
    default mode decodes the picture (the bytes past the end read as 0xAA, as the reference reads them further on)."""
    data = open(os.path.join(PIL_DIR, "tiny_gray_3x7_q7.jpg"), "rb").read()
    scan = mjx.ParsedScan(data)
    assert scan.desc.scan_len == 3
    assert scan.validate(layout=mjx.LAYOUT_STANDARD) == mjx.OK
    assert scan.validate(layout=mjx.LAYOUT_REF_COMPAT) == mjx.ERR_TRUNCATED
    assert scan.validate(strict_ref=True) == mjx.ERR_TRUNCATED
    with pytest.raises(orc.OracleError):
        orc.decode(data, layout=orc.LAYOUT_REF)
    ref = orc.decode(data, layout=orc.LAYOUT_STD)
    assert ref.rgb.shape == (7, 3, 3) and int(ref.rgb.max()) - int(ref.rgb.min()) <= 2      # a flat grey picture


def test_progressive_and_restart_files_are_rejected(mjx):
    prog = open(os.path.join(PIL_DIR, "progressive.jpg"), "rb").read()
    rst = open(os.path.join(PIL_DIR, "restart.jpg"), "rb").read()
    assert _code(mjx, prog, False) == mjx.ERR_UNSUPPORTED_FORMAT and _code(mjx, prog, True) == mjx.ERR_UNSUPPORTED_MARKER
    assert _code(mjx, rst, True) == mjx.ERR_DRI_UNSUPPORTED           # jpeg/mod.rs:424-428
    assert _code(mjx, rst, False) == mjx.OK                           # restart intervals are decoded


def test_rust_binding_mirrors_the_header():
    """bindings/rust/src/lib.rs is not compiled here (no Rust toolchain); this keeps it in step with include/mjx.h: every
    entry point is declared with the same number of parameters, and the #[repr(C)] structs list the header's fields in
    order."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "mjx.h")).read()
    rs = open(os.path.join(root, "bindings", "rust", "src", "lib.rs")).read()
    hdr_nc = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = dict((m.group(1), m.group(2)) for m in re.finditer(r"\b(mjx_[a-z_]+)\(([^;{]*?)\);", hdr_nc))
    assert len(protos) >= 24
    for name, params in protos.items():
        m = re.search(r"pub fn " + name + r"\((.*?)\)\s*(->[^;]*)?;", rs, flags=re.S)
        assert m, name + " is not declared in lib.rs"
        n_c = 0 if params.strip() in ("", "void") else params.count(",") + 1
        n_rs = 0 if not m.group(1).strip() else m.group(1).count(",") + 1
        assert n_c == n_rs, (name, n_c, n_rs)
    def c_fields(struct):
        body = re.search(r"typedef struct " + struct + r"\s*\{(.*?)\}\s*" + struct + ";", hdr_nc, flags=re.S).group(1)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = re.sub(r"^(const\s+)?[A-Za-z_0-9]+\s*\**", "", decl)
            out += [re.sub(r"[\[\]0-9\s\*]", "", n) for n in names.split(",")]
        return out
    def rs_fields(struct):
        body = re.search(r"pub struct " + struct + r"\s*\{(.*?)\n\}", rs, flags=re.S).group(1)
        return re.findall(r"pub ([a-z_0-9]+):", body)
    for struct in ("mjx_opts", "mjx_comp", "mjx_hufftab", "mjx_scan_part", "mjx_scan_desc", "mjx_image"):
        assert c_fields(struct) == rs_fields(struct), struct


def test_oracle_multi_scan_extension_is_pinned_by_twins(orc):
    """The oracle's ext_multiscan (not reference behaviour) must decode every non-interleaved twin to the picture of the
    interleaved file it was made from (tests/golden/make_multiscan.py; Pillow confirmed each twin when it was written),
    coefficients included wherever the scans carry the block; without the extension the oracle keeps to the reference."""
    pairs = {"ms_420_big": "std_420_big", "ms_444_q40": "opt_444_q40", "ms_422_q95": "opt_422_q95",
             "ms_420_q85_rst": "opt_420_q85", "ms_420_odd": "dri_420_r5_plain",
             "ms2_420_big": "std_420_big", "ms2_420_q85_rst": "opt_420_q85", "ms2_444_q40": "opt_444_q40"}    # ms2: "0; 1 2;"
    for ms, src in pairs.items():
        a = orc.decode(open(os.path.join(PIL_DIR, ms + ".jpg"), "rb").read(), layout=orc.LAYOUT_STD, ext_1bit=True, ext_dri=True,
                       ext_multiscan=True)
        b = orc.decode(open(os.path.join(PIL_DIR, src + ".jpg"), "rb").read(), layout=orc.LAYOUT_STD, ext_1bit=True)
        assert np.array_equal(a.rgb, b.rgb) and a.mcus == b.mcus, ms
        for c in range(3):
            real = np.abs(a.coefs[c]).sum(axis=1) != 0
            assert np.array_equal(a.coefs[c][real], b.coefs[c][real]), (ms, c)
    one = orc.decode(open(os.path.join(PIL_DIR, "ms_420_big.jpg"), "rb").read(), layout=orc.LAYOUT_STD)
    assert one.ncomp == 1                                   # jpeg/mod.rs:415-417: the first scan only


def test_multi_scan_files_are_parsed_into_parts(mjx):
    """One scan per component (tests/golden/make_multiscan.py): mjx_parse lists every scan as a part with its own
    de-stuffed data, tables and restart offsets; the bug-compatible modes keep the reference's view (first scan only /
    refused)."""
    for name, rst, shape in (("ms_420_big", 0, [[0], [1], [2]]), ("ms_422_q95", 0, [[0], [1], [2]]), ("ms_420_q85_rst", 7, [[0], [1], [2]]),
                             ("ms2_420_big", 0, [[0], [1, 2]]), ("ms2_420_q85_rst", 5, [[0], [1, 2]])):
        data = open(os.path.join(PIL_DIR, name + ".jpg"), "rb").read()
        scan = mjx.ParsedScan(data)
        d = scan.desc
        assert d.n_parts == len(shape) and d.ncomp == 3 and not d.scan
        assert [[d.parts[k].comp[q] for q in range(d.parts[k].ncomp)] for k in range(d.n_parts)] == shape
        assert [d.comp[c].id for c in range(3)] == [1, 2, 3]
        total = sum(d.parts[k].scan_len for k in range(d.n_parts))
        assert 0 < total < len(data)
        for k in range(d.n_parts):
            p = d.parts[k]
            assert p.restart_interval == rst and (p.n_restart > 0) == (rst > 0)
            for q in range(p.ncomp):
                assert sum(p.dc[q].bits) > 0 and sum(p.ac[q].bits) > 0
        assert scan.validate() == mjx.OK
        assert scan.validate(layout=mjx.LAYOUT_REF_COMPAT) == mjx.ERR_UNSUPPORTED_FORMAT
        # validate looks into every scan: an emptied one is refused
        d.parts[d.n_parts - 1].scan_len = 0
        assert scan.validate() == mjx.ERR_TRUNCATED
        strict = mjx.ParsedScan(data, strict_ref=True) if rst == 0 else None     # (strict: DRI is a reference panic)
        if strict is not None:
            assert strict.desc.n_parts == 0 and strict.desc.ncomp == 1           # jpeg/mod.rs:415-417: the first scan only


def test_first_scan_of_a_multi_scan_file_is_refused_unless_strict(mjx):
    """A scan with fewer components than the frame (non-interleaved baseline: not built) is refused in the default mode; in
    strict_ref mode the first scan goes through as in the reference (jpeg/mod.rs:415-417 returns after the first scan)."""
    data = bytearray(open(os.path.join(PIL_DIR, "std_420_big.jpg"), "rb").read())
    sos = data.index(b"\xff\xda")
    assert data[sos + 4] == 3
    # rewrite the scan header to a one-component scan of the first component (the entropy data then is not a valid
    # one-component scan, which the parser does not look at)
    hdr = bytes([0xff, 0xda, 0x00, 0x08, 0x01]) + bytes(data[sos + 5:sos + 7]) + bytes([0x00, 0x3f, 0x00])
    one = bytes(data[:sos]) + hdr + bytes(data[sos + 14:])
    assert _code(mjx, one, False) == mjx.ERR_UNSUPPORTED_FORMAT
    assert _code(mjx, one, True) == mjx.OK


DRI_FIXTURES = ["dri_420_r5", "dri_444_r1", "dri_422_rows", "dri_gray_r7", "dri_420_720p_rows", "dri_420_r300"]


@pytest.mark.parametrize("name", DRI_FIXTURES)
def test_restart_markers_are_parsed_out_of_the_scan(mjx, orc, name):
    """mjx_parse removes the RSTn markers and lists where the intervals begin; the oracle's ext_dri extension is pinned by
    the libjpeg twin of every file (same picture saved without restart markers = same quantised coefficients)."""
    data = open(os.path.join(PIL_DIR, name + ".jpg"), "rb").read()
    plain = open(os.path.join(PIL_DIR, name + "_plain.jpg"), "rb").read()
    scan = mjx.ParsedScan(data)
    d = scan.desc
    n_markers = sum(data.count(bytes([0xff, 0xd0 + k])) for k in range(8))
    assert d.restart_interval > 0 and d.n_restart == n_markers
    offs = [d.restart_offsets[k] for k in range(d.n_restart)]
    assert offs == sorted(offs) and offs[-1] <= d.scan_len
    sos = data.index(b"\xff\xda")
    raw = data[sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big"):]
    want, want_offs, k = bytearray(), [], 0
    while k < len(raw):                                               # FF00 -> FF, FF Dn dropped (offset recorded)
        if raw[k] == 0xff and k + 1 < len(raw) and raw[k + 1] == 0x00:
            want.append(0xff); k += 2
        elif raw[k] == 0xff and k + 1 < len(raw) and 0xd0 <= raw[k + 1] <= 0xd7:
            want_offs.append(len(want)); k += 2
        else:
            want.append(raw[k]); k += 1
    assert bytes(scan.scan_bytes()) == bytes(want) and offs == want_offs
    assert scan.validate() == mjx.OK
    ref = orc.decode(data, layout=orc.LAYOUT_STD, ext_dri=True)
    twin = orc.decode(plain, layout=orc.LAYOUT_STD)
    assert np.array_equal(orc.interleave(ref), orc.interleave(twin)) and np.array_equal(ref.rgb, twin.rgb)
    with pytest.raises(orc.OracleError):
        orc.decode(data, layout=orc.LAYOUT_STD)                       # the reference itself panics on DRI


# ---- 16-bit quantisation tables (Pq = 1, src/jpeg/mod.rs:245-256) ------------------------------------
@pytest.mark.parametrize("sub,q", [("420", 1), ("444", 3), ("422", 8), ("gray", 2), ("420", 20)])
def test_16bit_dqt_is_parsed_like_the_reference_parses_it(mjx, orc, emul, sub, q):
    """Files whose DQT segments carry 16-bit entries with values above 255 (what libjpeg writes at low quality without
    force_baseline): the host parse must deliver the table values of the file, the plan must accept them, the oracle (which
    restates mod.rs:245-256) must decode the file, and the emulated entropy path must give the oracle's coefficients."""
    import struct
    data = mjx.synth_jpeg(72, 40, sub, q, seed=q, dqt16=True)
    at = data.index(b"\xff\xdb")
    ln = struct.unpack(">H", data[at + 2:at + 4])[0]
    seg = data[at + 4:at + 2 + ln]
    ntab = 1 if sub == "gray" else 2
    assert ln == 2 + 129 * ntab and seg[0] >> 4 == 1                       # Pq = 1
    scan = mjx.ParsedScan(data)
    for t in range(ntab):
        want = struct.unpack(">64H", seg[129 * t + 1:129 * t + 129])
        assert [scan.desc.qt[t][k] for k in range(64)] == list(want)
    assert max(scan.desc.qt[0][k] for k in range(64)) > 255 or q >= 20     # really beyond 8 bits at the low qualities
    assert scan.validate() == mjx.OK and scan.validate(layout=mjx.LAYOUT_REF_COMPAT) in (mjx.OK, mjx.ERR_REF_PANIC)
    ref = orc.decode(data, layout=orc.LAYOUT_STD, strict_ref=True)
    rc, coefs, st = emul(data, 0)
    assert rc == 0 and np.array_equal(coefs, orc.interleave(ref))
    PIL = pytest.importorskip("PIL.Image")                                  # sanity bound only (libjpeg is not an oracle)
    pil = np.array(PIL.open(io.BytesIO(data)).convert("RGB")).astype(float)
    assert np.abs(pil - ref.rgb.astype(float)).mean() < 6


# ---- corrupt but decodable streams: the clamps of src/jpeg/huffman.rs:170-189 ------------------------------
@pytest.mark.parametrize("w,h,comps", [(48, 32, "420"), (64, 64, "444"), (40, 24, "gray"), (256, 128, "420")])
def test_emulated_decode_of_semantically_corrupt_streams_equals_the_oracle(mjx, orc, emul, w, h, comps):
    """Random sequences of valid codes from a table with all 256 run/size symbols, written with no regard to the 64
    coefficients of a block: the reference does not panic on them, it clamps runs onto coefficient 63, cuts ZRL short and
    decodes `0x?0` as r zeros and a 0.  The entropy path must reproduce the oracle's coefficients bit for bit."""
    import jpegwriter as jw
    data, _ = _corrupt_stream_file(jw, w, h, comps, seed=w + h)
    ref = orc.decode(data, layout=orc.LAYOUT_STD, strict_ref=True)
    assert ref.bits_used < 8 * (len(data) - 200)                           # the oracle stayed inside the scan
    flat = orc.interleave(ref)
    assert (np.abs(flat[:, 63]) > 0).mean() > 0.2                           # runs really were clamped onto coefficient 63
    for mode in (0, 1):
        rc, coefs, st = emul(data, 0, mode)
        assert rc == 0 and np.array_equal(coefs, flat), (mode, st)


def _corrupt_stream_file(jw, w, h, comps, seed, max_size=15, qmax=40):
    rng = np.random.default_rng(seed)
    dc_tab, ac_tab = jw.small_dc_table(8), jw.full_ac_table()
    frame = {"420": [(1, 2, 2, 0, 0, 0), (2, 1, 1, 0, 0, 0), (3, 1, 1, 0, 0, 0)],
             "444": [(1, 1, 1, 0, 0, 0), (2, 1, 1, 0, 0, 0), (3, 1, 1, 0, 0, 0)], "gray": [(1, 1, 1, 0, 0, 0)]}[comps]
    hmax, vmax = max(c[1] for c in frame), max(c[2] for c in frame)
    nmcu = -(-w // (8 * hmax)) * -(-h // (8 * vmax))
    need = nmcu * sum(c[1] * c[2] for c in frame) if len(frame) > 1 else -(-w // 8) * -(-h // 8)
    ent, blocks = jw.random_symbol_stream(rng, need * 40, dc_tab, ac_tab, max_size=max_size)
    assert blocks > need + 8
    qt = [int(v) for v in rng.integers(1, qmax, 64)]
    return jw.write_jpeg(w, h, frame, {0: qt}, {(0, 0): dc_tab, (1, 0): ac_tab}, ent), need


def test_pool_without_a_gpu_fails_loudly(mjx):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mjx.MjxError) as e:
        mjx.Pool([0, 0])
    assert e.value.code == mjx.ERR_DEVICE


# ---- single decode (round 5): the first decode emits ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def emul_single(mjx):
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "emul", "libhuff_emul.so"))
    lib.emul_single_decode_cp.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint,
                                          ctypes.c_uint, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]

    def run(data, layout=0, mode=0, sub_bits=0, warm=1024, head=32, cp_bits=1024):
        cap = 400000
        out = np.zeros((cap, 64), np.int16)
        nb, st = ctypes.c_size_t(), (ctypes.c_int * 8)()
        rc = lib.emul_single_decode_cp(data, len(data), layout, mode, sub_bits, warm, head, cp_bits, out.ctypes.data, cap, ctypes.byref(nb), st)
        return rc, out[: nb.value].copy(), list(st)
    return run


@pytest.mark.parametrize("warm,cp_bits", [(0, 256), (1024, 1024), (2048, 512), (512, 2048)])
def test_emulated_single_decode_equals_the_oracle(mjx, orc, emul_single, warm, cp_bits):
    """The kernel sequence of the single-decode path (k_huff_emit with its warm-up, counting merge rounds that keep the merge depth,
    k_huff_prefix right-aligning the re-decoded prefixes, k_block_gather, stage B's walk over run words with label offsets) on the
    CPU emulation, against the oracle's coefficients: the reference's samples, synthetic pictures of every sampling, noisy content
    that synchronises slowly, and short subsequences (sub_bits) where a warm-up spans a whole subsequence."""
    cases = [(_read(n), 0) for n in sorted(FIXTURES)]
    cases += [(mjx.synth_jpeg(w, h, sub, q, seed=w + h), sb) for w, h, sub, q, sb in
              [(64, 48, "444", 75, 0), (61, 45, "420", 95, 0), (100, 60, "gray", 75, 0), (750, 595, "420", 50, 0), (1920, 1080, "420", 75, 0),
               (1280, 720, "420", 92, 2048), (333, 217, "444", 85, 1024), (2560, 1440, "420", 60, 0)]]
    cases.append((open(os.path.join(ROOT, "tests", "golden", "pil", "slow_sync_444_q99.jpg"), "rb").read(), 0))
    for data, sb in cases:
        ref = orc.interleave(orc.decode(data, layout=orc.LAYOUT_STD))
        for mode in (0, 1):
            rc, coefs, st = emul_single(data, 0, mode, sb, min(warm, sb) if sb else warm, 32, cp_bits)
            assert rc == 0 and np.array_equal(coefs, ref), (len(data), sb, mode, st)


def test_emulated_single_decode_reports_a_prefix_without_head_room(mjx, orc, emul_single):
    """With no head room in front of the first decode's entries a prefix that has more entries than the wrong one it replaces
    cannot be written: the emulation must say so (the device hands such a picture to the two-pass kernels), never write out of
    its run."""
    data = mjx.synth_jpeg(1920, 1080, "420", 75, seed=5)
    ref = orc.interleave(orc.decode(data, layout=orc.LAYOUT_STD))
    rc, coefs, st = emul_single(data, warm=0, head=0, cp_bits=256)
    assert (rc == 0 and np.array_equal(coefs, ref)) or st[6] == 14, st
    assert st[5] > 0 and st[6] == 14, st          # (some prefix of this picture does grow: the case is exercised)
    rc, coefs, st = emul_single(data, warm=0, head=32, cp_bits=256)
    assert rc == 0 and np.array_equal(coefs, ref) and st[5] <= 32 * 8


# ---- measurement tooling (round-4 review, weak #7) ----------------------------------------------------------------------------------
def test_traffic_collection_sums_per_step_and_bench_refuses_per_launch_files(tmp_path):
    """tools/collect_traffic.py must SUM a counter over the step's dispatches of a kernel class (a batch of two chunks launches every
    kernel twice; averaging per launch under 'images_per_launch = the batch' halved every kernel's bytes in round 4) and check the
    pixel kernel's writes against 3*W*H*images; bench.py's committed_traffic() must accept only such per-step files."""
    import json
    import subprocess
    import sys as _sys

    def csv(path, counter, rows):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write('"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"\n')
            for did, name, val in rows:
                f.write('%d,"%s","%s",%f\n' % (did, name, counter, val))
    w, h, n = 64, 32, 10
    rgb_kb = 3 * w * h * n / 1024.0
    idct = "void mjx::k_idct_color<1, 8, true>(mjx::DevImage const*, unsigned int const*)"
    fetch_rows = [(1, "k_huff_emit", 100.0), (2, "k_huff_emit", 50.0), (3, idct, 10.0), (4, idct, 30.0), (5, "__amd_rocclr_copyBuffer", 999.0)]
    write_rows = [(1, "k_huff_emit", 7.0), (2, "k_huff_emit", 3.0), (3, idct, rgb_kb * 0.75), (4, idct, rgb_kb * 0.25), (5, "__amd_rocclr_copyBuffer", 999.0)]
    csv(str(tmp_path / "f" / "x" / "out_counter_collection.csv"), "FETCH_SIZE", fetch_rows)
    csv(str(tmp_path / "w" / "x" / "out_counter_collection.csv"), "WRITE_SIZE", write_rows)
    out = str(tmp_path / "t.json")
    tool = os.path.join(ROOT, "tools", "collect_traffic.py")
    r = subprocess.run([_sys.executable, tool, str(tmp_path / "f"), str(tmp_path / "w"), out, str(n), str(w), str(h)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    t = json.load(open(out))
    assert t["basis"] == "per_step" and t["images_per_step"] == n
    assert t["kernels"]["huff_emit"] == {"fetch_bytes": int(2 * 150 * 1024), "write_bytes": int(10 * 1024), "hbm_bytes": int(2 * 150 * 1024 + 10 * 1024), "launches": 2}
    assert t["kernels"]["idct_color"]["launches"] == 2 and abs(t["check"]["ratio"] - 1.0) < 1e-6
    assert "upload" not in t["kernels"]                       # (runtime copy kernels are not a decode class)
    # a collection that holds two steps (every class launched twice as often, the pictures written twice) must be refused
    csv(str(tmp_path / "w2" / "x" / "out_counter_collection.csv"), "WRITE_SIZE", write_rows + [(6, idct, rgb_kb)])
    csv(str(tmp_path / "f2" / "x" / "out_counter_collection.csv"), "FETCH_SIZE", fetch_rows + [(6, idct, 1.0)])
    r = subprocess.run([_sys.executable, tool, str(tmp_path / "f2"), str(tmp_path / "w2"), str(tmp_path / "t2.json"), str(n), str(w), str(h)], capture_output=True, text=True)
    assert r.returncode != 0 and "does not cover exactly one step" in (r.stdout + r.stderr)
    # bench.py: the newest committed collection is per step; a per-launch file (round 4's format) is not accepted
    import bench
    got = bench.committed_traffic()
    assert got is not None and got[1] == 256 and "idct_color" in got[0]
    assert abs(got[0]["idct_color"]["write_bytes"] / (3 * 3840 * 2160 * 256) - 1.0) < 0.01
    old = json.load(open(os.path.join(ROOT, "profiles", "r04h_traffic.json")))
    assert old.get("basis") != "per_step"                     # (what committed_traffic() would have scaled to half the real traffic)


# ---- round 5: multi-scan pictures without the gather; what the planner decides per scan -----------------------------------------------
@pytest.fixture(scope="module")
def emul_lib(mjx):
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "emul", "libhuff_emul.so"))
    lib.emul_planar_cuts.argtypes = [ctypes.c_uint] * 6 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.emul_plan_parts.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    return lib


@pytest.mark.parametrize("pic_mcux,pic_mcuy,T,hs,vs,clip_x,clip_y", [
    (240, 135, 32, 2, 2, 0, 0), (240, 135, 32, 1, 1, 0, 0), (121, 68, 32, 2, 2, 1, 1), (121, 68, 32, 1, 1, 0, 0), (33, 9, 32, 2, 1, 1, 0),
    (63, 14, 32, 1, 2, 0, 1), (84, 28, 64, 2, 1, 1, 0), (32, 5, 32, 2, 2, 0, 0), (31, 5, 32, 2, 2, 1, 1), (100, 3, 16, 4, 1, 3, 0)])
def test_segment_cuts_of_a_scan_are_what_stage_b_looks_up(emul_lib, pic_mcux, pic_mcuy, T, hs, vs, clip_x, clip_y):
    """DevImage::seg_S (mjx_kernels.h: planar_cut): the write pass of a one-component scan records a cut at every row start of the
    component's own block grid and wherever a tile of the picture begins.  Stage B (planar_probe) finds a tile's segment from the
    tile's first MCU alone -- table slot = row * S + (tile index - index of the row's first tile), its end = the next slot or the next
    row's first -- and that must be the writer's numbering for every tile, piece and block row, also where the component's grid is
    clipped short of the MCU grid (T.81 A.2.2)."""
    scan_mcux, scan_mcuy = pic_mcux * hs - clip_x, pic_mcuy * vs - clip_y        # the component's own grid (blocks)
    cap = scan_mcuy * (pic_mcux // T + 3) + 8
    mcu, slot = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    n = emul_lib.emul_planar_cuts(scan_mcux, scan_mcuy, T, pic_mcux, hs, vs, mcu.ctypes.data, slot.ctypes.data, cap)
    assert 0 < n <= cap
    mcu, slot = mcu[:n].astype(np.int64), slot[:n].astype(np.int64)
    S = (pic_mcux + T - 1) // T + 1
    assert mcu[0] == 0 and np.all(np.diff(mcu) > 0) and len(set(slot.tolist())) == n and slot.max() < scan_mcuy * S
    where = dict(zip(slot.tolist(), mcu.tolist()))
    nxt = dict(zip(mcu.tolist(), mcu.tolist()[1:] + [scan_mcux * scan_mcuy]))
    nmcu, covered = pic_mcux * pic_mcuy, 0
    for tile in range((nmcu + T - 1) // T):
        m0 = tile * T
        nm = min(T, nmcu - m0)
        m = m0
        while m < m0 + nm:                                  # the tile's pieces: one per MCU row it touches
            r, a = divmod(m, pic_mcux)
            b = min(pic_mcux, a + (m0 + nm - m))
            for v in range(vs):
                Rs, Ca = r * vs + v, a * hs
                if Rs >= scan_mcuy or Ca >= scan_mcux:
                    continue
                ia = Rs * S + (m // T - (r * pic_mcux) // T)
                ie = (Rs + 1) * S if b * hs >= scan_mcux else ia + 1
                assert where[ia] == Rs * scan_mcux + Ca, (tile, r, a, v)
                end = where.get(ie, scan_mcux * scan_mcuy)
                assert end == nxt[where[ia]] == Rs * scan_mcux + min(b * hs, scan_mcux), (tile, r, a, v)
                covered += end - where[ia]
            m += b - a
    assert covered == scan_mcux * scan_mcuy                 # every block of the scan belongs to exactly one segment


def test_planner_shares_tables_and_fills_workgroups(mjx, emul_lib):
    """mjx_plan.cpp, round 5.  (a) Slots that hold the same Huffman table share one decode table, and the decoder's block-in-MCU
    state counts inside the period of the MCU's table sequence: Cb + Cr in a scan of their own (both on the chroma tables) have
    period 1 -- two decodes that only disagree on which chroma block they are in coincide (1024 luma + chroma 4K files: 41.7 ms of
    merge rounds before).  (b) A scan under three quarters of a workgroup of 4096-bit subsequences is cut so that it fills a
    workgroup of 128, 256 or 512 lanes -- the fewest that hold it -- with subsequences of 1024 bits at least."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_multiscan

    def parts(data):
        out = (ctypes.c_int * 64)()
        n = emul_lib.emul_plan_parts(data, len(data), out, 8)
        assert 0 < n <= 8
        return [list(out[8 * i:8 * i + 8]) for i in range(n)]
    src = mjx.synth_jpeg(640, 480, "420", 85, seed=3)
    (p,) = parts(src)
    assert p[0] == 0 and p[3] == 6 and p[4] == 6 and p[7] == 0               # Y Y Y Y Cb Cr on two table pairs: period 6
    two = parts(make_multiscan.twin(src, chroma_together=True))
    assert [q[0] for q in two] == [1, 1, 2] and [q[1] for q in two[:2]] == [0, 1]
    assert two[1][2] == 2 and two[1][3] == 2 and two[1][4] == 1              # the chroma scan: two blocks per MCU, one state
    three = parts(make_multiscan.twin(src))
    assert [q[0] for q in three] == [1, 1, 1, 2] and all(q[4] == 1 for q in three[:3])
    gray = parts(mjx.synth_jpeg(64, 64, "gray", 75, seed=1))
    assert gray[0][3] == gray[0][4] == 1
    # (b): a scan below three quarters of a workgroup is cut for the fewest lanes -- 128, 256 or 512 -- that hold it at up to 4096
    # bits each, and fills them (not below 1024 bits per subsequence); 1080p (round 6): 238 long subsequences in a 256-lane workgroup
    for (w, h, q), (lo, hi), (nlo, nhi) in (((512, 512, 75), (2048, 2560), (100, 128)), ((1024, 768, 75), (3072, 4096), (200, 256)),
                                            ((1280, 720, 75), (3584, 4096), (200, 256)), ((1920, 1080, 75), (8192, 10240), (224, 256)),
                                            ((256, 256, 75), (1024, 1024), (40, 128)), ((96, 64, 75), (1024, 1024), (1, 16))):
        (p,) = parts(mjx.synth_jpeg(w, h, "420", q, seed=5))
        assert lo <= p[5] <= hi and p[5] % 256 == 0 and nlo <= p[6] <= nhi, (w, h, p)


def test_emulated_rounds_keep_two_decodes_per_subsequence(mjx, orc, emul, monkeypatch):
    """Gen2 (mjx_kernels.hip; the emulation mirrors it): where content does not synchronise -- a grey half beside a white half is a
    periodic bit string per MCU row -- the subsequences inside used to be put right one per round and decoded in full again; with the
    decode before the last kept, the truth meets the first decode at its first checkpoint and remembered exits cross a workgroup per
    round.  Same coefficients either way; the rounds drop from hundreds to a dozen, a photograph's do not rise."""
    Image = pytest.importorskip("PIL.Image")
    a = np.full((1080, 1920, 3), 255, np.uint8)
    a[:, :960] = 128
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, "JPEG", quality=75, subsampling=2)
    flat = buf.getvalue()
    photo = _read("2x2-chroma.jpeg")
    for data, sub_bits in ((flat, 512), (flat, 4096), (photo, 512), (_read("lena.jpeg"), 1024)):
        ref = orc.interleave(orc.decode(data, layout=orc.LAYOUT_STD))
        rounds = {}
        for memo in ("1", "0"):
            monkeypatch.setenv("MJX_MERGE_MEMO", memo)
            rc, coefs, st = emul(data, 0, 0, sub_bits)
            assert rc == 0 and np.array_equal(coefs, ref), (memo, sub_bits)
            rounds[memo] = st[1]
        assert rounds["1"] <= rounds["0"], rounds
        if data is flat:
            assert rounds["1"] * 5 < rounds["0"] and rounds["1"] <= 40, rounds
