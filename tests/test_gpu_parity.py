"""GPU parity tests (run on the MI355X box: pytest -m gpu).  Every call goes through the C ABI (libmjx.so) and is
compared with the CPU oracle on the same bytes:
  T0  coefficient stream bit-exact (integer Huffman / DC-prediction path)
  T2  full-image RGB within +-1 LSB per channel (float IDCT + truncating colour conversion)
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIXTURES = ["huff_simple0.jpg", "lena-bw.jpeg", "lena.jpeg", "2x2-chroma.jpeg"]
TOL = 1   # LSB per channel, BASELINE.json north_star


def _decode_both(mjx, orc, ctx, datas, layout_std=True, **kw):
    scans = [mjx.ParsedScan(d) for d in datas]
    batch = mjx.Batch(ctx, scans, keep_coefs=True,
                      layout=mjx.LAYOUT_STANDARD if layout_std else mjx.LAYOUT_REF_COMPAT, **kw)
    batch.decode()
    batch.wait()
    out = []
    for i, d in enumerate(datas):
        assert batch.status(i) == mjx.OK, "image %d status %d" % (i, batch.status(i))
        ref = orc.decode(d, layout=orc.LAYOUT_STD if layout_std else orc.LAYOUT_REF)
        out.append((ref, batch.coefs(i), batch.rgb(i)))
    batch.close()
    return out


def _check(ref, coefs, rgb, name):
    assert coefs.shape == (ref.mcus * sum(ref.hv), 64), name
    assert np.array_equal(coefs, _interleave(ref)), "T0 differs: " + name
    assert rgb.shape == ref.rgb.shape, name
    diff = np.abs(rgb.astype(np.int16) - ref.rgb.astype(np.int16))
    assert diff.max() <= TOL, "%s: max |diff| %d at %s" % (name, diff.max(), np.argwhere(diff > TOL)[:4].tolist())
    return float((diff > 0).mean())


def _interleave(ref):
    import oracle_binding
    return oracle_binding.interleave(ref)


@pytest.mark.parametrize("name", FIXTURES)
def test_reference_fixtures_standard_layout(mjx, orc, gpu_ctx, data_dir, name):
    data = open(os.path.join(data_dir, name), "rb").read()
    (ref, coefs, rgb), = _decode_both(mjx, orc, gpu_ctx, [data])
    frac = _check(ref, coefs, rgb, name)
    assert frac < 0.01, "%s: %.4f of samples differ by 1" % (name, frac)


CASES = [
    (16, 8, "444"), (64, 48, "444"), (64, 48, "422"), (64, 36, "420"), (60, 44, "420"), (61, 45, "420"),
    (33, 17, "422"), (100, 60, "gray"), (7, 5, "gray"), (48, 64, "440"), (750, 595, "420"), (512, 512, "422"),
    (1920, 1080, "420"), (1, 1, "444"), (17, 33, "420"),
]


@pytest.mark.parametrize("w,h,sub", CASES)
@pytest.mark.parametrize("quality", [50, 90])
def test_synthetic_standard_layout(mjx, orc, gpu_ctx, w, h, sub, quality):
    data = mjx.synth_jpeg(w, h, sub, quality, seed=w * 31 + h)
    (ref, coefs, rgb), = _decode_both(mjx, orc, gpu_ctx, [data])
    _check(ref, coefs, rgb, "%dx%d %s q%d" % (w, h, sub, quality))


@pytest.mark.parametrize("w,h,sub,quality", [(1920, 1080, "420", 40), (1920, 1080, "420", 88), (1920, 1080, "444", 60),
                                             (2560, 1440, "420", 75), (2560, 1440, "422", 90), (1600, 1200, "420", 95)])
def test_scan_sizes_around_the_workgroup_boundaries(mjx, orc, gpu_ctx, w, h, sub, quality):
    """Scans of 0.1 .. 1.5 MB: the per-image subsequence length (512 .. 640 bytes, chosen so that the image fills whole
    512-lane workgroups) takes different values, with and without a remainder workgroup."""
    data = mjx.synth_jpeg(w, h, sub, quality, seed=w + quality)
    (ref, coefs, rgb), = _decode_both(mjx, orc, gpu_ctx, [data])
    _check(ref, coefs, rgb, "%dx%d %s q%d" % (w, h, sub, quality))


def test_heterogeneous_batch_and_chunking(mjx, orc, gpu_ctx, data_dir):
    datas = [open(os.path.join(data_dir, n), "rb").read() for n in FIXTURES]
    datas += [mjx.synth_jpeg(w, h, s, 75, seed=i) for i, (w, h, s) in enumerate(CASES[:10])]
    for chunk in (0, 3):
        res = _decode_both(mjx, orc, gpu_ctx, datas, chunk_images=chunk)
        for i, (ref, coefs, rgb) in enumerate(res):
            _check(ref, coefs, rgb, "batch image %d chunk=%d" % (i, chunk))


def test_4k_image_many_workgroups(mjx, orc, gpu_ctx):
    data = mjx.synth_jpeg(3840, 2160, "420", 75, seed=11)
    (ref, coefs, rgb), = _decode_both(mjx, orc, gpu_ctx, [data])
    _check(ref, coefs, rgb, "4K")


def test_bad_image_does_not_kill_batch(mjx, orc, gpu_ctx, data_dir):
    good = open(os.path.join(data_dir, "lena.jpeg"), "rb").read()
    scans = [mjx.ParsedScan(good), mjx.ParsedScan(good)]
    scans[0].desc.dc_present = 0       # scan references an undefined table (decoder.rs:158-160 unwrap)
    batch = mjx.Batch(gpu_ctx, scans)
    assert batch.create_status[0] == mjx.ERR_MISSING_TABLE and batch.create_status[1] == mjx.OK
    batch.decode()
    batch.wait()
    ref = orc.decode(good, layout=orc.LAYOUT_STD)
    assert np.abs(batch.rgb(1).astype(int) - ref.rgb.astype(int)).max() <= TOL
    batch.close()


def test_one_shot_and_mirror_api(mjx, orc, gpu_ctx, data_dir):
    data = open(os.path.join(data_dir, "lena.jpeg"), "rb").read()
    ref = orc.decode(data, layout=orc.LAYOUT_STD)
    rgb = mjx.decode(data)
    assert np.abs(rgb.astype(int) - ref.rgb.astype(int)).max() <= TOL
    img = mjx.JPEGImage.parse(data, ctx=gpu_ctx)
    assert (img.width(), img.height()) == (512, 512)
    assert np.array_equal(img.image_data(), rgb)
    with pytest.raises(mjx.MjxError) as e:     # the reference panics on APP12 (jpeg/mod.rs:445-447)
        mjx.JPEGImage.parse(open(os.path.join(data_dir, "huff_simple0.jpg"), "rb").read(), strict_ref=True, ctx=gpu_ctx)
    assert e.value.code == mjx.ERR_UNSUPPORTED_MARKER


def test_decode_batch_from_files(mjx, orc, gpu_ctx, data_dir):
    """mjx_decode_batch: the outer surface for a list of files (parse on host threads, decode on the GPU).  A file that does
    not parse keeps its parse status and does not disturb the others."""
    pil = os.path.join(os.path.dirname(__file__), "golden", "pil")
    names = ["lena.jpeg", "2x2-chroma.jpeg", "huff_simple0.jpg"]
    datas = [open(os.path.join(data_dir, n), "rb").read() for n in names]
    datas += [open(os.path.join(pil, n), "rb").read() for n in ("ms2_420_big.jpg", "dri_420_r5.jpg", "progressive.jpg")]
    datas += [b"not a jpeg", datas[0][:300], mjx.synth_jpeg(333, 217, "444", 85, seed=5)] * 3
    for threads in (0, 1, 5):
        batch, st = mjx.decode_batch(gpu_ctx, datas, threads=threads)
        assert len(batch) == len(datas)
        for i, d in enumerate(datas):
            try:
                ref = orc.decode(d, layout=orc.LAYOUT_STD, ext_dri=True, ext_multiscan=True)
            except orc.OracleError:
                assert st[i] != mjx.OK, i
                continue
            assert st[i] == mjx.OK and batch.status(i) == mjx.OK, (i, st[i])
            assert np.abs(batch.rgb(i).astype(int) - ref.rgb.astype(int)).max() <= TOL, i
        assert st[5] == mjx.ERR_UNSUPPORTED_FORMAT and st[6] != mjx.OK and st[7] != mjx.OK
        batch.close()


def test_tiled_batch_round_trip(mjx, orc, gpu_ctx):
    datas = [mjx.synth_jpeg(320, 240, "420", 75, seed=s) for s in range(3)]
    scans = [mjx.ParsedScan(d) for d in datas]
    base = mjx.Batch(gpu_ctx, scans)
    big = base.tile(5)
    assert len(big) == 15
    big.decode()
    big.wait()
    for i in range(15):
        ref = orc.decode(datas[i % 3], layout=orc.LAYOUT_STD)
        assert np.abs(big.rgb(i).astype(int) - ref.rgb.astype(int)).max() <= TOL
    # idempotence: decoding again gives the same bytes
    first = big.rgb(7).copy()
    big.decode()
    big.wait()
    assert np.array_equal(first, big.rgb(7))
    big.close()
    base.close()


# ---- REF_COMPAT layout: bug-for-bug placement of decoder.rs:239-312 ---------------------------------
@pytest.mark.parametrize("name", FIXTURES)
def test_reference_fixtures_ref_compat_layout(mjx, orc, gpu_ctx, data_dir, name):
    data = open(os.path.join(data_dir, name), "rb").read()
    (ref, coefs, rgb), = _decode_both(mjx, orc, gpu_ctx, [data], layout_std=False)
    frac = _check(ref, coefs, rgb, name + " (REF_COMPAT)")
    assert frac < 0.01


REF_CASES = [(64, 36, "420"), (750, 595, "420"), (16, 8, "444"), (64, 48, "422"), (33, 17, "422"), (100, 60, "gray"),
             (48, 64, "440"), (24, 40, "420"), (1920, 1080, "420"), (512, 512, "420")]


@pytest.mark.parametrize("w,h,sub", REF_CASES)
def test_synthetic_ref_compat_layout(mjx, orc, gpu_ctx, w, h, sub):
    data = mjx.synth_jpeg(w, h, sub, 75, seed=w + 7 * h)
    (ref, coefs, rgb), = _decode_both(mjx, orc, gpu_ctx, [data], layout_std=False)
    _check(ref, coefs, rgb, "%dx%d %s REF_COMPAT" % (w, h, sub))


def test_ref_compat_reports_reference_panics(mjx, orc, gpu_ctx):
    for w, h in [(64, 44), (64, 90), (60, 48)]:
        data = mjx.synth_jpeg(w, h, "420", 75, seed=3)
        with pytest.raises(orc.OracleError):
            orc.decode(data, layout=orc.LAYOUT_REF)
        batch = mjx.Batch(gpu_ctx, [mjx.ParsedScan(data)], layout=mjx.LAYOUT_REF_COMPAT)
        assert batch.create_status[0] == mjx.ERR_REF_PANIC
        batch.decode()
        batch.wait()
        batch.close()


def test_cli_writes_the_reference_ppm(mjx, orc, gpu_ctx, data_dir, tmp_path):
    import subprocess
    cli = os.path.join(os.path.dirname(mjx.lib_path()), "mjx_cli")
    out = tmp_path / "lena.ppm"
    subprocess.check_call([cli, os.path.join(data_dir, "lena.jpeg"), str(out)])
    tok = out.read_text().split()
    assert tok[:4] == ["P3", "512", "512", "255"]           # main.rs:35
    px = np.array(tok[4:], dtype=np.int32).reshape(512, 512, 3)
    ref = orc.decode(open(os.path.join(data_dir, "lena.jpeg"), "rb").read(), layout=orc.LAYOUT_STD)
    assert np.abs(px - ref.rgb.astype(np.int32)).max() <= TOL
    rc = subprocess.call([cli, os.path.join(data_dir, "huff_simple0.jpg"), str(out), "--strict"])
    assert rc == mjx.ERR_UNSUPPORTED_MARKER
    dri = os.path.join(os.path.dirname(__file__), "golden", "pil", "dri_422_rows.jpg")      # restart intervals, s8(f)-3
    subprocess.check_call([cli, dri, str(out)])
    tok = out.read_text().split()
    ref = orc.decode(open(dri, "rb").read(), layout=orc.LAYOUT_STD, ext_dri=True)
    assert tok[:4] == ["P3", "333", "222", "255"]
    assert np.abs(np.array(tok[4:], dtype=np.int32).reshape(222, 333, 3) - ref.rgb.astype(np.int32)).max() <= TOL
    assert subprocess.call([cli, dri, str(out), "--strict"]) == mjx.ERR_DRI_UNSUPPORTED
    # --p6: the binary twin of main.rs:35-39 -- same header line, then the pixels as raw bytes; it must hold exactly the numbers
    # the text file holds, for a colour and for a greyscale picture (main.rs writes r g b for both)
    for name in ("lena.jpeg", "lena-bw.jpeg"):
        src = os.path.join(data_dir, name)
        p3, p6 = tmp_path / "a.ppm", tmp_path / "b.ppm"
        subprocess.check_call([cli, src, str(p3)])
        subprocess.check_call([cli, src, str(p6), "--p6"])
        raw = p6.read_bytes()
        assert raw.startswith(b"P6\n512 512\n255\n")
        body = raw[len(b"P6\n512 512\n255\n"):]
        assert len(body) == 512 * 512 * 3
        text = np.array(p3.read_text().split()[4:], dtype=np.int32)
        assert np.array_equal(np.frombuffer(body, np.uint8).astype(np.int32), text)
    # --ref-compat writes the reference's own (bug-compatible) picture
    c3 = tmp_path / "c.ppm"
    two = os.path.join(data_dir, "2x2-chroma.jpeg")
    subprocess.check_call([cli, two, str(c3), "--p6", "--ref-compat"])
    raw = c3.read_bytes()
    hdr = b"P6\n750 595\n255\n"
    assert raw.startswith(hdr)
    ref = orc.decode(open(two, "rb").read(), layout=orc.LAYOUT_REF)
    assert np.abs(np.frombuffer(raw[len(hdr):], np.uint8).reshape(595, 750, 3).astype(np.int32) - ref.rgb.astype(np.int32)).max() <= TOL


# ---- hostile inputs: one bad image must not kill the batch, and nothing may fault on the device --------------------
def test_truncated_and_garbage_scans(mjx, orc, gpu_ctx, data_dir):
    import ctypes
    good = open(os.path.join(data_dir, "lena.jpeg"), "rb").read()
    big = mjx.synth_jpeg(1024, 768, "420", 75, seed=5)
    rng = np.random.default_rng(11)
    scans, expect = [], []
    for cut in (0.3, 0.7, 0.97):                       # file cut inside the entropy-coded segment
        scans.append(mjx.ParsedScan(big[: int(len(big) * cut)]))
        expect.append({mjx.ERR_TRUNCATED})
    for fill in ("random", "ff", "zero"):              # same headers, meaningless scan bytes
        s = mjx.ParsedScan(big)
        n = s.desc.scan_len
        raw = {"random": rng.integers(0, 256, n, dtype=np.uint8).tobytes(), "ff": b"\xff" * n, "zero": b"\x00" * n}[fill]
        ctypes.memmove(s.desc.scan, raw, n)
        scans.append(s)
        expect.append({mjx.OK, mjx.ERR_TRUNCATED, mjx.ERR_BAD_HUFFMAN})
    scans.append(mjx.ParsedScan(good))
    expect.append({mjx.OK})
    for layout in (mjx.LAYOUT_STANDARD, mjx.LAYOUT_REF_COMPAT):
        batch = mjx.Batch(gpu_ctx, scans, layout=layout)
        for _ in range(2):
            batch.decode()
            batch.wait()
        got = [batch.status(i) for i in range(len(scans))]
        for g, e in zip(got, expect):
            assert g in e, (got, layout)
        ref = orc.decode(good, layout=orc.LAYOUT_STD if layout == mjx.LAYOUT_STANDARD else orc.LAYOUT_REF)
        assert np.abs(batch.rgb(len(scans) - 1).astype(int) - ref.rgb.astype(int)).max() <= TOL
        batch.close()


def test_repair_path_when_the_enqueued_rounds_do_not_converge(mjx, orc, data_dir, tmp_path):
    """MJX_FIX_PASSES=1 leaves the synchronisation unverified after the enqueued rounds, so mjx_batch_wait must take
    its repair path (more rounds, then re-run the tail of the pipeline) and still produce the exact result."""
    import subprocess, sys
    script = tmp_path / "repair.py"
    script.write_text(
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "import __graft_entry__ as ge, oracle_binding as orc\n"
        "mjx = ge.load_package()\n"
        "ctx = mjx.Context(0)\n"
        "datas = [mjx.synth_jpeg(3840, 2160, '420', 75, seed=s) for s in (1, 2)] + [mjx.synth_jpeg(640, 480, '422', 60, seed=3)]\n"
        "b = mjx.Batch(ctx, [mjx.ParsedScan(d) for d in datas], keep_coefs=True, chunk_images=2)\n"
        "b.decode(); b.wait()\n"
        "for i, d in enumerate(datas):\n"
        "    ref = orc.decode(d, layout=orc.LAYOUT_STD)\n"
        "    assert b.status(i) == 0\n"
        "    assert np.array_equal(b.coefs(i), orc.interleave(ref))\n"
        "    assert np.abs(b.rgb(i).astype(int) - ref.rgb.astype(int)).max() <= 1\n"
        "print('repair ok')\n" % (os.path.dirname(data_dir.rstrip('/')).rsplit('/tests', 1)[0], os.path.dirname(data_dir.rstrip('/')).rsplit('/tests', 1)[0]))
    env = dict(os.environ, MJX_FIX_PASSES="1")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "repair ok" in out.stdout, out.stdout + out.stderr


def test_two_streams_give_the_same_bytes(mjx, orc, data_dir, tmp_path):
    """MJX_STREAMS=2 runs odd chunks on a second stream with their own scratch buffers; a five-chunk batch without
    kept coefficients must produce the same RGB as the oracle allows, and the last chunk's coefficients stay readable."""
    import subprocess, sys
    root = os.path.dirname(data_dir.rstrip('/')).rsplit('/tests', 1)[0]
    script = tmp_path / "streams.py"
    script.write_text(
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "import __graft_entry__ as ge, oracle_binding as orc\n"
        "mjx = ge.load_package()\n"
        "ctx = mjx.Context(0)\n"
        "datas = [mjx.synth_jpeg(1920, 1080, '420', 75, seed=s) for s in range(4)] + [mjx.synth_jpeg(640, 480, '422', 60, seed=9),\n"
        "         mjx.synth_jpeg(333, 217, '444', 85, seed=5), mjx.synth_jpeg(64, 48, 'gray', 50, seed=6), mjx.synth_jpeg(3840, 2160, '420', 75, seed=7),\n"
        "         mjx.synth_jpeg(100, 100, '420', 30, seed=8)]\n"
        "b = mjx.Batch(ctx, [mjx.ParsedScan(d) for d in datas], chunk_images=2)\n"
        "b.decode(); b.wait()\n"
        "refs = [orc.decode(d, layout=orc.LAYOUT_STD) for d in datas]\n"
        "for i, ref in enumerate(refs):\n"
        "    assert b.status(i) == 0\n"
        "    assert np.abs(b.rgb(i).astype(int) - ref.rgb.astype(int)).max() <= 1, i\n"
        "assert np.array_equal(b.coefs(len(datas) - 1), orc.interleave(refs[-1]))\n"
        "b.decode(); b.wait()\n"
        "assert np.abs(b.rgb(7).astype(int) - refs[7].rgb.astype(int)).max() <= 1\n"
        "print('streams ok')\n" % (root, root))
    env = dict(os.environ, MJX_STREAMS="2")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "streams ok" in out.stdout, out.stdout + out.stderr


def test_slowly_synchronising_stream_with_stale_scratch(mjx, orc, data_dir, tmp_path):
    """A 27x546 4:4:4 quality-99 picture of noise needs twelve synchronisation rounds where four are enqueued (found by
    tools/fuzz_parity.py).  Until the host has seen that at mjx_batch_wait, nothing behind the synchronisation may run on
    the unconverged state: with scratch that holds stale bytes (MJX_POISON fills it) stage B used to read tile offsets
    nobody had written and fault.  The picture shares a chunk with ordinary ones, all of which must come out right."""
    import subprocess, sys
    root = os.path.dirname(data_dir.rstrip('/')).rsplit('/tests', 1)[0]
    script = tmp_path / "slow.py"
    script.write_text(
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "import __graft_entry__ as ge, oracle_binding as orc\n"
        "mjx = ge.load_package()\n"
        "ctx = mjx.Context(0)\n"
        "slow = open(os.path.join(%r, 'tests', 'golden', 'pil', 'slow_sync_444_q99.jpg'), 'rb').read()\n"
        "datas = [mjx.synth_jpeg(640, 480, '420', 75, seed=1), slow, mjx.synth_jpeg(333, 217, '444', 85, seed=5), slow,\n"
        "         mjx.synth_jpeg(1920, 1080, '420', 75, seed=2)]\n"
        "for chunk in (0, 2):\n"
        "    b = mjx.Batch(ctx, [mjx.ParsedScan(d) for d in datas], keep_coefs=True, chunk_images=chunk)\n"
        "    b.decode(); b.wait()\n"
        "    for i, d in enumerate(datas):\n"
        "        ref = orc.decode(d, layout=orc.LAYOUT_STD)\n"
        "        assert b.status(i) == 0, (chunk, i, b.status(i))\n"
        "        assert np.array_equal(b.coefs(i), orc.interleave(ref)), (chunk, i)\n"
        "        assert np.abs(b.rgb(i).astype(int) - ref.rgb.astype(int)).max() <= 1, (chunk, i)\n"
        "    b.close()\n"
        "print('slow ok')\n" % (root, root, root))
    for poison, passes in (("255", None), ("165", "2"), ("1", "4")):       # (default: 6 rounds enqueued; the picture needs 12)
        env = dict(os.environ, MJX_POISON=poison)
        if passes:
            env["MJX_FIX_PASSES"] = passes
        out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "slow ok" in out.stdout, poison + ": " + out.stdout[-2000:] + out.stderr[-2000:]


PIL_FIXTURES = {"opt_420_q85.jpg": True, "opt_444_q40.jpg": True, "opt_422_q95.jpg": False, "std_420_q100.jpg": False,
                "opt_gray_q70.jpg": False, "opt_420_q10.jpg": True, "std_420_big.jpg": False,
                "tiny_gray_3x7_q7.jpg": False}      # one byte of entropy data: found by tools/fuzz_parity.py


@pytest.mark.parametrize("name", sorted(PIL_FIXTURES))
def test_libjpeg_written_files(mjx, orc, gpu_ctx, name):
    """Optimised (non Annex-K) Huffman tables, quality 10..100, all samplings; files with a 1-bit code are beyond
    the reference and are checked against the oracle's 1-bit extension."""
    data = open(os.path.join(os.path.dirname(__file__), "golden", "pil", name), "rb").read()
    scan = mjx.ParsedScan(data)
    batch = mjx.Batch(gpu_ctx, [scan], keep_coefs=True)
    batch.decode()
    batch.wait()
    assert batch.status(0) == mjx.OK
    ref = orc.decode(data, layout=orc.LAYOUT_STD, ext_1bit=PIL_FIXTURES[name])
    _check(ref, batch.coefs(0), batch.rgb(0), name)
    batch.close()


# ---- multi-scan (non-interleaved) baseline files -- beyond the reference, which stops after scan 1 ------
MULTISCAN = {"ms_420_big": "std_420_big", "ms_444_q40": "opt_444_q40", "ms_422_q95": "opt_422_q95",
             "ms_420_q85_rst": "opt_420_q85", "ms_420_odd": "dri_420_r5_plain",
             # luma alone, then the two chroma components interleaved ("0; 1 2;", libjpeg wizard.txt)
             "ms2_420_big": "std_420_big", "ms2_420_q85_rst": "opt_420_q85", "ms2_444_q40": "opt_444_q40"}


@pytest.mark.parametrize("name", sorted(MULTISCAN))
def test_multi_scan_files_decode_like_their_interleaved_twins(mjx, orc, gpu_ctx, name):
    """tests/golden/make_multiscan.py re-encodes the coefficients of an interleaved file as one scan per component (blocks in
    the component's raster order, MCU padding blocks dropped, one twin with restart intervals).  Same coefficients, same
    picture: the RGB must be bit-identical to the twin's, which in turn is checked against the oracle."""
    pil = os.path.join(os.path.dirname(__file__), "golden", "pil")
    ms = open(os.path.join(pil, name + ".jpg"), "rb").read()
    src = open(os.path.join(pil, MULTISCAN[name] + ".jpg"), "rb").read()
    scans = [mjx.ParsedScan(src), mjx.ParsedScan(ms), mjx.ParsedScan(ms), mjx.ParsedScan(src)]
    assert scans[1].desc.n_parts == (2 if name.startswith("ms2_") else 3) and scans[0].desc.n_parts == 0
    for chunk in (0, 1, 2):
        batch = mjx.Batch(gpu_ctx, scans, keep_coefs=True, chunk_images=chunk)
        assert len(batch) == 4
        batch.decode()
        batch.wait()
        assert [batch.status(i) for i in range(4)] == [mjx.OK] * 4
        ref = orc.decode(src, layout=orc.LAYOUT_STD, ext_1bit=True)
        _check(ref, batch.coefs(0), batch.rgb(0), name)
        for i in (1, 2, 3):
            assert np.array_equal(batch.rgb(i), batch.rgb(0)), (name, chunk, i)
        # ... and against the oracle's own multi-scan extension (pinned to the twins in tests/test_host.py): the same
        # coefficients block for block, zeros where the interleaved order has MCU padding blocks
        ref_ms = orc.decode(ms, layout=orc.LAYOUT_STD, ext_1bit=True, ext_dri=True, ext_multiscan=True)
        _check(ref_ms, batch.coefs(1), batch.rgb(1), name + " (multi-scan)")
        a, b = batch.coefs(0), batch.coefs(1)
        real = np.abs(b).sum(axis=1) != 0
        assert a.shape == b.shape and np.array_equal(a[real], b[real]) and real.mean() > 0.5
        batch.close()
    # replicated on the device, without kept coefficients, in small chunks: still the twin's picture
    small = mjx.Batch(gpu_ctx, scans[:2], chunk_images=1)
    tiled = small.tile(3)
    assert len(tiled) == 6
    tiled.decode()
    tiled.wait()
    first = tiled.rgb(0)
    for i in range(6):
        assert tiled.status(i) == mjx.OK and np.array_equal(tiled.rgb(i), first), (name, i)
    tiled.close()
    small.close()


def test_multi_scan_errors_stay_with_their_picture(mjx, orc, gpu_ctx, data_dir):
    """A multi-scan file with a truncated chroma scan fails as a whole; its neighbours in the batch are not affected, and
    the bug-compatible modes refuse the format (the reference decodes the first scan only)."""
    pil = os.path.join(os.path.dirname(__file__), "golden", "pil")
    ms = open(os.path.join(pil, "ms_420_big.jpg"), "rb").read()
    sos = [i for i in range(len(ms) - 1) if ms[i] == 0xff and ms[i + 1] == 0xda]
    assert len(sos) == 3
    cut = ms[:sos[2] + 10 + 40] + b"\xff\xd9"                      # third scan: header + 40 bytes of data
    lena = open(os.path.join(data_dir, "lena.jpeg"), "rb").read()
    scans = [mjx.ParsedScan(lena), mjx.ParsedScan(cut), mjx.ParsedScan(ms), mjx.ParsedScan(lena)]
    batch = mjx.Batch(gpu_ctx, scans)
    batch.decode()
    batch.wait()
    st = [batch.status(i) for i in range(4)]
    assert st[0] == st[2] == st[3] == mjx.OK and st[1] == mjx.ERR_TRUNCATED, st
    ref = orc.decode(lena, layout=orc.LAYOUT_STD)
    assert np.abs(batch.rgb(3).astype(int) - ref.rgb.astype(int)).max() <= TOL
    batch.close()
    assert mjx.ParsedScan(ms).validate(layout=mjx.LAYOUT_REF_COMPAT) == mjx.ERR_UNSUPPORTED_FORMAT
    # the one-shot surface (mjx_decode, what the CLI and the Rust binding's decode() call)
    one = mjx.decode(ms)
    twin = mjx.decode(open(os.path.join(pil, "std_420_big.jpg"), "rb").read())
    assert one.shape == (480, 640, 3) and np.array_equal(one, twin)
    img = mjx.JPEGImage.parse(ms)                            # ... and the mirror of the reference's JPEGImage
    assert (img.width(), img.height()) == (640, 480) and np.array_equal(img.image_data(), twin)


# ---- restart intervals (beyond the reference, which panics on DRI) -------------------------------
DRI_FIXTURES = ["dri_420_r5", "dri_444_r1", "dri_422_rows", "dri_gray_r7", "dri_420_720p_rows", "dri_420_r300"]


@pytest.mark.parametrize("name", DRI_FIXTURES)
def test_restart_intervals(mjx, orc, gpu_ctx, name):
    """Every restart interval is an independent segment (own subsequences, known start state, DC predictors from 0).
    Checked against the oracle's ext_dri extension, which is pinned by the libjpeg twin without restart markers."""
    d = os.path.join(os.path.dirname(__file__), "golden", "pil")
    data = open(os.path.join(d, name + ".jpg"), "rb").read()
    plain = open(os.path.join(d, name + "_plain.jpg"), "rb").read()
    batch = mjx.Batch(gpu_ctx, [mjx.ParsedScan(data), mjx.ParsedScan(plain)], keep_coefs=True)
    batch.decode()
    batch.wait()
    assert batch.status(0) == mjx.OK and batch.status(1) == mjx.OK
    ref = orc.decode(data, layout=orc.LAYOUT_STD, ext_dri=True)
    _check(ref, batch.coefs(0), batch.rgb(0), name)
    assert np.array_equal(batch.coefs(0), batch.coefs(1)) and np.array_equal(batch.rgb(0), batch.rgb(1))
    batch.close()


def test_restart_intervals_mixed_batch_and_errors(mjx, orc, gpu_ctx, data_dir):
    d = os.path.join(os.path.dirname(__file__), "golden", "pil")
    dri = [open(os.path.join(d, n + ".jpg"), "rb").read() for n in DRI_FIXTURES]
    others = [open(os.path.join(data_dir, n), "rb").read() for n in FIXTURES] + [mjx.synth_jpeg(1920, 1080, "420", 75, seed=1)]
    datas = [dri[0], others[0], dri[4], others[1], dri[3], others[-1], dri[5]]
    res_dri = {id(x) for x in dri}
    batch = mjx.Batch(gpu_ctx, [mjx.ParsedScan(x) for x in datas], keep_coefs=True, chunk_images=3)
    batch.decode()
    batch.wait()
    for i, x in enumerate(datas):
        ref = orc.decode(x, layout=orc.LAYOUT_STD, ext_dri=id(x) in res_dri)
        assert batch.status(i) == mjx.OK
        _check(ref, batch.coefs(i), batch.rgb(i), "mixed %d" % i)
    batch.close()
    # an interval's marker missing: fewer segments than the MCU count needs -> reported, not decoded
    x = bytearray(dri[0])
    k = x.rindex(b"\xff\xd0") if b"\xff\xd0" in x else None
    cut = bytes(x[:k] + x[k + 2:])
    b2 = mjx.Batch(gpu_ctx, [mjx.ParsedScan(cut)], keep_coefs=True)
    assert b2.status(0) == mjx.ERR_TRUNCATED
    b2.close()
    # REF_COMPAT has no restart intervals (the reference panics on DRI)
    b3 = mjx.Batch(gpu_ctx, [mjx.ParsedScan(dri[0])], layout=mjx.LAYOUT_REF_COMPAT)
    assert b3.status(0) == mjx.ERR_DRI_UNSUPPORTED
    b3.close()


def test_restart_intervals_hostile_inputs(mjx, orc, gpu_ctx):
    """Corrupted files with restart intervals (stray / missing / moved markers, garbage inside intervals): nothing may
    fault on the device, every image gets a status, and an intact image in the same batch is unaffected."""
    d = os.path.join(os.path.dirname(__file__), "golden", "pil")
    rng = np.random.default_rng(3)
    good = open(os.path.join(d, "dri_422_rows.jpg"), "rb").read()
    scans, n_ok_parse = [mjx.ParsedScan(good)], 0
    for name in ("dri_420_r5", "dri_420_720p_rows", "dri_444_r1"):
        base = open(os.path.join(d, name + ".jpg"), "rb").read()
        sos = base.index(b"\xff\xda")
        for k in range(12):
            b = bytearray(base)
            for _ in range(int(rng.integers(1, 10))):
                b[int(rng.integers(sos + 14, len(b)))] = int(rng.choice([0xff, 0xd0, 0xd3, 0x00, int(rng.integers(0, 256))]))
            try:
                scans.append(mjx.ParsedScan(bytes(b)))
                n_ok_parse += 1
            except mjx.MjxError:
                pass
    assert n_ok_parse > 10
    batch = mjx.Batch(gpu_ctx, scans, keep_coefs=True, chunk_images=7)
    batch.decode()
    batch.wait()
    for i in range(1, len(scans)):
        assert batch.status(i) in (mjx.OK, mjx.ERR_TRUNCATED, mjx.ERR_BAD_HUFFMAN, mjx.ERR_INVALID_ARG)
    ref = orc.decode(good, layout=orc.LAYOUT_STD, ext_dri=True)
    assert batch.status(0) == mjx.OK
    _check(ref, batch.coefs(0), batch.rgb(0), "intact image next to corrupted ones")
    batch.close()


# ---- byte de-stuffing on the device ---------------------------------------------------------------
def test_device_side_destuffing_matches_host_destuffing(mjx, orc, gpu_ctx, data_dir):
    datas = [open(os.path.join(data_dir, n), "rb").read() for n in FIXTURES]          # lena.jpeg holds 464 FF00 pairs
    datas += [mjx.synth_jpeg(w, h, s, q, seed=i) for i, (w, h, s, q) in enumerate(
        [(1920, 1080, "420", 95), (640, 480, "444", 98), (333, 222, "422", 90), (64, 64, "gray", 100), (17, 9, "420", 99)])]
    stuffed = [mjx.ParsedScan(d, device_destuff=True) for d in datas]
    plain = [mjx.ParsedScan(d) for d in datas]
    n_pairs = 0
    for s, p in zip(stuffed, plain):
        assert s.desc.scan_is_stuffed == 1 and p.desc.scan_is_stuffed == 0
        assert s.desc.scan_len >= p.desc.scan_len
        n_pairs += s.desc.scan_len - p.desc.scan_len
    assert n_pairs > 1000                                      # the inputs really exercise the compaction
    mixed = [s if i % 2 == 0 else p for i, (s, p) in enumerate(zip(stuffed, plain))]
    for scans in (stuffed, mixed):
        batch = mjx.Batch(gpu_ctx, scans, keep_coefs=True)
        assert all(st == mjx.OK for st in batch.create_status)
        batch.decode()
        batch.wait()
        for i, d in enumerate(datas):
            ref = orc.decode(d, layout=orc.LAYOUT_STD)
            _check(ref, batch.coefs(i), batch.rgb(i), "device destuff image %d" % i)
        batch.close()


def _hostile_stuffed_variants(mjx, data):
    """Entropy-coded bytes that tempt a local (per-byte) de-stuffing rule: fill bytes in front of markers, a lone FF at the very
    end, FF FF 00, markers back to back, a marker as the last two bytes."""
    sos = data.rfind(b"\xff\xda")
    hdr_len = int.from_bytes(data[sos + 2:sos + 4], "big")
    head, body = data[:sos + 2 + hdr_len], data[sos + 2 + hdr_len:]
    out = [head + body[:-2] + b"\xff", head + body[:-2] + b"\xff\xff\x00\xff\xd9", head + body[:len(body) // 2] + b"\xff",
           head + body[:-2] + b"\xff\xd0", head + body[:-2] + b"\xff\xff\xd3\xff\xd4\xff\xd9"]
    k = body.find(b"\xff\x00")
    if k > 0:
        out.append(head + body[:k] + b"\xff\xff\x00" + body[k + 2:])          # a fill byte in front of a stuffed FF
    return out


def test_device_side_destuffing_and_marker_scan_through_every_front_door(mjx, orc, gpu_ctx, data_dir):
    """opts.device_destuff: the host leaves the entropy-coded bytes alone; FF00 compaction, the search for RSTn markers, the
    scan's length and the geometry that follows from it are the device's (k_destuff_*, k_restart_geometry).  Files with and
    without restart intervals (one marker per MCU, per row, a handful), multi-scan files (cut apart on the host as before),
    broken files and hostile byte patterns -- through mjx_decode_batch in one group and pipelined in many, through the batch
    API (ParsedScan), tiled from a de-stuffed base, and through the pool -- must give what the host-side path gives: the same
    statuses, the same coefficients, the same bytes."""
    pil = os.path.join(os.path.dirname(__file__), "golden", "pil")
    names = sorted(n for n in os.listdir(pil) if n.startswith(("dri_", "ms", "slow_sync")) and n.endswith(".jpg"))
    assert len([n for n in names if n.startswith("dri_")]) >= 4 and len([n for n in names if n.startswith("ms")]) >= 2
    datas = [open(os.path.join(pil, n), "rb").read() for n in names]
    datas += [open(os.path.join(data_dir, n), "rb").read() for n in FIXTURES]
    datas += [mjx.synth_jpeg(w, h, s, q, seed=i) for i, (w, h, s, q) in enumerate(
        [(1920, 1080, "420", 95), (640, 480, "444", 98), (64, 64, "gray", 100), (17, 9, "420", 99), (3840, 2160, "420", 75)])]
    dri = open(os.path.join(pil, "dri_420_r5.jpg"), "rb").read()
    datas += _hostile_stuffed_variants(mjx, dri) + _hostile_stuffed_variants(mjx, datas[-2])
    datas += [b"not a jpeg", dri[:len(dri) // 2], dri[:700]]
    host, st_host = mjx.decode_batch(gpu_ctx, datas, keep_coefs=True, device_destuff=False)
    n_ok = sum(1 for s in st_host if s == mjx.OK)
    assert n_ok >= len(datas) - 8
    # (MJX_DESTUFF_AUTO, what a zeroed mjx_opts says: the GPU for lists of MJX_AUTO_DESTUFF_MB and more -- here: for any list)
    for env in ({}, {"MJX_GROUP_MB": "1"}, {"MJX_AUTO_DESTUFF_MB": "0", "MJX_GROUP_MB": "2"}):
        os.environ.update(env)
        try:
            for keep in (True, False):
                if env and keep:
                    continue                                      # (kept coefficients: one group anyway)
                dev, st_dev = mjx.decode_batch(gpu_ctx, datas, device_destuff=None if "MJX_AUTO_DESTUFF_MB" in env else True, keep_coefs=keep, threads=3)
                assert st_dev == st_host, [(i, a, b) for i, (a, b) in enumerate(zip(st_dev, st_host)) if a != b]
                ok = [i for i, s in enumerate(st_host) if s == mjx.OK]
                mx, cnt = dev.compare_rgb(ok, host, ok)
                assert int(mx.max()) == 0, "pictures %s differ from the host-de-stuffed decode" % [ok[j] for j in np.nonzero(mx)[0][:8]]
                if keep:
                    for i in ok:
                        assert np.array_equal(dev.coefs(i), host.coefs(i)), i
                dev.close()
        finally:
            for k in env:
                del os.environ[k]
    # against the oracle (the host path is compared with it elsewhere; here: the device path directly, restart files included)
    for i in (0, 1, len(names) + 2, len(names) + 4):
        ref = orc.decode(datas[i], layout=orc.LAYOUT_STD, ext_dri=True, ext_multiscan=True)
        assert np.array_equal(host.coefs(i), orc.interleave(ref)), i
    # the batch API: stuffed scans (restart files too), a base that is tiled, REF_COMPAT geometry on a stuffed scan
    picks = [i for i, s in enumerate(st_host) if s == mjx.OK][:12]
    scans = [mjx.ParsedScan(datas[i], device_destuff=True) for i in picks]
    base = mjx.Batch(gpu_ctx, scans)
    big = base.tile(3)
    big.decode()
    big.wait()
    idx = list(range(len(big)))
    mx, cnt = big.compare_rgb(idx, host, [picks[i % len(picks)] for i in idx])
    assert int(mx.max()) == 0
    big.close()
    base.close()
    two = open(os.path.join(data_dir, "2x2-chroma.jpeg"), "rb").read()
    (ref, coefs, rgb), = _decode_both(mjx, orc, gpu_ctx, [two], layout_std=False)
    b2 = mjx.Batch(gpu_ctx, [mjx.ParsedScan(two, device_destuff=True)], layout=mjx.LAYOUT_REF_COMPAT)
    b2.decode()
    b2.wait()
    assert b2.status(0) == mjx.OK and np.array_equal(b2.rgb(0), rgb)
    b2.close()
    # the pool hands the option to every slot
    pool = mjx.Pool([0, 0])
    res = pool.decode_batch(datas, threads_per_device=2, device_destuff=True)
    assert res.status == st_host
    for i in picks[:6]:
        assert np.array_equal(res.rgb(i), host.rgb(i)), i
    res.close()
    pool.close()
    host.close()
