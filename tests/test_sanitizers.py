"""Sanitizer builds of everything that runs on the host (round-4 review, missing #5).  The reference leans on Rust's
bounds checks -- an out-of-range index is a panic (src/jpeg/decoder.rs:370-371, src/jpeg/huffman.rs:240-247), never a wild read --
so the C / C++ that replaces it is run under AddressSanitizer + UBSan over the fixtures and mutated copies of them, and the
multi-GPU front's threads under ThreadSanitizer:

  * tools/sanitize/asan_parse_fuzz.cpp   mjx_parse + planning (the host side of the C ABI) over mutated files
  * tools/sanitize/asan_emul_fuzz.cpp    the per-lane entropy routine of the kernels (tests/emul) over mutated tables and scans
  * tools/sanitize/asan_oracle_run.c     the CPU oracle itself (test infrastructure must not lie because of a stray write)
  * tools/sanitize/tsan_pool_stub.cpp    jpeg-rust_amd/csrc/mjx_pool.cpp against stubs of the single-device ABI: 8 slots, a failing
                                         slot, an empty list, two callers at once

CPU only (GPU AddressSanitizer is not available on the pool); sized to finish in about a minute in all."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "jpeg-rust_amd", "csrc")
SAN = os.path.join(ROOT, "tools", "sanitize")
INC = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I" + os.path.join(ROOT, "oracle")]
ASAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
           TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
ENV.pop("LD_PRELOAD", None)


def fixtures(patterns):
    out = []
    for p in patterns:
        out += sorted(glob.glob(os.path.join(ROOT, "tests", p)))
    assert out, patterns
    return out


def build(tmp_path, name, cmd):
    exe = str(tmp_path / name)
    r = subprocess.run(cmd + ["-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return exe


def run(exe, args, timeout=600):
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=timeout, env=ENV, cwd=ROOT)
    assert r.returncode == 0, "sanitizer report or failure:\n" + r.stdout[-2000:] + r.stderr[-6000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
    return r.stdout


def test_host_parser_and_planner_under_asan_ubsan(tmp_path):
    exe = build(tmp_path, "asan_parse_fuzz", ["g++", "-std=c++17"] + ASAN + INC + [
        os.path.join(SAN, "asan_parse_fuzz.cpp"), os.path.join(CSRC, "mjx_parse.cpp"), os.path.join(CSRC, "mjx_plan.cpp"),
        os.path.join(CSRC, "mjx_lut.cpp")])
    files = [f for f in fixtures(["data/*", "golden/pil/*.jpg"]) if os.path.getsize(f) < 200000]
    out = run(exe, ["60"] + files)
    assert "asan parse fuzz:" in out and int(out.split()[3]) > 3000, out


def test_entropy_routine_of_the_kernels_under_asan_ubsan(tmp_path):
    exe = build(tmp_path, "asan_emul_fuzz", ["g++", "-std=c++17"] + ASAN + INC + [
        os.path.join(SAN, "asan_emul_fuzz.cpp"), os.path.join(ROOT, "tests", "emul", "huff_emul.cpp"), os.path.join(CSRC, "mjx_parse.cpp"),
        os.path.join(CSRC, "mjx_plan.cpp"), os.path.join(CSRC, "mjx_lut.cpp")])
    out = run(exe, ["12"] + fixtures(["golden/pil/opt_*.jpg", "golden/pil/tiny*.jpg", "data/*.jp*"]))
    assert "asan emul fuzz:" in out and int(out.split()[3]) > 50, out


def test_oracle_under_asan_ubsan(tmp_path):
    exe = build(tmp_path, "asan_oracle_run", ["gcc", "-std=c11"] + ASAN + ["-ffp-contract=off", "-fno-fast-math"] + INC + [
        os.path.join(SAN, "asan_oracle_run.c"), os.path.join(ROOT, "oracle", "mjx_oracle.c"), "-lm", "-lpthread"])
    files = [f for f in fixtures(["data/*", "golden/pil/*.jpg"]) if os.path.getsize(f) < 200000]
    out = run(exe, ["25"] + files)
    assert "asan oracle run:" in out, out
    decodes, pictures = int(out.split()[3]), int(out.split()[5])
    assert decodes > 800 and pictures > 100, out


def test_pool_threads_under_tsan(tmp_path):
    exe = build(tmp_path, "tsan_pool_stub", ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread"] + INC + [
        os.path.join(SAN, "tsan_pool_stub.cpp"), os.path.join(CSRC, "mjx_pool.cpp"), "-pthread"])
    out = run(exe, [])
    assert "tsan pool stub: ok" in out, out
