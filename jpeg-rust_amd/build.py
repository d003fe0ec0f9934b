"""In-tree build of the native pieces (no JIT cache: the built .so files travel with the repo snapshot).

    libmjx.so            product: host parse + HIP kernels + C ABI   (hipcc --offload-arch=gfx950)
    synth/libmjx_synth.so synthetic baseline-JPEG generator           (gcc)
    ../oracle/libmjx_oracle.so  CPU oracle, test infrastructure only  (make, gcc)
    ../tests/emul/libhuff_emul.so CPU emulation of the entropy algorithm, test infrastructure only (g++)
"""
import contextlib
import fcntl
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")

HIP_SOURCES = ["mjx_kernels.hip", "mjx_api.hip"]
CXX_SOURCES = ["mjx_parse.cpp", "mjx_lut.cpp", "mjx_plan.cpp", "mjx_pool.cpp"]
HEADERS = ["mjx_huff.h", "mjx_kernels.h", "mjx_plan.h", os.path.join(ROOT, "include", "mjx.h")]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, cwd=None):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build step failed: " + " ".join(cmd))
    return r.stdout


@contextlib.contextmanager
def _build_lock():
    """One builder at a time: `torchrun` starts N ranks together and each calls build(); the first one in builds, the
    others wait here and then find everything up to date."""
    with open(os.path.join(PKG, ".build.lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def _link(cmd, out):
    """Runs a link command that writes `out` (named after "-o") into a temporary name and renames it into place: a
    process that has the old file mapped keeps its inode."""
    tmp = "%s.tmp.%d" % (out, os.getpid())
    cmd = [tmp if c == out else c for c in cmd]
    try:
        log = _run(cmd)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return log


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def build_product(force=False, verbose=False):
    out = os.path.join(PKG, "libmjx.so")
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES + CXX_SOURCES]
    deps = srcs + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    if not force and not _newer(out, deps):
        return out
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    objs, log = [], ""
    for s in HIP_SOURCES:      # device + host code: hipcc cross-compiles gfx950 without a GPU
        o = os.path.join(objdir, s + ".o")
        log += _run([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-c", os.path.join(CSRC, s), "-o", o] + inc)
        objs.append(o)
    for s in CXX_SOURCES:      # host-only code: plain C++
        o = os.path.join(objdir, s + ".o")
        log += _run(["g++", "-O2", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, s), "-o", o] + inc)
        objs.append(o)
    log += _link([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, out)
    if verbose:
        print(log)
    return out


def build_synth(force=False):
    src = os.path.join(PKG, "synth", "mjx_synth.c")
    out = os.path.join(PKG, "synth", "libmjx_synth.so")
    if force or _newer(out, [src]):
        _link(["gcc", "-O2", "-fPIC", "-shared", "-o", out, src, "-lm"], out)
    return out


def build_oracle(force=False):
    d = os.path.join(ROOT, "oracle")
    out = os.path.join(d, "libmjx_oracle.so")
    if force or _newer(out, [os.path.join(d, "mjx_oracle.c"), os.path.join(d, "mjx_oracle.h")]):
        _run(["make", "-C", d, "-B"])
    return out


def build_emul(force=False):
    d = os.path.join(ROOT, "tests", "emul")
    src = os.path.join(d, "huff_emul.cpp")
    out = os.path.join(d, "libhuff_emul.so")
    host_only = [s for s in CXX_SOURCES if s != "mjx_pool.cpp"]      # (the pool needs the device entry points)
    deps = [src] + [os.path.join(CSRC, s) for s in host_only + ["mjx_huff.h", "mjx_plan.h"]]
    if os.path.exists(src) and (force or _newer(out, deps)):
        _link(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
               "-o", out, src] + [os.path.join(CSRC, s) for s in host_only], out)
    return out


def build_cli(force=False):
    """mjx_cli: the reference's main.rs counterpart, linked against libmjx.so (rpath = package dir)."""
    src = os.path.join(CSRC, "mjx_cli.cpp")
    out = os.path.join(PKG, "mjx_cli")
    if force or _newer(out, [src, os.path.join(CSRC, "jpeg.hpp"), os.path.join(PKG, "libmjx.so")]):
        _link(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", out, src,
               "-L" + PKG, "-lmjx", "-Wl,-rpath," + PKG, "-Wl,-rpath,$ORIGIN"], out)
    return out


def build_all(force=False, verbose=False):
    with _build_lock():
        return {
            "libmjx": build_product(force, verbose),
            "cli": build_cli(force),
            "synth": build_synth(force),
            "oracle": build_oracle(force),
            "emul": build_emul(force),
        }


if __name__ == "__main__":
    for k, v in build_all(force="--force" in sys.argv, verbose=True).items():
        print(k, "->", v)
