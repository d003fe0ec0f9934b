"""The kernels' register and scratch budgets, read from the compiler (CPU test: hipcc cross-compiles gfx950 without a GPU).

Round 6 learned this the slow way: two `uint4` members added to a sink struct were left in scratch memory by the compiler and the
emitting entropy pass went from 28.1 to 30.4 ms per step -- nothing failed, nothing warned.  `-Rpass-analysis=kernel-resource-usage`
says it in one line per kernel.  What the design relies on:
  * no kernel uses scratch memory or spills vector registers; the hot kernels spill no scalar registers either;
  * stage B's 4:2:0 forms stay at or under 168 VGPRs (three waves per SIMD: its 52 KB tiles allow three workgroups per CU, and a
    fourth wave's worth of registers would be wasted -- but 169 would cost the third);
  * the entropy kernels stay at or under 128 VGPRs (four waves per SIMD = two 512-lane workgroups per CU, what their LDS allows);
  * every kernel's SGPR count stays within the wave's 102 (+ the special pairs the remark counts on top).
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "jpeg-rust_amd", "csrc")


def _hipcc():
    for cand in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(cand) or cand == "hipcc":
            return cand


@pytest.fixture(scope="module")
def usage(tmp_path_factory):
    out = tmp_path_factory.mktemp("res") / "k.o"
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage",
           "-c", os.path.join(CSRC, "mjx_kernels.hip"), "-o", str(out), "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    kernels, cur = {}, None
    for line in (r.stdout + r.stderr).splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"\s([A-Za-z][A-Za-z ]*?)(?: \[[^\]]*\])?: (\d+) \[-Rpass-analysis", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    assert len(kernels) > 20, list(kernels)[:5]
    return kernels


def test_no_kernel_uses_scratch_or_spills(usage):
    # (scalar registers that do not fit go to lanes of a vector register -- v_writelane --, not to memory: a few kernels with long
    # uniform preambles have some, the hot 4:2:0 and entropy kernels must have none)
    bad = {k: v for k, v in usage.items() if v.get("ScratchSize", 0) or v.get("VGPRs Spill", 0)}
    assert not bad, bad
    hot = {k: v.get("SGPRs Spill", 0) for k, v in usage.items() if "k_idct_colorILi1E" in k or k in ("k_huff_emit", "k_huff_spec", "k_huff_write", "k_huff_merge")}
    assert hot and not any(hot.values()), hot


def test_register_budgets_keep_the_occupancy_the_design_counts_on(usage):
    for name, v in usage.items():
        assert v["TotalSGPRs"] <= 108, (name, v)                           # (102 + VCC + flat-scratch / XNACK pairs as the remark counts them)
        if "k_idct_color" in name:
            assert v["VGPRs"] <= 168, (name, v["VGPRs"])                    # three waves per SIMD
        elif name.startswith("k_huff_") or "k_dc_" in name or name in ("k_block_gather",):
            assert v["VGPRs"] <= 128, (name, v["VGPRs"])                    # four waves per SIMD
    emit = usage["k_huff_emit"]
    assert emit["VGPRs"] <= 96, emit                                        # (73 today: room, but not a licence)
    stage_b = [v["VGPRs"] for k, v in usage.items() if "k_idct_colorILi1ELi8ELi1E" in k]
    assert stage_b and stage_b[0] <= 160, stage_b                           # (146 today)
