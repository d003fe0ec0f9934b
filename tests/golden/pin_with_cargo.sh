#!/bin/bash
# Pins the CPU oracle (oracle/mjx_oracle.c) against the REAL reference, for anyone who has a Rust toolchain -- this image has
# none (no rustc / cargo, no network for itertools 0.4.18), which is the only reason DESIGN.md s2 says "parity unpinned".
#
#   tests/golden/pin_with_cargo.sh /path/to/jpeg-rust [--gpu]
#
# What it does (the reference's own recipe is Makefile:4-7: `cargo run X.jpeg X-gen.ppm` + a visual diff; src/main.rs:35-39 writes
# the byte-comparable artefact: "P3\n{w} {h}\n255\n" and one "r g b\n" line per pixel):
#   1. `cargo build --release` in the reference checkout (Cargo.lock pins itertools 0.4.18; edition 2015: `try!`, `extern crate`);
#   2. runs the reference binary on the three samples it can decode -- lena.jpeg, working-jpegs/lena-bw.jpeg, 2x2-chroma.jpeg
#      (working-jpegs/huff_simple0.jpg panics in its parser: APP12/APP14; the script checks that it does);
#   3. hashes the pictures (SHA-256 over the R,G,B bytes in row-major order) and compares them with
#        a. tests/golden/ref_emul_golden.json  (the answers both restatements agree on, committed)   -> must be EQUAL
#        b. the oracle run now, ORC_LAYOUT_REF, faithful cosf                                         -> must be EQUAL, byte for byte
#        c. with --gpu: `mjx_cli --ref-compat` on this machine's GPU                                  -> every sample within 1 LSB
#   4. prints PINNED / NOT PINNED and exits 0 / 1.
# A PINNED run is what turns SURVEY s8(c) from "partial" into "yes": commit its output as tests/golden/pin_result.txt.
set -u
REF=${1:?usage: pin_with_cargo.sh /path/to/jpeg-rust [--gpu]}
GPU=${2:-}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
OUT=$(mktemp -d /tmp/mjx_pin.XXXXXX)
command -v cargo >/dev/null || { echo "cargo not found: this recipe needs a Rust toolchain (any rustc that still accepts edition 2015)"; exit 2; }
( cd "$REF" && cargo build --release ) || { echo "cargo build failed"; exit 2; }
BIN="$REF/target/release/jpeg-rust"
[ -x "$BIN" ] || BIN=$(ls "$REF"/target/release/* 2>/dev/null | while read f; do [ -x "$f" ] && [ -f "$f" ] && echo "$f"; done | head -1)
[ -x "$BIN" ] || { echo "no reference binary under $REF/target/release"; exit 2; }
make -C "$ROOT/oracle" >/dev/null || exit 2
fail=0
for f in lena.jpeg working-jpegs/lena-bw.jpeg 2x2-chroma.jpeg; do
    n=$(basename "$f")
    ( cd "$REF" && RUST_BACKTRACE=1 "$BIN" "$f" "$OUT/$n.ref.ppm" ) || { echo "$n: the reference failed on a sample it should decode"; fail=1; continue; }
    args=("$OUT/$n.ref.ppm" "$REF/$f" "$n")
    if [ "$GPU" = "--gpu" ]; then
        "$ROOT/jpeg-rust_amd/mjx_cli" "$REF/$f" "$OUT/$n.gpu.ppm" --ref-compat || { echo "$n: mjx_cli failed"; fail=1; continue; }
        args+=("$OUT/$n.gpu.ppm")
    fi
    python3 "$HERE/pin_compare.py" "${args[@]}" || fail=1
done
# Q1: the reference panics on huff_simple0.jpg (APP12 at offset 20)
if ( cd "$REF" && "$BIN" working-jpegs/huff_simple0.jpg "$OUT/hs.ppm" ) >/dev/null 2>&1; then
    echo "huff_simple0.jpg: the reference decoded it -- SURVEY Q1 says it panics on APP12/APP14 at this commit; check the checkout"; fail=1
else
    echo "huff_simple0.jpg: reference panics as SURVEY Q1 records (oracle: strict_ref=1 -> ORC_ERR_UNSUPPORTED)"
fi
rm -rf "$OUT"
if [ $fail = 0 ]; then echo "PINNED: the oracle equals the reference byte for byte on its three decodable samples"; else echo "NOT PINNED"; fi
exit $fail
