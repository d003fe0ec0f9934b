#!/bin/bash
# A/B runs on ONE GPU box (boxes differ by a few per cent: never compare across gpurun calls).  One script since round 6; it replaces the
# family ab.sh / abn.sh / abx.sh / abv.sh / abv1.sh / abv2.sh / abp.sh / abe.sh / abe_cfg.sh / ab_emit.sh / ab_stageb.sh of rounds 2-5.
#
#   tools/ab.sh [-l "lib1 lib2 ..."] [-e "ENV=a ENV2=b" -e "ENV=c" ...] [-m default|one|both|pixels] [-r rounds] [-g] [-- bench args]
#
#   -l  builds to compare (tools/build_variant.sh NAME -D... writes ab/libmjx_NAME.so); default: the library as built
#   -e  an environment setting to compare (repeatable; "-" = none); every build runs under every setting
#   -m  default = the library's default streams (the headline's mode), one = MJX_STREAMS=1 (no overlap: per-kernel times mean something),
#       both = one after the other, pixels = stage B alone on BASELINE config 4 (4096 x 1080p, --stages pixels)
#   -r  passes over the whole set (default 2: the second pass shows the box's drift)
#   -g  measurement builds that decode garbage: statuses and parity are not looked at, pictures stay on their path
#   bench args follow "--" (e.g. -- --quality 90, -- --width 1920 --height 1080 --images-per-gpu 4096)
# Prints per run: build | setting | mode, Mpixels/s, ms per step, per-kernel-class ms per step, geometry of the batch.
LIBS="jpeg-rust_amd/libmjx.so"; ENVS=(); MODE=default; ROUNDS=2; GARBAGE=""
while [ $# -gt 0 ]; do case "$1" in
  -l) LIBS=$2; shift 2;; -e) ENVS+=("$2"); shift 2;; -m) MODE=$2; shift 2;; -r) ROUNDS=$2; shift 2;; -g) GARBAGE="MJX_BENCH_IGNORE_STATUS=1 MJX_EXP_NO_FALLBACK=1"; shift;;
  --) shift; break;; *) echo "unknown option $1" >&2; exit 2;; esac; done
[ ${#ENVS[@]} -eq 0 ] && ENVS=("-")
show() { grep '^{' | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(sys.argv[1], '|', sys.argv[2], '|', sys.argv[3], round(d['value']), d['ms_per_step'], {k: round(v['ms'] / d['steps'], 2) for k, v in d['kernels'].items()},
      'sub_bytes', d['config'].get('subsequence_bytes'), 'chunks', d['config'].get('chunks_per_step'), 'tiled_diff', d.get('parity', {}).get('tiled_max_abs_diff'))" "$1" "$2" "$3"; }
run() {   # lib, env, mode
  local extra="" pre=""
  case "$3" in one) pre="MJX_STREAMS=1";; pixels) extra="--stages pixels --width 1920 --height 1080 --images-per-gpu 4096 --steps 5";; esac
  local e="$2"; [ "$e" = "-" ] && e=""
  env $e $pre $GARBAGE MJX_LIB=$PWD/$1 timeout 900 python3 bench.py --no-cpu-baseline --no-extra --no-parity $extra "${@:4}" 2>/dev/null | show "$1" "$2" "$3"
}
for r in $(seq $ROUNDS); do for L in $LIBS; do for E in "${ENVS[@]}"; do
  if [ "$MODE" = both ]; then run "$L" "$E" one "$@"; run "$L" "$E" default "$@"; else run "$L" "$E" "$MODE" "$@"; fi
done; done; done
