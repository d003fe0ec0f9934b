"""A/B of mjx_decode_batch settings on one box, alternating: python tools/e2e_ab.py [files] -- each setting is an environment for a fresh context."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
mjx = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
uniq = [mjx.synth_jpeg(3840, 2160, "420", 75, seed=s) for s in range(64)]
datas = [uniq[i % 64] for i in range(n)]
settings = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[2:]] or [{}]
res = {i: [] for i in range(len(settings))}
for rnd in range(3):
    for i, env in enumerate(settings):
        for k in ("MJX_UPLOAD_APART", "MJX_GROUP_ALT", "MJX_GROUP_MB", "MJX_GROUP_FIRST_MB", "MJX_GROUP_GROW"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ctx = mjx.Context(0)
        b, st = mjx.decode_batch(ctx, datas[:8]); b.close()
        b, st = mjx.decode_batch(ctx, datas); b.close()
        for rep in range(4):
            t = time.perf_counter()
            b, st = mjx.decode_batch(ctx, datas)
            res[i].append(time.perf_counter() - t)
            assert all(s == mjx.OK for s in st)
            b.close()
        ctx.close()
for i, env in enumerate(settings):
    v = sorted(res[i])
    print("%-60s best %.2f ms  median %.2f ms  = %.1f Gpx/s (best)" % (env, v[0] * 1e3, v[len(v) // 2] * 1e3, n * 3840 * 2160 / v[0] / 1e9))
