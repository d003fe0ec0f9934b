"""Builds profiles/<name>_traffic.json from two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE collected in separate
passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) of `bench.py --images-per-gpu 128` (one chunk = one launch
per kernel).  FETCH_SIZE is doubled: on gfx950 it reports half the bytes of wide coalesced reads (guide, section HBM);
for the narrow reads of the entropy kernels that correction is an upper bound.  Usage:
    python tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01b_traffic.json [images per launch]"""
import collections, csv, glob, json, sys

def per_launch(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void mjx::", "").split("<")[0]
            acc[k][0] += float(r["Counter_Value"]) * 1024.0
            acc[k][1] += 1
    return {k: v / n for k, (v, n) in acc.items() if not k.startswith("__amd")}

fetch, write = per_launch(sys.argv[1], "FETCH_SIZE"), per_launch(sys.argv[2], "WRITE_SIZE")
alias = {"k_huff_spec": "huff_sync", "k_huff_merge": "huff_fix", "k_huff_merge_tail": "huff_fix_tail", "k_huff_scan": "huff_scan",
         "k_huff_write": "huff_write", "k_idct_color": "idct_color", "k_dc_sums_t": "dc_sums", "k_dc_apply_t": "dc_apply", "k_dc_scan_t": "dc_scan",
         "k_huff_merge_loop": "huff_fix_loop"}
out = {"images_per_launch": int(sys.argv[4]) if len(sys.argv) > 4 else 128, "note": "bytes per launch; fetch = 2 x FETCH_SIZE (gfx950 correction), write = WRITE_SIZE",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    out["kernels"][alias.get(k, k)] = {"fetch_bytes": int(2 * fetch.get(k, 0)), "write_bytes": int(write.get(k, 0)),
                                       "hbm_bytes": int(2 * fetch.get(k, 0) + write.get(k, 0))}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
