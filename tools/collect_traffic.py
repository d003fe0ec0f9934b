"""Builds profiles/<name>_traffic.json from two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE collected in separate
passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) of ONE step of `bench.py --steps 1 --warmup 0 --images-per-gpu N`
on one stream.  FETCH_SIZE is doubled: on gfx950 it reports half the bytes of wide coalesced reads (guide, section HBM);
for the narrow reads of the entropy kernels that correction is an upper bound.

The figures are SUMS over the step's dispatches of each kernel class ("basis": "per_step"), with the number of launches and
the pictures of the step recorded beside them: a batch that is cut into two chunks launches every kernel twice, and an average
per launch under "images_per_launch = the whole batch" reported half of every kernel's bytes (round-4 review, weak #7).
bench.py's committed_traffic() accepts only "per_step" collections.  Usage:
    python tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r05_traffic.json <images per step> [W H]
With W H the script checks idct_color.write_bytes against 3*W*H*images (+-1 %) and fails loudly when it is off."""
import collections, csv, glob, json, sys

ALIAS = {"k_huff_spec": "huff_sync", "k_huff_merge": "huff_fix", "k_huff_merge_tail": "huff_fix", "k_huff_merge_loop": "huff_fix",
         "k_huff_scan": "huff_scan", "k_huff_write": "huff_write", "k_huff_emit": "huff_emit", "k_huff_prefix": "huff_prefix", "k_block_gather": "huff_prefix",
         "k_idct_color": "idct_color", "k_ref_color": "idct_color",
         "k_dc_sums_t": "dc_scan", "k_dc_apply_t": "dc_scan", "k_dc_scan_t": "dc_scan", "k_dc_sums": "dc_scan", "k_dc_apply": "dc_scan",
         "k_dc_restart": "dc_scan", "k_planar_count": "gather", "k_planar_offsets": "gather", "k_planar_copy": "gather",
         "k_scan_interleave": "upload", "k_destuff_count": "upload", "k_destuff_prefix": "upload", "k_destuff_scatter": "upload",
         "k_restart_geometry": "upload"}


def per_step(d, counter):
    acc = collections.defaultdict(lambda: [0.0, set()])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void mjx::", "").split("<")[0].strip()
            if k.startswith("__amd") or k not in ALIAS:
                continue
            acc[ALIAS[k]][0] += float(r["Counter_Value"]) * 1024.0
            acc[ALIAS[k]][1].add(r["Dispatch_Id"])
    return {k: (v, len(ids)) for k, (v, ids) in acc.items()}


def main():
    fetch, write = per_step(sys.argv[1], "FETCH_SIZE"), per_step(sys.argv[2], "WRITE_SIZE")
    images = int(sys.argv[4])
    out = {"basis": "per_step", "images_per_step": images,
           "note": "bytes per step (sum over the step's dispatches of each class; one stream, one step, no warm-up); "
                   "fetch = 2 x FETCH_SIZE (gfx950 correction), write = WRITE_SIZE; `launches` = dispatches of the class in the step",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        if nf and nw and nf != nw:
            raise SystemExit("class %s: %d dispatches in the FETCH pass, %d in the WRITE pass -- not the same step" % (k, nf, nw))
        out["kernels"][k] = {"fetch_bytes": int(2 * f), "write_bytes": int(w), "hbm_bytes": int(2 * f + w), "launches": max(nf, nw)}
    if len(sys.argv) > 6 and "idct_color" in out["kernels"]:
        want = 3 * int(sys.argv[5]) * int(sys.argv[6]) * images
        got = out["kernels"]["idct_color"]["write_bytes"]
        out["check"] = {"idct_color_write_bytes": got, "rgb_bytes_3WH_images": want, "ratio": round(got / want, 4)}
        if abs(got / want - 1.0) > 0.01:
            json.dump(out, sys.stdout, indent=1)
            raise SystemExit("\nidct_color wrote %.3f x the pictures' bytes: the collection does not cover exactly one step" % (got / want))
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
