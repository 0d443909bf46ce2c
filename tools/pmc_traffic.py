"""Dev tool: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) -> profiles/<name>.json.

usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>
HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB as reported; the factor 2 is the gfx950 correction for wide
16-B/lane streams prescribed by MI355X_MICROARCH.md's HBM section)."""
import csv, glob, json, sys

def load(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter or not r["Kernel_Name"].startswith(("void conv_", "conv_")):
            continue
        key = (r["Kernel_Name"].replace("void ", "").split("(")[0], int(r["Grid_Size"]))
        a = acc.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    return {k: (v[1] / v[0], v[0]) for k, v in acc.items()}

fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace) over tools/bench_embed.py "
                "256 (IResNet-100 forward, 256 faces), KiB per dispatch averaged over dispatches of the same kernel and grid. "
                "Correction per MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of wide coalesced "
                "16-B/lane streams -> hbm_read = 2*FETCH_SIZE; WRITE_SIZE is exact for 16-B/lane streaming stores.",
       "kernels": {}}
for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, (0, 0))[0] + wr.get(k, (0, 0))[0])):
    f, w = fe.get(k, (None, 0))[0], wr.get(k, (None, 0))[0]
    rec = {"grid_threads": k[1], "dispatches": max(fe.get(k, (0, 0))[1], wr.get(k, (0, 0))[1]),
           "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w}
    if f is not None and w is not None:
        rec["hbm_bytes_per_launch_corrected"] = int((2 * f + w) * 1024)
    out["kernels"][f"{k[0]} grid={k[1]}"] = rec
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out["kernels"].items():
    print(k[:70], v.get("hbm_bytes_per_launch_corrected"))
