"""Per-kernel HBM bytes per launch from the two PMC passes of tools/pmc_traffic.sh.

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 KB per count as rocprofv3
reports them; on gfx950 FETCH_SIZE tallies each 128-B request at 64 B, i.e. reports exactly 1/2 of a wide coalesced streaming
read -> doubled here.  WRITE_SIZE is uncalibrated in the guide: it is calibrated in-run against `ln_mod_kernel`, whose write
volume is known exactly (rows x D bf16, every byte written once), and the factor is recorded in the output.
"""
import collections
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1]


def load(kind):
    agg = collections.defaultdict(list)
    grids = {}
    for f in glob.glob(os.path.join(out_dir, "**", f"*{kind}_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            key = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            agg[(key, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return agg


fetch, write = load("fetch"), load("write")
KB = 1024.0
# calibration kernel: ln_mod_kernel over the joint buffer [B*S, D] (B=4, S=3226, D=1920): writes rows*D*2 bytes
cal = None
for (k, g), v in write.items():
    if k.startswith("ln_mod_kernel") and len(v) >= 30:
        rows = 4 * 3226
        known = rows * 1920 * 2
        meas = sum(v) / len(v) * KB
        if meas > 0 and 0.2 < known / meas < 5:
            cal = known / meas
            break
res = {"_units": "bytes per launch", "_fetch_correction": 2.0, "_write_calibration": cal,
       "_note": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE x calibration from ln_mod_kernel's known write volume"}
rows_out = []
for (k, g) in sorted(set(fetch) | set(write), key=lambda kg: -sum(fetch.get(kg, [0]))):
    f = fetch.get((k, g), [])
    w = write.get((k, g), [])
    if len(f) < 2 and len(w) < 2:
        continue
    fb = 2.0 * KB * sum(f) / len(f) if f else None
    wb = (cal or 1.0) * KB * sum(w) / len(w) if w else None
    rows_out.append((k, g, len(f), fb, wb))
    ent = {"grid": g, "launches": len(f), "fetch_bytes": fb, "write_bytes": wb,
           "total_bytes": (fb or 0) + (wb or 0)}
    res.setdefault(k, ent) if k not in res else None
    res.setdefault("by_grid", {})[f"{k}@{g}"] = ent
json.dump(res, open(os.path.join(out_dir, "hbm_traffic.json"), "w"), indent=1)
print(f"write calibration factor (ln_mod_kernel known bytes / WRITE_SIZE): {cal}")
print(f"{'kernel':60s} {'grid':>9s} {'n':>5s} {'fetch MB':>10s} {'write MB':>10s}")
for k, g, n, fb, wb in rows_out[:24]:
    print(f"{k[:60]:60s} {g:9d} {n:5d} {(fb or 0)/1e6:10.2f} {(wb or 0)/1e6:10.2f}")
