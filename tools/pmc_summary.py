#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE counter CSVs (two separate passes, as MI355X_MICROARCH.md prescribes)
-> per-kernel HBM traffic summary JSON.

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Units/corrections (MI355X_MICROARCH.md, HBM section): the counters are in KB; on gfx950 FETCH_SIZE reports exactly half the
bytes of a wide coalesced read stream, so fetch bytes = 2 * FETCH_SIZE * 1024 (calibrated for 16 B/lane streams; other
widths are uncalibrated -- ratios between builds stay valid).  WRITE_SIZE is exact for 16 B/lane stores.
"""
import collections
import csv
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = per_kernel(fetch), per_kernel(write)
    res = {}
    for k in sorted(set(f) | set(w)):
        fk, nf = f.get(k, (0.0, 0))
        wk, nw = w.get(k, (0.0, 0))
        res[k] = {"dispatches": max(nf, nw), "FETCH_SIZE_KB": round(fk, 1), "WRITE_SIZE_KB": round(wk, 1),
                  "hbm_bytes_per_dispatch": round((2.0 * fk + wk) * 1024)}
    json.dump({"note": "fetch doubled per the gfx950 correction; per-dispatch means", "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_dispatch"])[:12]:
        print(f"{k[:60]:60s} {v['hbm_bytes_per_dispatch'] / 1e6:8.2f} MB/dispatch  (n={v['dispatches']})")


if __name__ == "__main__":
    main()
