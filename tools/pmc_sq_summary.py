#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_* counter CSV (tools/pmc_mfma.sh) -> per-kernel sums + matrix-pipe busy fraction and LDS bank-conflict fraction.

    python tools/pmc_sq_summary.py <m_counter_collection.csv> <out.json>

mfma_busy_frac_of_simd_cycles = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); lds_bank_conflict_frac =
SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import collections
import csv
import json
import sys


def main():
    src, out = sys.argv[1:3]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(src)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    res = {}
    for k, c in agg.items():
        d = {"dispatches": len(disp[k])}
        d.update({n: int(v) for n, v in sorted(c.items())})
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        d["mfma_busy_frac_of_simd_cycles"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024), 4) if gui else None
        act = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
        d["lds_bank_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / act, 4) if act else 0.0
        res[k] = d
    json.dump({"what": "config 5 (bf16 mode), rocprofv3 --pmc (own pass): matrix-pipe busy cycles and LDS bank conflicts per kernel; "
                       "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)", "kernels": res}, open(out, "w"), indent=1)
    for k, d in sorted(res.items(), key=lambda kv: -(kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))):
        if d.get("SQ_INSTS_MFMA", 0):
            print(f"{k:72s} mfma busy {d['mfma_busy_frac_of_simd_cycles']:.3f}  lds conflict {d['lds_bank_conflict_frac']:.3f}")


if __name__ == "__main__":
    main()
