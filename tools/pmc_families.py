"""Kernel name -> bench.py kernel family, and the HBM traffic of a family from a committed rocprofv3 --pmc summary.

bench.py's `roofline.traffic` is the counter-measured HBM bytes (FETCH_SIZE + WRITE_SIZE, gfx950 corrections applied by
tools/pmc_summary.py) of ONE launch scope of the dominant family.  A family is the set of kernels the executor launches inside one
profiling class (`hmp_net_profile`, csrc/net.hip: KC_*), so the mapping below names those kernels EXACTLY -- a substring match
swept the forward GEMM instantiations into the backward family and missed `agg_bwd_kernel` next to `agg_bwd_dx_kernel`.

    family          kernels (csrc/)
    front           front_kernel
    plan            plan_*_kernel
    pack            pack_kernel
    gemm_fwd        gemm_kernel<.., FORM = 0, ..> | gemm_bf16_kernel<.., FORM = 0> | gemm_x3_kernel<., FORM = 0> | gemm_bf16_ws_kernel
    gemm_bwd        gemm_kernel<.., FORM = 1 | 2 | 3, ..> | gemm_bf16_kernel / gemm_x3_kernel<.., FORM = 1 | 2 | 3> | gemm_tn_direct_kernel |
                    gemm_tn_tall_kernel | gemm_bf16_dx_kernel | gemm_bf16_dw_kernel
    agg_fwd         agg_proj_fwd_kernel | agg_fwd_kernel | agg_fwd_win_kernel | agg_fwd_mm_kernel | seg_mean_rows_kernel
    agg_bwd         agg_bwd_dx_kernel | agg_bwd_kernel | agg_bwd_win_kernel | agg_bwd_mm_kernel | seg_mean_rows_t_kernel
    gat_fwd         gat_fwd_kernel
    gat_bwd         gat_bwd1_kernel | gat_bwd2_kernel
    grad_reduce     grad_reduce_kernel | adam_kernel
    loss            masked_ce_kernel | masked_ce_rows_kernel

FORM is the layout template argument of the two tiled GEMM kernels (gemm.hip: 4th argument, gemm_bf16.hip: last argument):
0 = NT (x * W^T, the forward projection), 1 = NN (input gradient), 2 = TN (weight gradient), 3 = run-time mix (backward launches only).
"""
from __future__ import annotations

import glob
import json
import os
import re
from typing import Dict, Optional

_SIMPLE = [
    ("front", r"\bfront_kernel\b"),
    ("plan", r"\bplan_[a-z_]+_kernel\b"),
    ("pack", r"\bpack_kernel\b"),
    ("gemm_fwd", r"\bgemm_bf16_ws_kernel\b"),
    ("gemm_bwd", r"\bgemm_tn_direct_kernel\b|\bgemm_tn_tall_kernel\b|\bgemm_bf16_dx_kernel\b|\bgemm_bf16_dw_kernel\b"),
    ("agg_fwd", r"\bagg_proj_fwd_kernel\b|\bagg_fwd_kernel\b|\bagg_fwd_win_kernel\b|\bagg_fwd_mm_kernel\b|\bseg_mean_rows_kernel\b"),
    ("agg_bwd", r"\bagg_bwd_dx_kernel\b|\bagg_bwd_kernel\b|\bagg_bwd_win_kernel\b|\bagg_bwd_mm_kernel\b|\bseg_mean_rows_t_kernel\b"),
    ("gat_fwd", r"\bgat_fwd_kernel\b"),
    ("gat_bwd", r"\bgat_bwd[12]_kernel\b"),
    ("grad_reduce", r"\bgrad_reduce_kernel\b|\badam_kernel\b"),
    ("loss", r"\bmasked_ce(_rows)?_kernel\b"),
]


def _template_args(name: str, kernel: str):
    m = re.search(r"\b" + kernel + r"<([^>]*)>", name)
    return [a.strip() for a in m.group(1).split(",")] if m else None


def kernel_family(name: str) -> Optional[str]:
    """bench.py family of a (demangled) kernel name, or None for kernels outside the step (torch fills, copies, collation)."""
    a = _template_args(name, "gemm_kernel")
    if a is not None:
        if len(a) < 4:  # first builds of round 1: the layout was a run-time value, one instantiation served both directions
            return "gemm_mixed"
        return "gemm_fwd" if a[3] == "0" else "gemm_bwd"
    a = _template_args(name, "gemm_bf16_kernel")
    if a is not None:
        return "gemm_fwd" if a[-1] == "0" else "gemm_bwd"
    m = re.search(r"\bgemm_x3_kernel<.*,\s*(\d)>\s*\(", name)  # <tile configuration (itself a template), ONES, FORM>
    if m:
        return "gemm_fwd" if m.group(1) == "0" else "gemm_bwd"
    for fam, pat in _SIMPLE:
        if re.search(pat, name):
            return fam
    return None


def steps_of(kernels: Dict[str, dict]) -> int:
    """steps the counter run took = dispatches of the once-per-step gradient un-pack kernel"""
    return max([v["dispatches"] for k, v in kernels.items() if "grad_reduce_kernel" in k] or [0])


def family_bytes_per_step(kernels: Dict[str, dict], family: str) -> float:
    steps = steps_of(kernels)
    if not steps:
        return 0.0
    return sum(v["hbm_bytes_per_dispatch"] * v["dispatches"] for k, v in kernels.items() if kernel_family(k) == family) / steps


def family_traffic(kernels: Dict[str, dict], family: str, scopes_per_step: int) -> Optional[int]:
    """HBM bytes per launch scope of `family`: its kernels' bytes per step / the profiling scopes the family has per step."""
    b = family_bytes_per_step(kernels, family)
    if b <= 0 or scopes_per_step <= 0:
        return None
    return round(b / scopes_per_step)


def pick_pmc_file(profiles_dir: str, config: int, batch: int, default_batch: int, tag: str = "") -> Optional[str]:
    """Latest committed counter summary of THIS workload: `r*_pmc_cfg{config}{tag}.json` for the config's default batch,
    `r*_pmc_cfg{config}{tag}_batch{B}.json` otherwise (a batch-2048 line must not read the batch-32 file)."""
    suffix = "" if batch == default_batch else f"_batch{batch}"
    files = sorted(glob.glob(os.path.join(profiles_dir, f"r*_pmc_cfg{config}{tag}{suffix}.json")))
    return files[-1] if files else None


if __name__ == "__main__":
    import sys

    d = json.load(open(sys.argv[1]))["kernels"]
    steps = steps_of(d)
    fams: Dict[str, float] = {}
    for k, v in d.items():
        f = kernel_family(k) or "(outside the step)"
        fams[f] = fams.get(f, 0.0) + v["hbm_bytes_per_dispatch"] * v["dispatches"] / max(steps, 1)
    for f, b in sorted(fams.items(), key=lambda kv: -kv[1]):
        print(f"{f:22s} {b / 1e6:12.3f} MB / step")
