"""bench.py's `roofline.traffic` must follow from the committed counter summaries by hand: every kernel family is an
exact list of kernel names (tools/pmc_families.py), the file is the one of the benchmarked batch size, and the figure is the family's
HBM bytes per step divided by its launch scopes per step."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_families as pf  # noqa: E402

PROF = os.path.join(ROOT, "profiles")


def kernels(name):
    return json.load(open(os.path.join(PROF, name)))["kernels"]


def test_config2_fused_backward_family_is_dx_plus_plain_kernel():
    k = kernels("r02_h_pmc_cfg2.json")
    # 2 x agg_bwd_dx_kernel<16> + 1 x agg_bwd_kernel<16,1> per step, three launch scopes: (2 x 4 948 956 + 4 877 579) / 3
    want = round((2 * 4948956 + 4877579) / 3)
    assert pf.family_traffic(k, "agg_bwd", 3) == want == 4925164
    assert pf.family_traffic(k, "agg_fwd", 3) == round((2 * 6008616 + 728491) / 3)
    assert pf.family_traffic(k, "front", 1) == 17609246
    assert pf.family_traffic(k, "gemm_bwd", 1) == 13770140  # the direct TN kernel only


def test_config5_backward_gemms_exclude_every_forward_kernel():
    k = kernels("r02_e_pmc_cfg5.json")
    fam = {name: pf.kernel_family(name) for name in k}
    fwd = [n for n, f in fam.items() if f == "gemm_fwd"]
    bwd = [n for n, f in fam.items() if f == "gemm_bwd"]
    assert all("ws_kernel" in n or n.rstrip(")").split(">")[0].rstrip().endswith(" 0") for n in fwd), fwd
    assert len([n for n in fwd if "gemm_bf16_ws_kernel" in n]) == 2
    assert all("ws_kernel" not in n for n in bwd)
    forms = sorted(n.split("<")[1].split(">")[0].split(",")[-1].strip() for n in bwd)
    assert forms == ["1", "2"], forms  # input-gradient (NN) and weight-gradient (TN) instantiations
    per_step = pf.family_bytes_per_step(k, "gemm_bwd")
    assert abs(per_step - (2 * 2152506572 + 6838934393)) < 1.0  # 11.1 GB per step over 3 scopes (2 x dX, 1 x dW)
    assert pf.family_traffic(k, "gemm_bwd", 3) == round((2 * 2152506572 + 6838934393) / 3)


def test_every_engine_kernel_of_the_committed_summaries_has_a_family():
    for f in sorted(os.listdir(PROF)):
        if "_pmc_cfg" not in f or not f.endswith(".json") or "mfma" in f or "_l2" in f:
            continue
        for name in kernels(f):
            if "hmp::" not in name:
                continue
            if any(s in name for s in ("collate", "dropout_mask", "argmax", "segment_mean", "plan_small_kernel", "chain_kernel")):
                continue
            assert pf.kernel_family(name) is not None, (f, name)


def test_a_batch_2048_line_does_not_read_the_batch_32_file():
    p32 = pf.pick_pmc_file(PROF, 2, 32, 32)
    p2048 = pf.pick_pmc_file(PROF, 2, 2048, 32)
    assert p32 and "batch" not in os.path.basename(p32)
    assert p2048 and os.path.basename(p2048).endswith("_batch2048.json")
    assert pf.pick_pmc_file(PROF, 2, 77, 32) is None


def test_split_kernel_instances_map_by_their_form_argument():
    """gemm_x3_kernel<tile configuration, ONES, FORM>: the configuration is a template of its own, so FORM is read as the LAST argument"""
    k = kernels("r03_z_pmc_cfg2_batch2048.json")
    fam = {name: pf.kernel_family(name) for name in k if "gemm_x3_kernel" in name}
    assert sorted(fam.values()) == ["gemm_bwd", "gemm_fwd"], fam
    for name, f in fam.items():
        assert name.split(">(")[0].rstrip().endswith("0") == (f == "gemm_fwd"), name
