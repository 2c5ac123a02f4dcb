"""ctypes binding of ``libhydra_mp.so`` (C ABI declared in ``include/hydra_mp.h``).

There is deliberately no fallback: if the shared library is missing, or no gfx950 device is
visible, :func:`load` / :func:`require_device` raise and every model/op built on them fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
# HMP_LIB selects another build of the same ABI (the profiling variant libhydra_mp_kt.so from `make KTIME=1`)
LIB_PATH = os.environ.get("HMP_LIB") or os.path.join(HERE, "libhydra_mp.so")

MAX_NODE_TYPES, MAX_EDGE_TYPES, MAX_LAYERS, MAX_CONVS = 8, 16, 8, 16
N_KCLASS = 14
KCLASS_NAMES = ["plan", "pack", "gemm_fwd", "aggregate_fwd", "loss", "aggregate_bwd", "gemm_bwd", "grad_reduce",
                "adam", "gat_fwd", "gat_bwd", "pool", "front", "chain"]
CONV_SAGE, CONV_GAT = 0, 1
ACT_NONE, ACT_RELU, ACT_ELU = 0, 1, 2


class Plan(C.Structure):
    _fields_ = [
        ("n_src", C.c_int32), ("n_dst", C.c_int32), ("n_edges", C.c_int64),
        ("d_rowptr", C.c_void_p), ("d_col", C.c_void_p), ("d_eid", C.c_void_p),
        ("d_t_rowptr", C.c_void_p), ("d_t_col", C.c_void_p), ("d_t_pos", C.c_void_p),
    ]


class GatArgs(C.Structure):
    _fields_ = [
        ("heads", C.c_int32), ("channels", C.c_int32), ("self_loops", C.c_int32),
        ("edge_dim", C.c_int32), ("dropout_p", C.c_float),
        ("seed", C.c_uint64), ("rng_stream", C.c_uint32), ("rng_step", C.c_uint32),
    ]


class ConvSpec(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("edge_type", C.c_int32), ("src", C.c_int32), ("dst", C.c_int32),
        ("f_out", C.c_int32), ("heads", C.c_int32), ("concat", C.c_int32), ("self_loops", C.c_int32),
        ("edge_dim", C.c_int32), ("fill_mean", C.c_int32), ("shared_lin", C.c_int32), ("active", C.c_int32),
        ("agg_first", C.c_int32), ("att_dropout", C.c_float),
        ("w0", C.c_int64), ("w1", C.c_int64), ("w2", C.c_int64), ("a0", C.c_int64), ("a1", C.c_int64),
        ("a2", C.c_int64), ("b0", C.c_int64),
    ]


class LayerSpec(C.Structure):
    _fields_ = [
        ("n_convs", C.c_int32), ("act", C.c_int32), ("dropout", C.c_float), ("group_mean", C.c_int32),
        ("out_dim", C.c_int32 * MAX_NODE_TYPES),
        ("passthrough", C.c_int32 * MAX_NODE_TYPES),
        ("convs", ConvSpec * MAX_CONVS),
    ]


class NetSpec(C.Structure):
    _fields_ = [
        ("n_node_types", C.c_int32), ("n_edge_types", C.c_int32), ("n_layers", C.c_int32),
        ("in_dim", C.c_int32 * MAX_NODE_TYPES),
        ("edge_src", C.c_int32 * MAX_EDGE_TYPES), ("edge_dst", C.c_int32 * MAX_EDGE_TYPES),
        ("readout_type", C.c_int32), ("pool_edge_type", C.c_int32), ("aux_readout_type", C.c_int32),
        ("n_params", C.c_int64), ("n_active_params", C.c_int64),
        ("layers", LayerSpec * MAX_LAYERS),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32 * MAX_NODE_TYPES),
        ("d_x", C.c_void_p * MAX_NODE_TYPES),
        ("ldx", C.c_int32 * MAX_NODE_TYPES),
        ("n_edges", C.c_int64 * MAX_EDGE_TYPES),
        ("d_edge_index", C.c_void_p * MAX_EDGE_TYPES),
        ("d_edge_attr", C.c_void_p * MAX_EDGE_TYPES),
        ("n_out", C.c_int32),
        ("d_labels", C.c_void_p),
        ("plan_valid", C.c_int32),
        ("d_node_ptr", C.c_void_p * MAX_NODE_TYPES),
        ("n_graphs", C.c_int32),
        ("max_graph_nodes", C.c_int32),
        ("d_edge_ptr", C.c_void_p * MAX_EDGE_TYPES),
    ]


class TrainArgs(C.Structure):
    _fields_ = [
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float),
        ("ignored_label", C.c_int64), ("seed", C.c_uint64), ("training", C.c_int32), ("d_step", C.c_void_p),
    ]


class CollateItem(C.Structure):
    _fields_ = [("d_src", C.c_void_p), ("d_ptr", C.c_void_p), ("src_total", C.c_int64), ("row_bytes", C.c_int64),
                ("slot", C.c_int32), ("slot_src", C.c_int32), ("slot_dst", C.c_int32)]


_STRUCTS = [Plan, GatArgs, ConvSpec, LayerSpec, NetSpec, Batch, TrainArgs]

_VP, _I32, _I64, _F32, _U64, _U32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64, C.c_uint32

# name -> (restype, argtypes).  Kept in the order of include/hydra_mp.h; tests/test_abi.py checks that
# every function the header declares is listed here and exported by the library.
SIGNATURES = {
    "hmp_abi_version": (C.c_int, []),
    "hmp_last_error": (C.c_char_p, []),
    "hmp_sizeof": (C.c_size_t, [C.c_int]),
    "hmp_device_count": (C.c_int, []),
    "hmp_plan_scratch_bytes": (C.c_size_t, [_I64, _I32, _I32]),
    "hmp_plan_build": (C.c_int, [_VP, Plan, _VP, _VP, _VP]),
    "hmp_segment_mean_fwd": (C.c_int, [_VP, _I32, _I32, Plan, _VP, _I32, _VP]),
    "hmp_segment_mean_bwd": (C.c_int, [_VP, _I32, _I32, Plan, _VP, _I32, _VP]),
    "hmp_gemm_f32": (C.c_int, [_VP, _I32, _I32, _VP, _I32, _I32, _VP, _I32, _I32, _I32, _I32, _VP]),
    "hmp_gemm_bf16": (C.c_int, [_VP, _I32, _I32, _VP, _I32, _I32, _VP, _I32, _I32, _I32, _I32, _VP]),
    "hmp_gemm_bf16_a16": (C.c_int, [_VP, _I32, _VP, _I32, _VP, _I32, _I32, _I32, _I32, _I32, _VP]),
    "hmp_net_set_compute": (C.c_int, [_VP, _I32]),
    "hmp_collate_rows": (C.c_int, [_VP, C.c_int64, _VP, _VP, _VP, _I32, C.c_int64, _VP, _VP]),
    "hmp_collate_edges": (C.c_int, [_VP, C.c_int64, _VP, _VP, _VP, _VP, _VP, _I32, C.c_int64, _VP, _VP]),
    "hmp_gat_fwd": (C.c_int, [_VP, _I32, _VP, _I32, _VP, _I32, _VP, _VP, Plan, GatArgs, _VP, _VP, _VP, _I32, _VP]),
    "hmp_gat_bwd": (C.c_int, [_VP, _I32, _VP, _I32, _VP, _I32, _VP, _I32, _VP, _VP, Plan, GatArgs, _VP, _VP, _VP, _VP, _VP,
                              _VP, _I32, _VP, _I32, _VP, _I32, _VP]),
    "hmp_masked_ce": (C.c_int, [_VP, _I32, _I32, _I32, _VP, _I64, _VP, _I32, _VP, _VP]),
    "hmp_argmax_rows": (C.c_int, [_VP, _I32, _I32, _I32, _VP, _VP]),
    "hmp_adam_flat": (C.c_int, [_VP, _VP, _VP, _VP, _I64, _F32, _F32, _F32, _F32, _F32, _I32, _VP, _VP]),
    "hmp_dropout_mask": (C.c_int, [_U64, _U32, _U32, _F32, _I32, _I32, _VP, _VP]),
    "hmp_net_create": (C.c_int, [C.POINTER(NetSpec), C.POINTER(_VP)]),
    "hmp_net_destroy": (None, [_VP]),
    "hmp_net_workspace_bytes": (C.c_size_t, [_VP, C.POINTER(_I32), C.POINTER(_I64)]),
    "hmp_net_bind_workspace": (C.c_int, [_VP, _VP, C.c_size_t, C.POINTER(_I32), C.POINTER(_I64)]),
    "hmp_net_forward": (C.c_int, [_VP, C.POINTER(Batch), _VP, _I32, _U64, _U32, C.POINTER(_VP), C.POINTER(_I32), _VP]),
    "hmp_net_backward": (C.c_int, [_VP, _VP, _I32, _VP, _VP, C.POINTER(_VP), _VP]),
    "hmp_net_aux_output": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_I32), C.POINTER(_I32)]),
    "hmp_net_backward2": (C.c_int, [_VP, _VP, _I32, _VP, _I32, _VP, _VP, C.POINTER(_VP), _VP]),
    "hmp_net_step_fwd_bwd": (C.c_int, [_VP, C.POINTER(Batch), _VP, _VP, C.POINTER(TrainArgs), _VP]),
    "hmp_net_step_adam": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.POINTER(TrainArgs), _VP]),
    "hmp_net_step_fused": (C.c_int, [_VP, C.POINTER(Batch), _VP, _VP, _VP, _VP, C.POINTER(TrainArgs), _VP]),
    "hmp_net_hidden": (C.c_int, [_VP, _I32, _I32, C.POINTER(_VP), C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "hmp_net_read_state": (C.c_int, [_VP, C.POINTER(_I32), C.POINTER(_I32), _VP]),
    "hmp_graph_begin": (C.c_int, [_VP]),
    "hmp_graph_end": (C.c_int, [_VP, C.POINTER(_VP)]),
    "hmp_graph_launch": (C.c_int, [_VP, _VP]),
    "hmp_graph_destroy": (None, [_VP]),
    "hmp_timer_create": (C.c_int, [C.POINTER(_VP)]),
    "hmp_timer_start": (C.c_int, [_VP, _VP]),
    "hmp_timer_stop": (C.c_int, [_VP, _VP]),
    "hmp_timer_elapsed_ms": (C.c_int, [_VP, C.POINTER(_F32)]),
    "hmp_timer_destroy": (None, [_VP]),
    "hmp_net_profile": (C.c_int, [_VP, _I32]),
    "hmp_net_profile_read": (C.c_int, [_VP, C.POINTER(_F32), C.POINTER(_I32)]),
    "hmp_object_edges_count": (C.c_int, [_VP, _VP, _VP, _I32, C.c_double, C.c_double, C.c_double, _VP, _VP, _VP]),
    "hmp_object_edges_fill": (C.c_int, [_VP, _VP, _VP, _I32, C.c_double, C.c_double, C.c_double, _VP, _VP, _I32, _VP]),
    "hmp_gcn_norm": (C.c_int, [Plan, _VP, _VP]),
    "hmp_segment_wsum": (C.c_int, [_VP, _I32, _I32, Plan, _I32, _VP, _VP, _VP, _I32, _VP]),
    "hmp_bias_act_drop_fwd": (C.c_int, [_VP, _I32, _I32, _I32, _VP, _I32, _F32, _U64, _U32, _U32, _VP, _I32, _VP]),
    "hmp_bias_act_drop_bwd": (C.c_int, [_VP, _I32, _VP, _I32, _I32, _I32, _I32, _F32, _VP, _I32, _VP]),
    "hmp_colsum": (C.c_int, [_VP, _I32, _I32, _I32, _VP, _VP]),
    "hmp_rowdot_sum": (C.c_int, [_VP, _I32, _VP, _I32, _I32, _I32, _VP, _VP]),
    "hmp_batchnorm_fwd": (C.c_int, [_VP, _I32, _I32, _I32, _VP, _VP, _VP, _VP, _F32, _F32, _I32, _VP, _I32, _VP, _VP]),
    "hmp_batchnorm_bwd": (C.c_int, [_VP, _I32, _VP, _I32, _I32, _I32, _VP, _VP, _I32, _VP, _I32, _VP, _VP, _VP]),
    "hmp_collator_create": (C.c_int, [_I32, C.POINTER(_VP), _I64, _I32, C.POINTER(CollateItem), C.POINTER(_VP)]),
    "hmp_collator_run": (C.c_int, [_VP, _VP, _I32, C.POINTER(_VP), C.POINTER(_I64), C.POINTER(_I64), _VP, _I32, _VP]),
    "hmp_collator_destroy": (None, [_VP]),
    "hmp_htree_build": (C.c_int, [_I32, _I32, _VP, _I64, _VP, _I64, _VP, _I64, C.POINTER(_VP)]),
    "hmp_htree_sizes": (C.c_int, [_VP, C.POINTER(_I32), C.POINTER(_I64), C.POINTER(_I64)]),
    "hmp_htree_fill": (C.c_int, [_VP, _VP, _VP, C.POINTER(_VP), C.POINTER(_VP)]),
    "hmp_htree_destroy": (None, [_VP]),
    "hmp_gemm_bf16_dx": (C.c_int, [_VP, _I32, _VP, _I32, _I32, _VP, _I32, _VP, _I32, _I32, _I32, _F32, _VP, _I32, _I32, _I32, _I32, _VP]),
    "hmp_gemm_bf16_dw": (C.c_int, [_VP, _I32, _VP, _I32, _I32, _VP, _I32, _I32, _VP, _I32, _I64, _I32, C.POINTER(_I32), _I32, _I32, _I32, _VP]),
    "hmp_comm_unique_id": (C.c_int, [_VP]),
    "hmp_comm_create": (C.c_int, [_VP, _I32, _I32, C.POINTER(_VP)]),
    "hmp_comm_destroy": (None, [_VP]),
    "hmp_comm_query": (C.c_int, [_VP, C.POINTER(_I32), C.POINTER(_I32)]),
    "hmp_comm_allreduce_sum_f32": (C.c_int, [_VP, _VP, _I64, _VP]),
    "hmp_comm_broadcast_f32": (C.c_int, [_VP, _VP, _I64, _I32, _VP]),
}

ABI_VERSION = 3  # the HMP_ABI_VERSION of include/hydra_mp.h this binding was written against (tests/test_abi.py compares them)

_lib: Optional[C.CDLL] = None


class HydraMPError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the HIP library (no GPU needed for this step) and bind every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so.7 / libhsa-runtime64: it must be in the process BEFORE our library is
    # dlopen'ed, so that both resolve to ONE HIP runtime (device pointers and streams are shared between them);
    # loading libhydra_mp.so first would pull /opt/rocm's runtime in and leave torch on a second, blind copy.
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise HydraMPError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C hydra-gnn_amd/csrc`.  hydra_gnn_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.hmp_abi_version() != ABI_VERSION:
        raise HydraMPError(f"ABI version mismatch: library {lib.hmp_abi_version()}, binding {ABI_VERSION} (stale build? run make)")
    for i, st in enumerate(_STRUCTS):
        if lib.hmp_sizeof(i) != C.sizeof(st):
            raise HydraMPError(f"struct layout mismatch for {st.__name__}: C {lib.hmp_sizeof(i)} vs ctypes {C.sizeof(st)}")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().hmp_last_error()
        raise HydraMPError(f"libhydra_mp error {rc}: {msg.decode() if msg else '?'}")


def require_device() -> C.CDLL:
    lib = load()
    if lib.hmp_device_count() < 1:
        raise HydraMPError("no gfx950 (MI355X) device visible: hydra_gnn_amd has no CPU fallback")
    return lib


def stream_ptr() -> int:
    """hipStream_t of torch's current stream (torch is the device-memory / stream plumbing)."""
    import torch

    return int(torch.cuda.current_stream().cuda_stream)
