#!/usr/bin/env python3
"""Headline benchmark: scene-graphs/sec of the full training step (plan + fwd + masked CE + bwd + grad
all-reduce + Adam) of Hydra-GNN's room classifier on an MP3D-like hetero batch, on N MI355X of one node.

    python bench.py [--gpus N --steps K --warmup W] [--config 2|3|4|5] [--batch B] [--no-graph] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Default workload (the one BASELINE.json's metric is quoted on) = configs[1]: MP3D repartitioned-rooms-like HeteroData
(objects 306-d, rooms 6-d, 4 edge types), 3-layer HeteroConv(SAGE) hidden 64, batch 32, fp32, dropout 0.25 (SURVEY.md
8(d) config 2).  --config 3 (GAT 4x128, B=64), 4 (H-tree SAGE h128 4 layers, 16 graphs/rank) and 5 (one 1M-object graph
per rank, hidden 256, fp32) are the other BASELINE configs; they are parity-test / diagnosis workloads, not the bench line.
Weak scaling: every rank steps its own batch; ONE RCCL all-reduce of the flat gradient (+loss,count) per step makes the
update the count-weighted global-batch update.

Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` (dominant kernel family, HIP-event timed on
the executor's streams) and `cpu_baseline` (the oracle = op-for-op PyG restatement, on the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "hydra-gnn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6.29 TB/s is the measured copy rate
MFMA_F32_PEAK_TF = 157.3   # dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16), MI355X_MICROARCH.md

DEFAULT_BATCH = {2: 32, 3: 64, 4: 16, 5: 1}
HT_DIMS = {"object": 306, "room": 6, "object-room": 6, "room-room": 6, "object_virtual": 306, "room_virtual": 6}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5])
    ap.add_argument("--batch", type=int, default=None, help="graphs per rank")
    ap.add_argument("--big-objects", type=int, default=1_000_000, help="config 5: objects per graph")
    ap.add_argument("--graph", action="store_true", help="hipGraph replay instead of eager launches (eager is faster on ROCm 7.2: "
                    "graph replay adds ~3 us per node, the eager chain runs back to back)")
    ap.add_argument("--no-graph", action="store_true", help="(default) eager launches")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default=None,
                    help="compute mode of the large GEMMs (default: bf16 for config 5 as BASELINE.json names it, fp32 otherwise)")
    ap.add_argument("--force-collective", action="store_true",
                    help="1 GPU only: run the data-parallel launch structure (un-pack, RCCL all-reduce in a 1-rank group, Adam) "
                         "to price the collective path without a second GPU; diagnosis, not the bench line")
    ap.add_argument("--edge-order", choices=["grouped", "random"], default="grouped",
                    help="config 5: 'grouped' = edge lists as the generator writes them, destination by destination (the order a "
                         "scene-graph builder produces); 'random' = every edge list permuted (worst case of the plan's "
                         "wave-aggregated counters; results are identical, the plan restores the stable order)")
    ap.add_argument("--collective", choices=["rccl", "torch"], default="rccl",
                    help="N > 1 (or --force-collective): 'rccl' = the library enqueues ncclAllReduce on the executor's own stream "
                         "(hmp_comm_*); 'torch' = torch.distributed all_reduce (ProcessGroupNCCL: own stream + event hand-offs)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="'gloo' (diagnosis): ranks may SHARE a GPU (RCCL refuses two ranks on one device) and the gradient all-reduce "
                         "goes through torch/gloo; rehearses the N-rank engine path on a 1-GPU box, never a bench line")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="launcher / rendezvous / timing-protocol rehearsal WITHOUT a GPU: gloo ranks, the step is the flat "
                         "all-reduce of a config-2-sized gradient buffer only; prints the JSON line with value null")
    ap.add_argument("--min-timed-s", type=float, default=0.2,
                    help="the K-step timed region is repeated (each repeat bracketed by barrier + synchronize) until this much "
                         "time has been timed; `steps` stays K, `timed_regions` says how many")
    ap.add_argument("--fresh-batches", action="store_true",
                    help="extra leg (config 2 / 3): every step takes a DIFFERENT batch, collated on the device from a resident "
                         "dataset of min(64 x batch, max(2048, 4 x batch)) graphs (store.BatchStream: one host call + one kernel) and described afresh -- "
                         "the loop a trainer actually runs; reported as `fresh_batches` beside the headline")
    ap.add_argument("--gat-edge", action="store_true",
                    help="config 3: the GAT_edge variant (SURVEY 8(d): inputs after compute_relative_pos -- objects 303-d, rooms 3-d, "
                         "edge_attr [E, 3] -- conv_block='GAT_edge'); reported separately from the config-3 line")
    ap.add_argument("--real-fixture", action="store_true",
                    help="config 2: the reference's real scene graph (tests/golden/dsg_x8F5xyUWy9e.json through the DSG reader: 5 rooms, "
                         "62 objects, 356 + 2 + 62 + 62 directed edges; 300-d semantic block from a seeded per-label table) "
                         "replicated x batch instead of the synthetic generator (SURVEY 8(d) config 2, second input)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    a = ap.parse_args()
    if a.batch is None:
        a.batch = DEFAULT_BATCH[a.config]
    if a.precision is None:
        a.precision = "bf16" if a.config == 5 else "fp32"
    if a.steps is None:
        a.steps = 200 if a.config != 5 else 10
    if a.warmup is None:
        a.warmup = 20 if a.config != 5 else 3
    return a


def make_workload(args, rank):
    """(model constructor kwargs, model class names, cpu batch, label accessor, description)"""
    from hydra_gnn_amd import workloads

    if args.config == 2:
        kw = dict(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=64,
                  num_layers=3, dropout=0.25)
        if args.real_fixture:
            return kw, "HeterogeneousNetwork", workloads.real_fixture_batch(args.batch), "rooms", (
                f"BASELINE configs[1], second input (SURVEY 8(d)): the reference's real MP3D scene graph x8F5xyUWy9e (5 rooms, 62 objects, "
                f"482 directed edges) replicated x{args.batch}, 3-layer HeteroConv(SAGE) hidden 64, dropout 0.25")
        return kw, "HeterogeneousNetwork", workloads.config2_batch(args.batch, rank=rank), "rooms", (
            "BASELINE configs[1]: MP3D-like HeteroData (objects 306-d, rooms 6-d, 4 edge types), 3-layer HeteroConv(SAGE) "
            "hidden 64, dropout 0.25")
    if args.config == 3:
        kw = dict(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, conv_block="GAT", GAT_hidden_dims=[128, 128],
                  GAT_heads=[4, 4, 4], GAT_concats=[True, True, False], dropout=0.25)
        if args.gat_edge:
            kw.update(input_dim_dict={"objects": 303, "rooms": 3}, conv_block="GAT_edge")
            return kw, "HeterogeneousNetwork", workloads.config3_batch(args.batch, rank=rank, edge=True), "rooms", (
                "BASELINE configs[2], GAT_edge variant (SURVEY 8(d)): inputs after compute_relative_pos (objects 303-d, rooms 3-d, "
                "edge_attr [E, 3]), 3-layer HeteroConv(GAT_edge, 4 heads) hidden 128, dropout 0.25")
        return kw, "HeterogeneousNetwork", workloads.config3_batch(args.batch, rank=rank), "rooms", (
            "BASELINE configs[2]: MP3D-like HeteroData, 3-layer HeteroConv(GAT, 4 heads) hidden 128, dropout 0.25")
    if args.config == 4:
        kw = dict(input_dim_dict=HT_DIMS, output_dim=26, conv_block="GraphSAGE", hidden_dim=128, num_layers=4,
                  disable_initialization=True, dropout=0.25)
        return kw, "HeterogeneousNeuralTreeNetwork", workloads.htree_batch(args.batch, seed=workloads.BASE_SEED + 4 + 1000 * rank), \
            "room_virtual", ("BASELINE configs[3]: MP3D Neural-Tree H-tree decomposition (fixture topologies from the reference's "
                             "junction-tree code), 4-layer HeteroConv(SAGE) hidden 128 + LeafPool, dropout 0.25")
    kw = dict(input_dim_dict={"objects": 256, "rooms": 256}, output_dim=26, conv_block="GraphSAGE", hidden_dim=256, num_layers=3,
              dropout=0.25)
    g = workloads.big_hetero_graph(n_obj=args.big_objects, n_rooms=max(args.big_objects // 100, 1), seed=workloads.BASE_SEED + 5 + rank)
    if args.edge_order == "random":
        gen = torch.Generator().manual_seed(1234 + rank)
        for et in g.edge_types:
            ei = g[et].edge_index
            g[et].edge_index = ei[:, torch.randperm(ei.size(1), generator=gen)].contiguous()
    return kw, "HeterogeneousNetwork", g, "rooms", (
        f"BASELINE configs[4]: ONE synthetic hetero graph per rank, {args.big_objects} objects, {max(args.big_objects // 100, 1)} rooms, "
        f"in-degree 16, 3-layer HeteroConv(SAGE) hidden 256; projections in {args.precision} MFMA with fp32 accumulation; "
        + ("fp32 storage" if args.precision == "fp32" else
           "projected rows Z, input gradients G, dZ and the hidden activations H (read only by those GEMMs) stored as bf16, features / logits / parameters / optimiser state fp32"))


# ------------------------------------------------------------------------------------------------------------
# algorithmic (compulsory) traffic / flops of one training step, per kernel family
# ------------------------------------------------------------------------------------------------------------
def step_costs(net, holder, fused, root_in_place=False, bf16_storage=False):
    """Algorithmic (compulsory) bytes / flops of ONE step per kernel class, from the shapes of this batch (formulas: DESIGN.md
    section 5), at the TRUE element size of every tensor: fp32 = 4 bytes everywhere except, with
    `bf16_storage` (config 5 in bf16 mode), the tensors the engine stores as bf16 -- the projected rows Z and their gradient
    dZ of a 256-wide layer, the hidden activations H and the input gradients G (2 bytes); features, weights, logits, the last
    layer's narrow Z / dZ and the split-K slabs stay fp32.  `fused` = the small-batch launch sequence ran (front kernel,
    projections / input gradients inside the aggregation kernels): the GEMM terms are then charged to the kernels that
    execute them."""
    from hydra_gnn_amd._lib import CONV_GAT

    nn_ = dict(zip(net.node_types, holder.n_nodes))
    ne = dict(zip(net.edge_types, holder.n_edges))
    dims = [dict(net.in_dims)]
    for layer in net.layers:
        dims.append(dict(layer.out_dims))
    al4 = lambda v: (v + 3) // 4 * 4
    keys = ("front", "plan", "pack", "gemm_fwd", "agg_fwd", "agg_bwd", "gemm_bwd", "gat_fwd", "gat_bwd", "grad_reduce")
    cost = {k: {"bytes": 0.0, "flops": 0.0, "launches": 0} for k in keys}

    def add(k, by, fl):
        cost[k]["bytes"] += by
        cost[k]["flops"] += fl

    L = len(net.layers)

    def wide(l):  # layer l's outputs are 256 wide: its Z / dZ / output H / the gradient G of that output are bf16-stored
        return bf16_storage and all(al4(w) == 256 for w in net.layers[l].out_dims.values() if w)

    zs = [2.0 if wide(l) else 4.0 for l in range(L)]                           # Z[l], dZ[l]
    hin = [4.0] + [2.0 if wide(l - 1) else 4.0 for l in range(1, L)]           # H[l] as read by layer l (l = 0: the features)
    hout = [2.0 if (wide(l) and l < L - 1) else 4.0 for l in range(L)]         # H[l + 1] as written by layer l
    gin = [2.0 if (wide(l) and l < L - 1) else 4.0 for l in range(L)]          # gradient of layer l's output as read by its backward
    n_params = float(net.n_active)
    E_all = float(sum(holder.n_edges))
    N_all = float(sum(holder.n_nodes))
    plan_b = 16.0 * E_all + 24.0 * E_all + 8.0 * N_all
    pack_b = 8.0 * n_params
    add("front" if fused else "plan", plan_b, 0.0)
    add("front" if fused else "pack", pack_b, 0.0)
    cost["front" if fused else "plan"]["launches"] += 1
    if not fused:
        cost["pack"]["launches"] += 1
    for l, layer in enumerate(net.layers):
        live = [c for c in layer.convs if c.active]
        gat = live[0].kind == CONV_GAT
        ncols = {t: 0 for t in net.node_types}
        dsts = {c.edge_type[2] for c in live}
        if gat:
            for c in live:
                H, Cc = c.gat["heads"], c.f_out
                ncols[c.edge_type[0]] += H * al4(Cc) + al4(H)
                ncols[c.edge_type[2]] += al4(H)
        else:
            for c in live:
                ncols[c.edge_type[0]] += al4(c.f_out)
            for t in dsts:
                ncols[t] += al4(layer.out_dims[t])
        # forward projection Z_s = H_s * Wp_s^T: front kernel (layer 0), previous layer's aggregation kernel (fused), else GEMM
        proj_cls = "gemm_fwd"
        if fused and not gat:
            proj_cls = "front" if l == 0 else "agg_fwd"
        elif fused and gat and l > 0 and net.layers[l - 1].convs[0].kind != CONV_GAT:
            proj_cls = "agg_fwd"
        for s, nc in ncols.items():
            if nc == 0:
                continue
            N, F = nn_[s], dims[l][s]
            add(proj_cls, (hin[l] * N * F if proj_cls != "agg_fwd" else 0) + 4.0 * nc * F + zs[l] * N * nc, 2.0 * N * F * nc)
        if proj_cls == "gemm_fwd":
            cost["gemm_fwd"]["launches"] += 1
        if gat:
            by = fl = 0.0
            for c in live:
                s, _, t = c.edge_type
                H, Cc = c.gat["heads"], c.f_out
                E = ne[c.edge_type] + (nn_[t] if c.gat.get("self_loops") else 0)
                by += 4.0 * (nn_[t] + 1) + 4.0 * E + (12.0 * E if c.gat.get("edge_dim") else 0.0) + 4.0 * nn_[s] * (H * Cc + H) + 4.0 * nn_[t] * H
                fl += E * H * (2.0 * Cc + 12.0)
            by += sum(4.0 * nn_[t] * layer.out_dims[t] for t in dsts)
            add("gat_fwd", by, fl)
            cost["gat_fwd"]["launches"] += 1
            add("gat_bwd", 2.0 * by + sum(16.0 * 8 * ne[c.edge_type] for c in live), 2.0 * fl)
            cost["gat_bwd"]["launches"] += 2
        else:
            # fused aggregation: indices + each projected source segment once + root + out
            for t in dsts:
                Fo = al4(layer.out_dims[t])
                b = (zs[l] + hout[l]) * nn_[t] * Fo  # root read + out write
                for c in live:
                    if c.edge_type[2] != t:
                        continue
                    b += 4.0 * (nn_[t] + 1) + 4.0 * ne[c.edge_type] + zs[l] * nn_[c.edge_type[0]] * Fo
                add("agg_fwd", b, sum(ne[c.edge_type] * Fo for c in live if c.edge_type[2] == t))
            cost["agg_fwd"]["launches"] += 1
            # backward: transposed aggregation (same compulsory traffic as forward, mirrored)
            add("agg_bwd", sum(4.0 * (nn_[c.edge_type[0]] + 1) + 8.0 * ne[c.edge_type] + gin[l] * nn_[c.edge_type[2]] * al4(c.f_out)
                               + zs[l] * nn_[c.edge_type[0]] * al4(c.f_out) for c in live)
                # root block of dZ = copy of the output gradient; bf16 mode at 10^6 rows: read in place by the GEMMs, no copy
                + (0.0 if (root_in_place and l < L - 1 and all(al4(layer.out_dims[t]) == 256 for t in dsts))
                   else sum((gin[l] + zs[l]) * nn_[t] * al4(layer.out_dims[t]) for t in dsts)),
                sum(2.0 * ne[c.edge_type] * al4(c.f_out) for c in live))
            cost["agg_bwd"]["launches"] += 1
        # weight gradient dWp = dZ^T [H | 1]  and (l > 0) input gradient dH = dZ * Wp
        dx_cls = "agg_bwd" if (fused and not gat) else "gemm_bwd"
        for s, nc in ncols.items():
            if nc == 0:
                continue
            N, F = nn_[s], dims[l][s]
            add("gemm_bwd", zs[l] * N * nc + hin[l] * N * F + 4.0 * nc * (F + 1), 2.0 * N * (F + 1) * nc)
            if l > 0:
                # input gradient: dZ (again), the weights, the activation mask (H[l]) and the gradient G[l] it writes
                add(dx_cls, (zs[l] * N * nc if dx_cls == "gemm_bwd" else 0) + 4.0 * nc * F + hin[l] * N * F + gin[l - 1] * N * F, 2.0 * N * F * nc)
        if l > 0 and dx_cls == "gemm_bwd":
            cost["gemm_bwd"]["launches"] += 1
    cost["gemm_bwd"]["launches"] += 1 if fused else L  # weight gradients: one merged launch when fused
    # gradient un-pack (reads the split-K slabs once) + Adam (p, g, m, v read; p, m, v written)
    add("grad_reduce", 4.0 * n_params * 2 + 28.0 * n_params, 12.0 * n_params)
    cost["grad_reduce"]["launches"] += 1
    return cost


_JSON_FD = None


def emit_json(obj) -> None:
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def free_port():
    import socket

    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher around it: this process becomes the parent of N ranks (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as torch.distributed.run would set them), relays rank 0's
    JSON line and returns the worst exit code.  The parent never touches the GPU and never execs."""
    import subprocess

    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    # rank 0's stdout is drained by a thread so that the parent can WATCH all children: the first one that exits non-zero (a
    # peer that died before or during the rendezvous would otherwise leave the others waiting out the store timeout, and the
    # driver's SCALE run with them) ends the run -- the others get `fail_grace_s` to notice, then are killed (exact PIDs of
    # children this process started; nothing is re-executed)
    import threading

    chunks = []
    drain = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    drain.start()
    fail_grace_s = float(os.environ.get("HMP_BENCH_FAIL_GRACE_S", "10"))
    rc, failed_at = 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad and failed_at is None:
            failed_at, rc = time.time(), bad[0]
            print(f"bench.py: rank {codes.index(bad[0])} exited with code {bad[0]}; stopping the other ranks", file=sys.stderr)
        if all(c is not None for c in codes):
            break
        if failed_at is not None and time.time() - failed_at > fail_grace_s:
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            break
        time.sleep(0.05)
    drain.join(timeout=5.0)
    for p in procs:
        rc = rc or (p.returncode or 0)
    if rc == 0:
        sys.stdout.write(b"".join(c for c in chunks if c).decode())
        sys.stdout.flush()
    return rc


def rehearse_cpu(args, rank, world):
    """No GPU: the launcher, the rendezvous, the barrier + MAX-over-ranks timing protocol and ONE flat all-reduce per step
    (gloo) of a buffer of config 2's gradient size.  Not a measurement of the hot path: value is null."""
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    die = os.environ.get("HMP_BENCH_TEST_DIE_RANK")  # tests/test_bench_launcher.py: a rank that dies AFTER the rendezvous
    if die is not None:
        dist.barrier()
        if int(die) == rank:
            os._exit(3)
    buf = torch.zeros(126568 + 2)
    for _ in range(args.warmup):
        dist.all_reduce(buf)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        buf.fill_(1.0)
        dist.all_reduce(buf)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = bool((buf == float(world)).all())
    if rank == 0:
        emit_json({"metric": "scene-graphs/sec (fwd+bwd) MP3D hetero batch", "value": None, "unit": "graphs/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * float(t) / args.steps, 5),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "rehearsal": "cpu/gloo: launcher + rendezvous + timing protocol + flat all-reduce only, no hot path",
                          "allreduce_ok": ok, "config": {"workload": "none (rehearsal)", "parallelism": f"dp{world}"}})
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


def main():
    args = parse()
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))  # before anything touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # stdout carries ONE JSON line: whatever native libraries (gloo, RCCL, MIOpen) print to fd 1 goes to stderr instead
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: reporting n_gpus = {world}", file=sys.stderr)
    if args.rehearse_cpu:
        sys.exit(rehearse_cpu(args, rank, world))
    import torch.distributed as dist

    n_dev = torch.cuda.device_count()
    if args.dist_backend == "gloo":
        local_rank = local_rank % max(n_dev, 1)  # diagnosis: ranks may share a device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    elif args.force_collective and args.collective == "torch":
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1, device_id=dev)

    import hydra_gnn_amd.models as hmodels
    from hydra_gnn_amd import _lib, parallel

    model_kw, cls_name, batch_cpu, label_type, workload = make_workload(args, rank)
    torch.manual_seed(1234 + rank)  # ranks start DIFFERENT on purpose: TrainStep broadcasts rank 0's weights
    import contextlib

    with contextlib.redirect_stdout(sys.stderr):  # the H-tree constructors print like the reference's; stdout carries ONE JSON line
        net = getattr(hmodels, cls_name)(**model_kw).to(dev)
    net.train()
    net.native().set_compute(args.precision)
    batch = batch_cpu.to(dev)
    labels = batch[label_type].y
    comm = None
    use_native_comm = args.collective == "rccl" and args.dist_backend == "nccl" and (world > 1 or args.force_collective)
    if use_native_comm:
        try:
            comm = parallel.NativeComm.from_process_group(dev) if world > 1 else parallel.NativeComm.single()
            ok = torch.ones(1, device=dev)
        except Exception as e:  # RCCL could not be bound by the library on this box: the same collective through torch
            print(f"bench.py[rank {rank}]: hmp_comm unavailable ({e}); falling back to torch.distributed all_reduce", file=sys.stderr)
            comm = None
            ok = torch.zeros(1, device=dev)
        if world > 1:  # all ranks must take the same path
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) == 0.0 and comm is not None:
                comm.close()
                comm = None
    rccl_ranks = None
    if comm is not None:
        rccl_ranks, rccl_rank = comm.query()
        assert rccl_ranks == max(world, 1) and rccl_rank == rank, (rccl_ranks, rccl_rank, world, rank)
    elif world > 1 and args.dist_backend == "nccl":
        rccl_ranks = dist.get_world_size()  # torch's ProcessGroupNCCL (RCCL) carries the collective
    step = net.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, seed=20250225, use_graph=args.graph,
                          process_group=True if ((world > 1 or args.force_collective) and comm is None) else None,
                          force_collective=args.force_collective, comm=comm)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed_region():
        """EXACTLY args.steps steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(batch, labels)
        sync_all()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for _ in range(args.warmup):
        step(batch, labels)
    elapsed = timed_region()
    regions = [elapsed]
    # a 2 ms timed region (20 steps of 0.1 ms) is within the noise of one scheduler tick: repeat the K-step region until
    # min_timed_s has been timed (every rank derives the same count from the MAX-reduced first region)
    n_more = min(5000, int(np.ceil(args.min_timed_s / max(elapsed, 1e-6))) - 1) if elapsed < args.min_timed_s else 0
    for _ in range(n_more):
        regions.append(timed_region())
    elapsed = float(sum(regions))
    timed_steps = args.steps * len(regions)
    # host cost of a step = time to ENQUEUE a few steps from an idle stream (no queue back-pressure); untimed extra steps
    host_us, extra_steps = 1e9, 0
    for _ in range(3):  # best of 3: the first launches after an idle period pay a wake-up that is not host work
        sync_all()
        th = time.perf_counter()
        for _ in range(10):
            step(batch, labels)
        host_us = min(host_us, 1e5 * (time.perf_counter() - th))
        extra_steps += 10
    sync_all()
    final_loss = step.loss()
    nstep, status = net.native().read_state()
    assert status == 0, f"engine status bits {status}"
    assert nstep == args.warmup + timed_steps + extra_steps, (nstep, args.warmup, timed_steps, extra_steps)
    ms_per_step = 1e3 * elapsed / timed_steps
    n_graphs = getattr(batch_cpu, "num_graphs", args.batch)
    value = n_graphs * world * timed_steps / elapsed

    nat = net.native()
    out = {
        "metric": "scene-graphs/sec (fwd+bwd) MP3D hetero batch",
        "value": round(value, 1),
        "unit": "graphs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5),
        "timed_regions": len(regions),
        "timed_steps_total": timed_steps,
        "ms_per_step_minmax": [round(1e3 * min(regions) / args.steps, 5), round(1e3 * max(regions) / args.steps, 5)],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "bf16 (MFMA operands, the gathered intermediates Z / G / dZ and the hidden activations; fp32 accumulation, logits, parameters)",
        "data": "synthetic",
        "config": {
            "workload": workload + ("; edge lists randomly permuted" if args.config == 5 and args.edge_order == "random" else "")
            + "; training step = CSR/CSC plan + fwd + masked CE + bwd + flat-grad all-reduce + Adam",
            "graphs_per_rank": n_graphs,
            "global_batch": n_graphs * world,
            "nodes_per_rank": dict(zip(nat.node_types, step._holder.n_nodes if step._holder else [])),
            "edges_per_rank": int(sum(step._holder.n_edges)) if step._holder else None,
            "parallelism": f"dp{world}",
            "rank": rank,
            # what RCCL itself reports for the communicator the gradient all-reduce ran on (ncclCommCount); None: no RCCL communicator
            "rccl_ranks": rccl_ranks,
            "collective": (None if (world == 1 and not args.force_collective) else
                           ("ncclAllReduce enqueued by libhydra_mp on the executor's stream (hmp_comm_*)" if comm is not None else
                            f"torch.distributed all_reduce ({args.dist_backend})")),
            "launch": "eager, one stream" if not args.graph else ("hipGraph replay, 1 graph/step" if world == 1 else
                                                                   "hipGraph replay, 2 graphs/step around the all-reduce"),
            "final_loss": round(final_loss, 5),
            "host_enqueue_us_per_step": round(host_us, 1),
        },
    }

    # ---- fresh-batch leg: the loop a trainer runs -- a new batch every step (base_training_job.py:202-216) --------------------
    if args.fresh_batches and args.config in (2, 3) and world == 1:
        from hydra_gnn_amd import workloads
        from hydra_gnn_amd.store import GraphStore

        n_pool = min(64 * n_graphs, max(2048, 4 * n_graphs))  # 2048 graphs at the default batch, 4 batches' worth at 2048
        rng_p = np.random.Generator(np.random.PCG64(workloads.BASE_SEED + 77))
        pool = [workloads.mp3d_like_graph(rng_p) for _ in range(n_pool)]
        store = GraphStore(pool, dev)
        fstep = net.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, seed=20250225, use_graph=False)
        stream = store.stream(net, n_graphs, label_type)
        perm = np.random.Generator(np.random.PCG64(5)).permutation(n_pool).astype(np.int32)
        batches = [perm[i * n_graphs:(i + 1) * n_graphs] for i in range(n_pool // n_graphs)]
        nb = len(batches)
        for i in range(args.warmup):
            fstep.run(stream.next(batches[i % nb]))
        f_steps = max(args.steps, 200)
        f_regions = []
        for _ in range(5):
            sync_all()
            t0 = time.perf_counter()
            for i in range(f_steps):
                fstep.run(stream.next(batches[i % nb]))
            sync_all()
            f_regions.append(time.perf_counter() - t0)
        sync_all()
        th = time.perf_counter()
        for i in range(50):
            fstep.run(stream.next(batches[i % nb]))
        f_host_us = 1e6 * (time.perf_counter() - th) / 50
        sync_all()
        f_ms = 1e3 * min(f_regions) / f_steps
        out["fresh_batches"] = {
            "ms_per_step": round(f_ms, 5), "graphs_per_s": round(n_graphs / (f_ms * 1e-3), 1),
            "vs_resident_batch": round(f_ms / ms_per_step, 3), "host_enqueue_us_per_step": round(f_host_us, 1),
            "steps": f_steps, "regions": 5, "pool_graphs": n_pool,
            "what": "every step: device collation of a new batch of the resident dataset (1 kernel) + descriptor + the training step",
        }
        assert net.native().read_state()[1] == 0
        stream.close()

    # ---- roofline leg: HIP events around every launch of each kernel family, eager launches, executor streams ----------
    if rank == 0 and not args.no_roofline:
        import ctypes as C

        prof_step = net.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, seed=20250225, use_graph=False)
        prof_steps = min(args.steps, 50)
        for _ in range(min(5, args.warmup)):
            prof_step(batch, labels)
        torch.cuda.synchronize(dev)
        _lib.check(nat._lib.hmp_net_profile(nat._handle, 1))
        for _ in range(prof_steps):
            prof_step(batch, labels)
        torch.cuda.synchronize(dev)
        ms = (C.c_float * _lib.N_KCLASS)()
        ln = (C.c_int32 * _lib.N_KCLASS)()
        _lib.check(nat._lib.hmp_net_profile_read(nat._handle, ms, ln))
        _lib.check(nat._lib.hmp_net_profile(nat._handle, 0))
        per = {name: (ms[i], ln[i]) for i, name in enumerate(_lib.KCLASS_NAMES)}
        fused = per["front"][1] > 0
        # A scope = event, launch(es), event: its elapsed time is the kernel's duration plus the dispatch / event hand-over
        # around it.  The launches of a step run back to back on one stream, so that per-scope extra is what the scopes of a
        # step add up to beyond the step's own (un-profiled) time; it is subtracted so that the per-launch figure is the
        # kernel's duration as rocprofv3 reports it (begin -> end) -- profiles/README.md compares the two.
        scopes_per_step = sum(v[1] for v in per.values()) / float(prof_steps)
        scopes_ms = sum(v[0] for v in per.values()) / float(prof_steps)
        event_overhead_us = max(0.0, 1e3 * (scopes_ms - out["ms_per_step"]) / max(scopes_per_step, 1.0))
        b16s = args.config == 5 and args.precision == "bf16" and os.environ.get("HMP_Z16") != "0"
        cost = step_costs(nat, prof_step._holder, fused, root_in_place=(b16s and os.environ.get("HMP_ROOTCOPY") != "1"), bf16_storage=b16s)
        fam = {
            "front": ("front_kernel (layer-0 projection tiles + plan parts + pack blocks, one launch)", *per["front"]),
            "gemm_fwd": ("gemm_kernel (fp32 MFMA 32x32x2, grouped, LDS-staged)", *per["gemm_fwd"]),
            "gemm_bwd": ("gemm_tn_direct_kernel (weight gradients, fp32 MFMA, register-direct)" if fused else
                         "gemm_kernel (fp32 MFMA 32x32x2, grouped, LDS-staged)", *per["gemm_bwd"]),
            "agg_fwd": ("agg_proj_fwd_kernel / agg_fwd_kernel (SAGE aggregation + next projection / + masked CE)" if fused else
                        "agg_fwd_kernel (fused SAGE aggregation)", *per["aggregate_fwd"]),
            "agg_bwd": ("agg_bwd_dx_kernel / agg_bwd_kernel (transposed aggregation + input-gradient GEMM)" if fused else
                        "agg_bwd_kernel (transposed aggregation)", *per["aggregate_bwd"]),
            "gat_fwd": ("gat_fwd_kernel (edge softmax + aggregation)", *per["gat_fwd"]),
            "gat_bwd": ("gat_bwd1/2_kernel", *per["gat_bwd"]),
            "grad_reduce": ("grad_reduce_kernel (slab reduction + Adam)", per["grad_reduce"][0] + per["adam"][0], per["grad_reduce"][1]),
        }
        table = {}
        for k, (name, tot_ms, launches) in fam.items():
            if launches == 0 or cost[k]["bytes"] == 0:
                continue
            # a profiling scope may cover several launches (GAT backward = 2 kernels; > 8 GEMM problems = 2 launches)
            avg_us = max(1e3 * tot_ms / launches - event_overhead_us, 1e-3)
            by = cost[k]["bytes"] * prof_steps / launches
            fl = cost[k]["flops"] * prof_steps / launches
            table[k] = {"kernel": name, "avg_us": round(avg_us, 3), "scopes_per_step": launches // prof_steps,
                        "alg_bytes_per_launch": round(by), "alg_flops_per_launch": round(fl),
                        "GBps": round(by / (avg_us * 1e-6) / 1e9, 2), "TFLOPs": round(fl / (avg_us * 1e-6) / 1e12, 3)}
        # fp32 GEMM launches of >= 1e9 multiply-adds run on the bf16 matrix pipe by the exact three-way split (csrc/gemm_x3.hip); the
        # figures stay ALGORITHMIC fp32 flops against the fp32 matrix peak -- the kernel issues six bf16 products per algorithmic one
        if not (args.precision == "bf16" and args.config == 5) and os.environ.get("HMP_GEMM_X3") != "0":
            for k in ("gemm_fwd", "gemm_bwd"):
                if k in table and table[k]["alg_flops_per_launch"] >= 2e9 and not (k == "gemm_bwd" and args.config == 2):
                    table[k]["kernel"] = ("gemm_x3_kernel (fp32 operands split exactly into three bf16 pieces, six v_mfma_f32_32x32x16_bf16 products "
                                          "per element product, fp32 accumulate: fp32 accuracy; flops counted once)")
        out["kernel_ms_per_step"] = {n: round(v[0] / prof_steps, 5) for n, v in per.items() if v[1]}
        out["event_overhead_us"] = round(event_overhead_us, 3)
        if table:
            dom = max(table, key=lambda k: table[k]["avg_us"] * table[k]["scopes_per_step"])
            d = table[dom]
            big_bf16 = args.precision == "bf16" and args.config == 5 and dom.startswith("gemm")
            mfma_peak = MFMA_BF16_PEAK_TF if big_bf16 else MFMA_F32_PEAK_TF
            if args.precision == "bf16" and args.config == 5:
                for k in ("gemm_fwd", "gemm_bwd"):
                    if k in table:
                        table[k]["kernel"] = ("gemm_bf16_kernel (v_mfma_f32_32x32x16_bf16, fp32 operands rounded on the way into LDS, "
                                              "fp32 accumulate)")
            t_hbm = d["alg_bytes_per_launch"] / (HBM_PEAK_GBS * 1e9)
            t_mfma = d["alg_flops_per_launch"] / (mfma_peak * 1e12)
            if dom.startswith("gemm") and t_mfma >= t_hbm:
                out["roofline"] = {"kernel": d["kernel"], "bound": "mfma", "achieved": d["TFLOPs"], "peak": mfma_peak,
                                   "unit": "TFLOP/s", "frac": round(d["TFLOPs"] / mfma_peak, 5), "traffic": None,
                                   "avg_launch_us": d["avg_us"]}
            else:
                out["roofline"] = {"kernel": d["kernel"], "bound": "hbm", "achieved": d["GBps"], "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": round(d["GBps"] / HBM_PEAK_GBS, 5), "traffic": None,
                                   "avg_launch_us": d["avg_us"]}
            out["roofline"]["note"] = ("launch/latency-bound regime: the whole step moves ~%.0f MB; every kernel is a chain of ~2 us "
                                       "dependent memory round trips (cold L2 after each kernel boundary), see DESIGN.md section 6"
                                       % (sum(c["bytes"] for c in cost.values()) / 1e6)) if args.config != 5 else (
                "bandwidth regime" if args.precision != "bf16" else
                "bandwidth regime; algorithmic bytes at the true element sizes (bf16 Z / dZ / H / G of the 256-wide layers, "
                "fp32 features, weights, logits)")
            out["roofline_all"] = table
            # HBM traffic of the dominant family from the committed PMC passes of this same command (rocprofv3 --pmc
            # FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 corrections applied by tools/pmc_summary.py)
            # (tools/pmc_families.py: exact kernel names per family, the file of THIS batch size, bytes of the family per step --
            # a launch scope may cover several kernels -- divided by the family's scopes per step)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import pmc_families

            tag = "_gat_edge" if (args.config == 3 and args.gat_edge) else ("_real_fixture" if (args.config == 2 and args.real_fixture) else "")
            pmc = pmc_families.pick_pmc_file(os.path.join(ROOT, "profiles"), args.config, args.batch, DEFAULT_BATCH[args.config], tag)
            if pmc and not (args.config == 5 and args.big_objects != 1_000_000):
                kern = json.load(open(pmc))["kernels"]
                tr = pmc_families.family_traffic(kern, dom, d["scopes_per_step"])
                if tr is not None:
                    out["roofline"]["traffic"] = tr
                    out["roofline"]["traffic_source"] = os.path.relpath(pmc, ROOT)

    # ---- CPU baseline: the oracle (op-for-op PyG restatement) on the host cores, bounded sample -------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # N = 1 only (the contract): other ranks would idle at the barrier
        from oracle import models as omodels

        # the GPU box gives one-GPU jobs a 16-CPU share; more intra-op threads than that only oversubscribes
        cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        torch.set_num_threads(cores)
        torch.manual_seed(1234)
        with contextlib.redirect_stdout(sys.stderr):
            ora = getattr(omodels, cls_name)(**model_kw)
        ora.train()
        opt = torch.optim.Adam(ora.parameters(), lr=0.002, weight_decay=0.001)
        cpu_scale = 1.0
        if args.config == 5:
            # bounded sample: the same generator at 1/20 of the objects (the oracle materialises [E, F] gathers: 16 GB at
            # full size); graphs/s is scaled by the node ratio
            from hydra_gnn_amd import workloads

            small = max(args.big_objects // 20, 1000)
            batch_cpu = workloads.big_hetero_graph(n_obj=small, n_rooms=max(small // 100, 1), seed=workloads.BASE_SEED + 5)
            cpu_scale = small / float(args.big_objects)
        y = batch_cpu[label_type].y
        mask = y != 25

        def cpu_step():
            opt.zero_grad()
            loss = ora.loss(ora(batch_cpu), y, mask)
            loss.backward()
            opt.step()

        budget = 20.0 if args.config != 5 else 60.0
        t_begin = time.perf_counter()
        cpu_step()  # warm-up (thread pools, allocator)
        times = []
        for _ in range(4):
            if time.perf_counter() - t_begin > budget / 4:
                break
            cpu_step()
        while len(times) < 30 and (not times or time.perf_counter() - t_begin < budget):
            t1 = time.perf_counter()
            cpu_step()
            times.append(time.perf_counter() - t1)
        times.sort()
        med = times[len(times) // 2]
        out["cpu_baseline"] = {
            "value": round(n_graphs / med * cpu_scale, 4), "unit": "graphs/s", "cores": cores, "kind": "port",
            "sample": f"median of {len(times)} full training steps ("
                      + (f"same {n_graphs}-graph batch" if cpu_scale == 1.0 else f"1/{round(1 / cpu_scale)}-size graph, rate scaled by the node ratio")
                      + ", fp32, dropout 0.25, torch.optim.Adam) of oracle/ = op-for-op torch restatement of the PyG path; "
                        "torch_geometric itself is not installable offline",
        }

    if rank == 0:
        emit_json(out)
    if world > 1:
        dist.barrier()
    if comm is not None:
        comm.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
