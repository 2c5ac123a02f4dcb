"""Batch-1 latency of the inference hot loop (bin/room_classification_server:283-287).

  a) the reference's expression on this engine:  model(data).argmax(dim=1).cpu()   under torch.no_grad()
  b) model.predict(data): native eval forward + hmp_argmax_rows + one pinned D2H of the labels

One MP3D-like scene graph per call (a fresh graph object every call, as the server receives a new frame), 306-d objects,
3-layer SAGE hidden 64 and the shipped GAT shape (3 layers, 3 heads, hidden 64, concat False).  Prints one JSON line.
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "hydra-gnn_amd")
from hydra_gnn_amd import workloads  # noqa: E402
from hydra_gnn_amd.models import HeterogeneousNetwork  # noqa: E402

DEV = "cuda:0"


def timed(fn, frames, reps):
    for f in frames[:8]:
        fn(f)
    torch.cuda.synchronize()
    ts = []
    for r in range(reps):
        f = frames[r % len(frames)]
        t0 = time.perf_counter_ns()
        fn(f)
        ts.append((time.perf_counter_ns() - t0) * 1e-3)
    ts = np.array(ts)
    return {"median_us": round(float(np.median(ts)), 1), "p90_us": round(float(np.percentile(ts, 90)), 1)}


def main():
    torch.manual_seed(0)
    frames = [workloads.mp3d_like_batch(1, 100 + i).to(DEV) for i in range(32)]
    out = {"frames": len(frames), "objects_median": int(np.median([f["objects"].x.size(0) for f in frames]))}
    nets = {
        "sage_h64_l3": dict(conv_block="GraphSAGE", hidden_dim=64, num_layers=3),
        "gat_h64x3_l3": dict(conv_block="GAT", GAT_hidden_dims=[64, 64], GAT_heads=[3, 3, 3], GAT_concats=[False, False, False]),
    }
    for name, kw in nets.items():
        net = HeterogeneousNetwork(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, dropout=0.25, **kw).to(DEV).eval()

        def ref_expr(f):
            with torch.no_grad():
                return net(f).argmax(dim=1).cpu()

        a = timed(ref_expr, frames, 400)
        b = timed(net.predict, frames, 400)
        for f in frames[:4]:
            assert torch.equal(ref_expr(f), net.predict(f))
        out[name] = {"forward_argmax_cpu": a, "predict": b}
    # ---- the step before the model: spark_dsg JSON frame -> HeteroData (hydra_gnn_amd/dsg.py), on the reference's test graph
    import os

    from hydra_gnn_amd import dsg

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dsg_x8F5xyUWy9e.json")
    raw = json.load(open(path))
    net6 = HeterogeneousNetwork(input_dim_dict={"objects": 6, "rooms": 6}, output_dim=26, dropout=0.25, **nets["sage_h64_l3"]).to(DEV).eval()
    stages = {}

    def stage(name, fn, reps=50):
        r = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter_ns()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        stages[name] = round((time.perf_counter_ns() - t0) * 1e-3 / reps, 1)
        return r

    sg = stage("parse_static_layers_us", lambda: dsg.load_dsg_json(raw))
    rog = stage("room_object_graph_us", lambda: dsg.RoomObjectGraph(sg))
    oo = stage("object_connectivity_hip_us", lambda: dsg.object_connectivity(rog, 1.5, 2.0, 0.2, DEV))
    data = stage("to_hetero_data_us", lambda: dsg.to_hetero_data(rog, oo, None, DEV))
    stage("predict_us", lambda: net6.predict(data))
    out["dsg_frame_62_objects"] = stages
    print(json.dumps(out))


if __name__ == "__main__":
    main()
