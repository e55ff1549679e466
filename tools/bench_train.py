#!/usr/bin/env python3
"""Secondary measurement (not the headline metric): wall time of one data-parallel training step
(mzs_train_grad -> all-reduce of the flat gradient bucket -> mzs_train_apply), BASELINE config C1
shape: CartPole MLP, batch 128 per GPU, unroll 5 (and the reference config's unroll 10).

  python tools/bench_train.py [--gpus N]     N > 1: starts N ranks itself (one per GPU, RCCL)

Rank 0 prints one JSON line per case: ms per step, and — for N > 1 — the time of the all-reduce
alone (the flat fp32 bucket of 12 734 floats + the 3K loss scalars), measured in a separate loop
of the same collective calls on the same buffers."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--lstm", action="store_true", help="also time the 'lstm_model' shape (config C4)")
    ap.add_argument("--vision", action="store_true", help="also time the 'vision_model' shape (98 x 98 x 3 images)")
    ap.add_argument("--only-vision", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)
    args = ap.parse_args()

    from muzero_amd import shard
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(shard.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist
    from muzero_amd.engine import SearchEngine
    from muzero_amd.train import Trainer, allreduce_gradient

    if shard.rank_world()[1] != args.gpus:
        sys.exit(f"bench_train.py: --gpus {args.gpus} but WORLD_SIZE={shard.rank_world()[1]}")
    if shard.visible_gpus() < args.gpus:
        sys.exit(f"bench_train.py: --gpus {args.gpus} but only {shard.visible_gpus()} GPU(s) are visible")
    rank, world = shard.init(args.backend)
    local = int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0
    torch.cuda.set_device(local)

    mlp = np.load(os.path.join(ROOT, "tests", "golden", "weights_tag123.npz"))
    lstm = np.load(os.path.join(ROOT, "tests", "golden", "weights_lstmA.npz"))
    cases = [(mlp, 0, 5, 128), (mlp, 0, 10, 128), (mlp, 0, 5, 1024)]
    if args.lstm:     # BASELINE config C4 shape; every head's LSTM runs along the batch axis
        cases += [(lstm, 1, 5, 128), (lstm, 1, 5, 512)]
    if args.vision or args.only_vision:
        vis = np.load(os.path.join(ROOT, "tests", "golden", "weights_visA.npz"))
        cases = ([] if args.only_vision else cases) + [(vis, 2, 5, 32), (vis, 2, 5, 128), (vis, 2, 5, 512)]
    for w, kind, K, B in cases:
        dims = tuple(int(w[k]) for k in ("obs_dim", "A", "S_dim", "H", "L")) + (kind,)
        rs = np.random.RandomState(rank)
        eng = SearchEngine(dims, 1, 0, device=local)
        eng.set_weights(w["blob"])
        tr = Trainer(eng, K, B, "adam", 0.01, "cosineannealinglr", 10000)
        dev = eng.device
        obs_shape = (B, 3, 98, 98) if kind == 2 else (B, dims[0])
        obs = torch.as_tensor(rs.uniform(0 if kind == 2 else -1, 1, obs_shape).astype(np.float32)).to(dev)
        act = torch.zeros((K, B, dims[1]), device=dev)
        act[..., 0] = 1.0
        tv = torch.as_tensor(rs.uniform(0, 50, (K, B)).astype(np.float32)).to(dev)
        trw = torch.ones((K, B), device=dev)
        tp = torch.full((K, B, dims[1]), 1.0 / dims[1], device=dev)
        isr = torch.ones(B, device=dev)

        def fence():
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize(dev)

        for _ in range(5):
            tr.step(obs, act, tv, trw, tp, isr)
        fence()
        n = args.steps
        t0 = time.perf_counter()
        for _ in range(n):
            tr.step(obs, act, tv, trw, tp, isr)
        fence()
        dt = shard.max_over_ranks(time.perf_counter() - t0, dev) / n
        ar_us = None
        if world > 1:
            t0 = time.perf_counter()
            for _ in range(n):
                allreduce_gradient(tr.bucket)
            fence()
            ar_us = 1e6 * shard.max_over_ranks(time.perf_counter() - t0, dev) / n
        shard.assert_replicas_identical(tr.weights(), dev)          # replicas stayed bit-identical
        if rank == 0:
            print(json.dumps({"metric": "training step wall time", "model": ("mlp_model", "lstm_model", "vision_model")[kind],
                              "unroll": K, "batch_per_gpu": B,
                              "n_gpus": world, "global_batch": B * world, "ms_per_step": 1e3 * dt,
                              "samples_per_s": B * world / dt, "allreduce_us": ar_us,
                              "bucket_floats": int(tr.bucket.numel()),
                              "note": "forward+losses+backward+wgrad, one all-reduce of the flat bucket, "
                                      "Adam; inputs resident"}), flush=True)
        eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
