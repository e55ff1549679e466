#!/usr/bin/env python3
"""bench.py — MCTS simulations/sec on the BASELINE.json workload.

One "step" = one batched search: initial inference + S simulations (select,
recurrent inference, expand, backup) for G games on this rank's GPU, with the
visit counts / root values delivered to pinned host memory (the search kernel
stores them there itself: mzs_set_result_sink; --copy-back uses a device->host
copy launch per step instead).  Inputs (observations,
weights, per-game Dirichlet + tie-break streams) are resident in HBM before the
timed region.  Games shard over ranks with no collective in the data path.

  python bench.py [--gpus N] [--steps K] [--warmup W]      (N > 1: starts the N ranks itself)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MATRIX_PEAK_TFLOPS = 157.3  # fp32 MFMA == fp32 vector peak
BF16_MATRIX_PEAK_TFLOPS = 16 * 157.3   # dense bf16 MFMA (guide: ~2.5 PF = 16x the fp32 matrix rate)


def algorithmic_bytes_per_sim(A, S_dim, dbar):
    """select d(24A+8) + expand 8*S_dim+4+16A + backup 20(d+1)."""
    return dbar * (24 * A + 8) + 8 * S_dim + 4 + 16 * A + 20 * (dbar + 1)


def flops_per_sim(A, S_dim, H, L, kind=0):
    if kind == 1:   # lstm_model: four heads, each Linear + L layers of three gate projections
        def head(n_in, out):
            return n_in * H + sum(3 * out * (H if l == 0 else out) for l in range(L))
        return 2 * (2 * head(S_dim + A, S_dim) + head(S_dim, A) + head(S_dim, S_dim))
    return 2 * ((S_dim + A) * H + 2 * H * S_dim + S_dim * H + H * A + H * S_dim) + 2 * H * H * 2 * L


def cpu_baseline(dims, blob, S, pbc, seconds_target=12.0):
    """The oracle (CPU restatement, bit-exact to the GPU path) timed on the host cores:
    a bounded sample of the same workload, 1 thread and up to 16 threads."""
    from muzero_amd import rng
    from oracle import mz_oracle as mzo
    d = mzo.dims(*dims)
    A = dims[1]
    n1 = 2048
    obs = np.random.RandomState(999).uniform(-0.05, 0.05, size=(n1, dims[0])).astype(np.float32)
    dirichlet, noise = rng.noise_streams(100000 + np.arange(n1), A, S, 0.25, True)
    kw = dict(obs=obs, dirichlet=dirichlet, noise=noise, train=True, want=("visits",))
    mzo.search(d, blob, S, 0.997, 0.25, pbc, n_threads=1, **{**kw, "obs": obs[:16],
               "dirichlet": dirichlet[:16], "noise": noise[:16]})
    t0 = time.perf_counter()
    mzo.search(d, blob, S, 0.997, 0.25, pbc, n_threads=1, **kw)
    t1 = time.perf_counter() - t0
    rate1 = n1 * S / t1
    threads = max(1, min(16, os.cpu_count() or 1))
    reps, tm = 0, 0.0
    t0 = time.perf_counter()
    while tm < seconds_target:
        mzo.search(d, blob, S, 0.997, 0.25, pbc, n_threads=threads, **kw)
        reps += 1
        tm = time.perf_counter() - t0
    cpu_model = "?"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": n1 * reps * S / tm, "unit": "sims/s", "cores": threads, "kind": "port",
            "value_1thread": rate1,
            "sample": f"oracle/mz_oracle.c (bit-exact CPU port): {n1} games x {S} sims on 1 thread "
                      f"({t1:.2f} s), {n1 * reps} games x {S} sims on {threads} threads ({tm:.2f} s)",
            "cpu": cpu_model}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--games", type=int, default=4096, help="games per GPU")
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--weights", default="tag123")
    ap.add_argument("--dtype", default="f32", choices=("f32", "bf16", "bf16x3"),
                    help="f32: the bit-exact headline; bf16: the 1e-2 variant (weights / layer inputs in bfloat16, "
                         "v_mfma_f32_4x4x4_16B_bf16), its own roofline against the bf16 matrix peak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--copy-back", action="store_true", help="read results back with a copy launch per step")
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)
    args = ap.parse_args()

    from muzero_amd import shard
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver calls it: this process becomes the launcher of N
        # ranks (one per GPU, RCCL).  It has not touched the GPU and never will.
        sys.exit(shard.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist
    from muzero_amd import rng
    from muzero_amd.engine import SearchEngine

    env_world = shard.rank_world()[1]
    if env_world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks")
    if shard.visible_gpus() < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but only {shard.visible_gpus()} GPU(s) are visible")
    rank, world = shard.init(args.backend)    # nccl == RCCL on ROCm; no-op for one rank
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    w = np.load(os.path.join(ROOT, "tests", "golden", f"weights_{args.weights}.npz"))
    kind = int(w["kind"]) if "kind" in w.files else 0
    dims = tuple(int(w[k]) for k in ("obs_dim", "A", "S_dim", "H", "L")) + (kind, {"f32": 0, "bf16": 1, "bf16x3": 2}[args.dtype])
    obs_dim, A, S_dim, H, L = dims[:5]
    model_name = "LSTM" if kind == 1 else "MLP"
    G, S = args.games, args.sims

    # synthetic inputs of the BASELINE shape
    obs, seeds = shard.shard_inputs(rank, G, obs_dim)
    dirichlet, noise = rng.noise_streams(seeds, A, S, 0.25, True)
    eng = SearchEngine(dims, G, S, discount=0.997, device=local_rank)
    eng.set_weights(w["blob"])
    kernel_name = eng.search_kernel_name(S)          # from the library's own dispatch
    obs_d = torch.as_tensor(obs).to(dev)
    dir_d = torch.as_tensor(dirichlet).to(dev)
    noise_d = torch.as_tensor(noise).to(dev)
    seeds_d = torch.as_tensor(seeds).to(dev)
    pack_h = torch.zeros(G * (A + 1), dtype=torch.int32).pin_memory()   # visits [G,A] | root values [G]
    visits_h = pack_h[: G * A].view(G, A)
    pack_d = eng.results_packed_view()
    if not args.copy_back:
        eng.set_result_sink(visits_h, pack_h[G * A:].view(torch.float32))

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(i=None):
        if i is not None:
            ev0[i].record()
        eng.search(obs_d, dir_d, noise_d, S, True)       # the search kernel(s), current stream
        if i is not None:
            ev1[i].record()
        if args.copy_back:
            pack_h.copy_(pack_d, non_blocking=True)       # one 48 KB read-back: visits + root values
        return eng.results_view()

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = step(i)
    fence()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, dev)

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    # SURVEY.md 8(d) asks for the median over >= 20 batches next to the whole-region rate: start-to-start
    # intervals of consecutive steps on the stream (kernel + the gap to the next launch)
    step_ms = [ev0[i].elapsed_time(ev0[i + 1]) for i in range(args.steps - 1)]
    step_ms_median = float(np.median(step_ms)) if step_ms else kernel_ms
    path_len = res["path_len"].cpu().numpy()
    assert int(visits_h.sum()) == G * S, "search did not complete every simulation"
    dbar = float(path_len.sum()) / (G * S)
    sims_per_step = world * G * S
    value = sims_per_step * args.steps / elapsed

    # the same step with its random streams generated on the device first (a self-play move needs
    # fresh streams: mzs_search_streams), timed separately: never part of `value`
    def step_streams():
        eng.search_streams(obs_d, seeds_d, S, True)

    for _ in range(min(args.warmup, 3)):
        step_streams()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step_streams()
    fence()
    elapsed_streams = shard.max_over_ranks(time.perf_counter() - t0, dev)

    out = {
        "metric": "MCTS sims/sec (whole node), CartPole MLP, 50 sims x 4096 games/GPU",
        "value": value, "unit": "sims/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
        "data": f"synthetic (obs ~ U(-0.05,0.05)^{obs_dim}, {args.weights} fixture weights, per-game "
                "RandomState streams pre-drawn on host and resident in HBM)",
        "config": {"workload": f"{'C2: CartPole-v1 ' if args.weights == 'tag123' else ''}{model_name} obs{obs_dim} A{A} "
                               f"S_dim{S_dim} H{H} L{L}, {G} games/GPU x {S} sims, fused engine (1 launch per "
                               f"search, {'4 games' if 'quad' in kernel_name else '1 game'} per wavefront)",
                   "games_per_gpu": G, "num_simulations": S, "engine": "fused",
                   "parallelism": f"games sharded over {world} GPU(s), no collective",
                   "world_size": dist.get_world_size() if world > 1 else 1,
                   "backend": (dist.get_backend() if world > 1 else None)},
        "value_median": world * G * S / (step_ms_median * 1e-3), "step_ms_median": step_ms_median,
        "value_with_streams": sims_per_step * args.steps / elapsed_streams,
        "ms_per_step_with_streams": 1e3 * elapsed_streams / args.steps,
    }
    if rank == 0:
        bps = algorithmic_bytes_per_sim(A, S_dim, dbar)
        ach = bps * G * S / (kernel_ms * 1e-3) / 1e9
        fps = flops_per_sim(A, S_dim, H, L, kind)
        fl = fps * G * S / (kernel_ms * 1e-3) / 1e12
        # The fused kernel holds both halves of the path: the tree work (HBM-roofline accounting,
        # SURVEY.md 8d bytes) and the MLP on v_mfma_f32_4x4x1 (MFMA-roofline accounting).  The
        # larger fraction is reported as the bound; the other one rides along.  `traffic` = HBM
        # bytes per launch from rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, KiB units of the
        # guide), read from the committed profile of this same command.
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                tj = json.load(f)
            if tj.get("games") == G and tj.get("sims") == S:
                traffic = tj["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        peak = BF16_MATRIX_PEAK_TFLOPS if args.dtype != "f32" else F32_MATRIX_PEAK_TFLOPS
        if args.dtype != "f32":
            traffic = None      # profiles/traffic.json is the fp32 kernel's
        out["roofline"] = {"bound": "mfma", "achieved": fl, "peak": peak,
                           "unit": "TFLOP/s", "frac": fl / peak, "traffic": traffic,
                           "kernel": kernel_name, "kernel_ms": kernel_ms,
                           "flops_per_sim": fps, "mean_path_len": dbar,
                           "hbm_equiv": {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": ach / HBM_PEAK_GBS, "bytes_per_sim": bps},
                           "note": "neither roofline binds at this model size: one wave per SIMD, "
                                   "instruction issue 63 % / LDS and memory waits 31 % of wave cycles "
                                   "(profiles/r3_quad_summary.md, DESIGN.md 4)"}
        if world == 1 and not args.no_cpu_baseline and args.dtype == "f32":
            out["cpu_baseline"] = cpu_baseline(dims[:6], w["blob"], S, eng.pbc)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
