#!/usr/bin/env python3
"""BK-Core benchmark: forward + backward diag-elements/s on MI355X, with HBM roofline and a CPU baseline.

    python bench.py --gpus N --steps K --warmup W

One "step" = one BK-Core forward (G and features) plus one reference-mode backward (grad_he_diag) over a
batch of B = 1024 rows of N = 32768 (BASELINE.json configs[1]) held in HBM.  With N > 1 GPUs (launched by
torch.distributed.run, one rank per GPU) every rank processes its own B rows -- the operator shards by
batch row and needs no collective -- and the reported value is the whole-job rate (weak scaling).

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (the forward scan) by its algorithmic
bytes (28 B per diag-element: read a, b, c; write G and features) over its average launch time measured
with stream events; `cpu_baseline` times the CPU oracle (C port of the reference algorithm, OpenMP) on a
bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

FWD_BYTES = 28.0   # per diag-element: a,b,c reads (3x4) + G write (8) + features write (8)
BWD_BYTES = 20.0   # per diag-element: G read (8) + grad_features read (8) + grad_he_diag write (4)
HBM_PEAK = 8.0e12  # MI355X HBM3E nominal bytes/s


def make_inputs(B, N, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    a = (-2.0 + 0.5 * torch.randn(B, N, generator=g, device=device)).requires_grad_(True)
    b = torch.ones(B, N - 1, device=device)
    c = torch.ones(B, N - 1, device=device)
    gfeat = torch.randn(B, N, 2, generator=g, device=device)
    z = torch.tensor(1j, dtype=torch.complex64, device=device)
    return a, b, c, z, gfeat


def cpu_baseline(N, budget_s=12.0):
    """C port of the reference algorithm (oracle/bk_oracle.c), fwd + reference bwd, all host threads OpenMP gives."""
    import numpy as np
    from oracle import bk_oracle_c as C
    threads = C.num_threads()
    rng = np.random.default_rng(0)

    def run(rows):
        a = (-2.0 + 0.5 * rng.standard_normal((rows, N))).astype(np.float32)
        b = np.ones((rows, N - 1), np.float32)
        gf = rng.standard_normal((rows, N, 2)).astype(np.float32)
        t0 = time.perf_counter()
        _, G = C.forward_f32(a, b, b, 1j)
        C.backward_ref(G, gf, None, 0.5)
        return time.perf_counter() - t0

    rows = 16 * threads
    run(threads)                          # warm-up: page in, spin up the OpenMP team
    total_t, total_rows = 0.0, 0
    while total_t < budget_s:
        total_t += run(rows)
        total_rows += rows
    return {"value": total_rows * N / total_t, "unit": "diag-elems/s", "cores": threads, "kind": "port",
            "sample": f"C/OpenMP oracle, fwd+bwd, {total_rows} rows x N={N} in {total_t:.2f} s"}


def cpu_baseline_torch_loop(N_sub=16384, B=64):
    """Restatement of the reference's own cost model (Python loop over N of batched complex128 torch ops),
    forward only, on an N-subsample; the reference path costs ~constant time per sequence position."""
    from oracle import bk_oracle_torch as O
    g = torch.Generator().manual_seed(0)
    a = -2.0 + 0.5 * torch.randn(B, N_sub, generator=g)
    b = torch.ones(B, N_sub - 1)
    t0 = time.perf_counter()
    O.forward(a, b, b, 1j)
    t = time.perf_counter() - t0
    return {"value": B * N_sub / t, "unit": "diag-elems/s (fwd only)", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"torch complex128 loop (reference cost model), fwd, B={B} x N={N_sub} in {t:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024, help="rows per GPU")
    ap.add_argument("--seq", type=int, default=32768)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(device)

    import bkcore
    from bkcore import BKCoreFunction
    B, N = args.batch, args.seq
    a, b, c, z, gfeat = make_inputs(B, N, device, 1234 + rank)

    def step():
        a.grad = None
        feats, G = BKCoreFunction.apply(a, b, c, z)
        torch.autograd.backward((feats,), (gfeat,))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # per-kernel launch durations with stream events (same stream the kernels are launched on)
    def timed(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    reps = max(5, args.steps)
    ad = a.detach()
    with torch.no_grad():
        G = bkcore.bk_forward(ad, b, c, z)[1]
        t_fwd = timed(lambda: bkcore.bk_forward(ad, b, c, z), reps)
        t_bwd = timed(lambda: bkcore.bk_backward_reference(G, gfeat, None, 0.5), reps)

    if rank == 0:
        # HBM traffic of the dominant kernel from the last committed PMC run (profiles/), if it was this workload
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
            if tj.get("rows") == B and tj.get("seq_len") == N:
                traffic = tj["bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        elems = B * N
        ms = dt / args.steps * 1e3
        value = world * elems / (dt / args.steps)
        fwd_bw = FWD_BYTES * elems / t_fwd
        out = {
            "metric": "bk_core_fwd_bwd_diag_elems_per_sec",
            "value": value,
            "unit": "diag-elems/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BK-Core fwd+bwd B={B} N={N} complex64 per GPU (BASELINE.json configs[1]); "
                                   "he_diag=-2+0.5*randn, h0_super=h0_sub=1, z=1j; reference-mode backward",
                       "rows_per_gpu": B, "seq_len": N, "sharding": f"batch rows x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": fwd_bw / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": fwd_bw / HBM_PEAK, "traffic": traffic,
                         "kernel": "bk_ratio_forward_kernel", "bytes_per_elem": FWD_BYTES, "launch_ms": t_fwd * 1e3,
                         "launch_ms_covers": "one forward call = this kernel + the (empty) second launch, ~5 us"},
            "fwd_ms": t_fwd * 1e3,
            "bwd_ms": t_bwd * 1e3,
            "bwd_GBps": BWD_BYTES * elems / t_bwd / 1e9,
            "hbm_frac_fwd_bwd": (FWD_BYTES + BWD_BYTES) * elems / (t_fwd + t_bwd) / HBM_PEAK,
        }
        if world == 1:
            # what a plain device-to-device copy of the same volume reaches on this box (read + write bytes / time)
            with torch.no_grad():
                src = torch.empty(B, N, 2, device=device)
                dst = torch.empty_like(src)
                t_copy = timed(lambda: dst.copy_(src), reps)
                out["roofline"]["copy_ceiling_GBps"] = 2 * src.numel() * 4 / t_copy / 1e9
                del src, dst
            # secondary measurements (not part of `value`): analytic-mode backward at the same size, and the
            # long-context configuration BASELINE.json configs[2] (B=16, N=2^20: several workgroups per row)
            with torch.no_grad():
                t_ana = timed(lambda: bkcore.bk_backward_analytic(ad, b, c, z, gfeat, None), 5)
                a3, b3, c3, z3, g3 = make_inputs(16, 1 << 20, device, 99)
                a3 = a3.detach()
                G3 = bkcore.bk_forward(a3, b3, c3, z3)[1]
                t3f = timed(lambda: bkcore.bk_forward(a3, b3, c3, z3), reps)
                t3b = timed(lambda: bkcore.bk_backward_reference(G3, g3, None, 0.5), reps)
            # per-iteration stream events over 100 iterations after 10 warm-ups: median and minimum of forward+backward
            # and of the forward alone (the protocol of the reference's own benchmark, scripts/benchmark_bk_triton.py:101-119)
            def per_iter(fn, iters=100, warm=10):
                for _ in range(warm):
                    fn()
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
                for e0, e1 in ev:
                    e0.record()
                    fn()
                    e1.record()
                torch.cuda.synchronize()
                ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
                return ts[len(ts) // 2], ts[0]
            with torch.no_grad():
                fb_med, fb_min = per_iter(lambda: bkcore.bk_backward_reference(
                    bkcore.bk_forward(ad, b, c, z)[1], gfeat, None, 0.5))
                f_med, f_min = per_iter(lambda: bkcore.bk_forward(ad, b, c, z))
            out["events_100_iters"] = {"fwd_bwd_ms_median": fb_med, "fwd_bwd_ms_min": fb_min,
                                       "fwd_ms_median": f_med, "fwd_ms_min": f_min}
            # the layer's chain around the operator (tanh saturation + bfloat16 cast in front, clamp behind, and the
            # whole backward chain), two library launches each way, at the same size
            h0 = torch.full((1, N), -2.0, device=device)
            vpre = torch.randn(B, N, device=device)

            def layer_chain():
                vv = vpre.detach().requires_grad_(True)
                ff, _ = bkcore.FusedBKLayerFunction.apply(vv, h0, b[:1], c[:1], z, 5.0, 5.0)
                torch.autograd.backward((ff,), (gfeat,))
            t_chain = timed(layer_chain, 10)
            del h0, vpre
            out["secondary"] = {
                "layer_chain_fused_fwd_bwd_ms": t_chain * 1e3,
                "analytic_backward_ms": t_ana * 1e3,
                "analytic_backward_note": "grad for he_diag, h0_super, h0_sub; re-runs the forward inside",
                "long_context_B16_N1048576": {"fwd_ms": t3f * 1e3, "bwd_ms": t3b * 1e3,
                                              "diag_elems_per_sec": 16 * (1 << 20) / (t3f + t3b)},
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N)
            out["cpu_baseline_reference_cost_model"] = cpu_baseline_torch_loop()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
