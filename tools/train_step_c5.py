#!/usr/bin/env python3
"""End-to-end training step of the small ResNet-BK LM with the HIP BK-Core (BASELINE.json configs[4]).

    python tools/train_step_c5.py [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_step_c5.py

One process per GPU; with more than one, DistributedDataParallel over RCCL averages the gradients (the only
communication: the operator itself shards by batch row).  Dummy token batches, random-init weights.  Rank 0 prints one
JSON line with tokens/s over all ranks.  ``--hybrid``: the consumer configs/phase7_small.yaml selects (the resolvent
diagonal gates the block's mixing path; the operator's backward gets a cotangent on G_ii only).  ``--device cpu`` runs
the same control flow on gloo for tests, which must hand ``main`` an operator: the library has no CPU path."""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import torch  # noqa: E402


def main(argv=None, operator=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--fused", action="store_true", help="elementwise steps around the operator inside the library")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole step (forward, backward, clip, AdamW) in one HIP graph and replay it (1 GPU)")
    ap.add_argument("--hybrid", action="store_true", help="G_ii gates the mixing path (use_hybrid_attention)")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"])
    ap.add_argument("--force-dist", action="store_true",
                    help="a single rank also initialises the process group and wraps the model in DDP (rehearses the "
                         "RCCL path on a one-GPU machine; launch through torch.distributed.run)")
    ap.add_argument("--vocab", type=int, default=50257)
    ap.add_argument("--d-model", type=int, default=256)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--n-seq", type=int, default=256)
    args = ap.parse_args(argv)
    on_gpu = args.device == "cuda"
    if not on_gpu and operator is None:
        raise SystemExit("--device cpu needs an operator from the caller: the library computes on the GPU only")
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = torch.device("cuda", local_rank if world > 1 else 0) if on_gpu else torch.device("cpu")
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    if on_gpu:
        torch.cuda.set_device(device)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as dist
        if on_gpu:
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    from bkcore.lm import SmallResNetBKLM, training_step
    torch.manual_seed(0)
    model = SmallResNetBKLM(vocab_size=args.vocab, d_model=args.d_model, n_layers=args.layers, n_seq=args.n_seq,
                            fused=args.fused, hybrid=args.hybrid).to(device)
    if operator is not None:
        model.set_bk_core(operator)
    if dist_on:
        # hybrid: output_proj and bk_scale of each layer take no part in the step (as in the reference layer)
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index] if on_gpu else None,
                                                          find_unused_parameters=args.hybrid)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.01, capturable=bool(args.graph))
    g = torch.Generator(device=device).manual_seed(100 + rank)
    n_seq = args.n_seq
    tokens = torch.randint(0, args.vocab, (args.batch, n_seq), device=device, generator=g)
    targets = torch.randint(0, args.vocab, (args.batch, n_seq), device=device, generator=g)
    step = lambda: training_step(model, opt, tokens, targets)
    if args.graph:
        assert world == 1 and on_gpu, "--graph is a single-GPU mode"
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up outside the capture (allocator, lazy init)
            for _ in range(3):
                training_step(model, opt, tokens, targets)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            static_loss = training_step(model, opt, tokens, targets)

        def step():
            graph.replay()
            return static_loss
    for _ in range(args.warmup):
        loss = step()
    if dist_on:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if dist_on:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    line = None
    if rank == 0:
        line = ({"metric": "resnet_bk_small_lm_train_tokens_per_sec", "value": world * args.batch * n_seq * args.steps / dt,
                          "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "ms_per_step": dt / args.steps * 1e3,
                          "final_loss": float(loss), "fused_layer": bool(args.fused), "hybrid": bool(args.hybrid), "hip_graph": bool(args.graph),
                 "device": args.device, "ddp": bool(dist_on), "config": {"workload": f"phase7_small dims: d_model {args.d_model}, {args.layers} layers, n_seq {n_seq}, "
                          f"batch {args.batch}/GPU, vocab {args.vocab}, AdamW 1e-4, clip 1.0; dummy tokens", "parallelism": f"dp{world}"}})
        print(json.dumps(line), flush=True)
    if dist_on:
        # every rank idle and the DDP wrapper (its reducer holds the process group) gone before the group is torn down
        dist.barrier()
        del step, opt, model
        dist.destroy_process_group()
    return line


if __name__ == "__main__":
    main()
