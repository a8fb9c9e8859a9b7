"""The small ResNet-BK LM around the operator: a training step on the GPU, and the data-parallel step (gradient
all-reduce through DistributedDataParallel) on two CPU processes over gloo with the CPU oracle plugged in as the operator."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _OracleOp(torch.autograd.Function):
    """Oracle forward + reference-mode backward on the CPU (test stand-in for the HIP operator)."""
    @staticmethod
    def forward(ctx, a, b, c, z):
        from oracle import bk_oracle_torch as O
        feats, G = O.forward(a.float(), b.float(), c.float(), complex(z))
        ctx.save_for_backward(G)
        return feats, G

    @staticmethod
    def backward(ctx, gf, gG):
        from oracle import bk_oracle_torch as O
        (G,) = ctx.saved_tensors
        return O.backward_reference(G, gf, gG, 0.5), None, None, None


def _tiny_model():
    from bkcore.lm import SmallResNetBKLM
    torch.manual_seed(3)
    m = SmallResNetBKLM(vocab_size=101, d_model=16, n_layers=2, n_seq=24)
    with torch.no_grad():
        for blk in m.blocks:
            blk.bk_layer.v_proj.weight.normal_(0.0, 0.3)
    return m


def _leave(code=0):
    """End a spawned rank here: results are on disk, the process group is gone.  (Interpreter shutdown of a rank that
    has used gloo occasionally aborts in a library thread's destructor -- 'terminate called without an active
    exception' -- after the worker has returned; nothing of the test runs there.)"""
    import sys
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(code)


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bkcore.lm import training_step
        m = _tiny_model()
        m.set_bk_core(_OracleOp.apply)
        ddp = torch.nn.parallel.DistributedDataParallel(m)
        opt = torch.optim.SGD(ddp.parameters(), lr=0.0)          # lr 0: keep the weights, inspect the gradients
        g = torch.Generator().manual_seed(10 + rank)
        tok = torch.randint(0, 101, (2, 24), generator=g)
        tgt = torch.randint(0, 101, (2, 24), generator=g)
        training_step(ddp, opt, tok, tgt, grad_clip=1e9)
        if rank == 0:
            np.save(os.path.join(out, "ddp.npy"), m.blocks[0].bk_layer.v_proj.weight.grad.clone().numpy())
    finally:
        dist.destroy_process_group()
    _leave()


def test_data_parallel_step_averages_gradients(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "ddp.npy")
    # the same two batches through one process: the mean of the per-batch gradients
    from bkcore.lm import training_step
    grads = []
    for rank in range(2):
        m = _tiny_model()
        m.set_bk_core(_OracleOp.apply)
        opt = torch.optim.SGD(m.parameters(), lr=0.0)
        g = torch.Generator().manual_seed(10 + rank)
        tok = torch.randint(0, 101, (2, 24), generator=g)
        tgt = torch.randint(0, 101, (2, 24), generator=g)
        training_step(m, opt, tok, tgt, grad_clip=1e9)
        grads.append(m.blocks[0].bk_layer.v_proj.weight.grad.numpy())
    ref = 0.5 * (grads[0] + grads[1])
    assert np.abs(ref).max() > 0
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-8)


@pytest.mark.gpu
def test_training_steps_on_gpu_reduce_the_loss():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bkcore.lm import SmallResNetBKLM, training_step
    torch.manual_seed(0)
    model = SmallResNetBKLM(vocab_size=1000, d_model=64, n_layers=2, n_seq=128).cuda()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    g = torch.Generator(device="cuda").manual_seed(1)
    tok = torch.randint(0, 1000, (4, 128), device="cuda", generator=g)
    tgt = torch.randint(0, 1000, (4, 128), device="cuda", generator=g)
    losses = [float(training_step(model, opt, tok, tgt)) for _ in range(8)]
    assert all(np.isfinite(losses))
    assert abs(losses[0] - np.log(1000)) < 0.5                   # random init: about ln(vocab)
    assert losses[-1] < losses[0]
    gv = model.blocks[0].bk_layer.v_proj.weight.grad
    assert gv is not None and torch.isfinite(gv).all() and gv.abs().max() > 0


def _tool_worker(rank, world, port, hybrid, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import train_step_c5
    argv = ["--device", "cpu", "--steps", "3", "--warmup", "1", "--batch", "2", "--vocab", "101", "--d-model", "16",
            "--layers", "2", "--n-seq", "24"] + (["--hybrid"] if hybrid else [])
    line = train_step_c5.main(argv, operator=_OracleOp.apply)
    if rank == 0:
        import json
        with open(os.path.join(out, "line.json"), "w") as fh:
            json.dump(line, fh)
    else:
        assert line is None
    _leave()


@pytest.mark.parametrize("hybrid", [False, True])
def test_training_step_tool_control_flow_on_two_cpu_ranks(hybrid, tmp_path):
    """tools/train_step_c5.py (BASELINE configs[4]'s driver: one process per GPU, DistributedDataParallel, barrier-
    bracketed timing, MAX over ranks, one JSON line from rank 0) on two gloo ranks with the oracle as the operator.
    hybrid: the layer of configs/phase7_small.yaml, whose unused parameters DDP must be told about."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    import json
    mp.spawn(_tool_worker, args=(2, port, hybrid, str(tmp_path)), nprocs=2, join=True)
    with open(tmp_path / "line.json") as fh:
        line = json.load(fh)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["hybrid"] == hybrid and line["device"] == "cpu"
    assert line["config"]["parallelism"] == "dp2"
    assert np.isfinite(line["final_loss"]) and abs(line["final_loss"] - np.log(101)) < 1.0
    assert line["value"] > 0 and line["ms_per_step"] > 0
