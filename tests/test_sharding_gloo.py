"""world_size-2 CPU test (gloo) of the batch-row sharding used for multi-GPU runs: the ranks' row blocks tile the
batch, each rank computes only its rows (the CPU oracle stands in for the kernel -- no GPU here), and the gathered
result equals the unsharded one bit for bit.  No collective is involved in the computation itself."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bkcore.sharding import gather_rows, row_shard, shard_rows


def test_row_shard_tiles_the_batch():
    for B in (0, 1, 7, 8, 1024, 1025, 8192):
        for world in (1, 2, 3, 4, 8):
            cover = []
            for r in range(world):
                a, b = row_shard(B, world, r)
                assert 0 <= a <= b <= B
                cover.extend(range(a, b))
            assert cover == list(range(B))
            sizes = [row_shard(B, world, r)[1] - row_shard(B, world, r)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        row_shard(8, 2, 2)


def test_row_units_policy_outside_a_distributed_job():
    from bkcore import bk_core
    assert not dist.is_initialized()
    assert not bk_core._whole_rows_only()
    bk_core.BKCoreFunction.ROW_UNITS = "never"
    try:
        assert bk_core._whole_rows_only()
    finally:
        bk_core.BKCoreFunction.ROW_UNITS = "auto"


def test_broadcast_operand_stays_a_view():
    b = torch.ones(1, 15).expand(8, -1)
    s = shard_rows(b, 2, 1)
    assert s.shape == (4, 15) and s.stride(0) == 0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, N, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import bk_oracle_c as C
        g = torch.Generator().manual_seed(1234)            # same global inputs on every rank
        a = -2.0 + 0.5 * torch.randn(B, N, generator=g)
        b = torch.ones(1, N - 1).expand(B, -1)
        c = torch.ones(1, N - 1).expand(B, -1)
        al, bl, cl = (shard_rows(t, world, rank) for t in (a, b, c))
        _, G_local = C.forward_f32(al.numpy(), bl.contiguous().numpy(), cl.contiguous().numpy(), 1j)
        G = gather_rows(torch.from_numpy(G_local), B)
        dist.barrier()
        # inside a distributed job the operator keeps to whole rows: a collective on another stream may hold the
        # compute units the form with units handing over inside one launch would need all at once
        from bkcore import bk_core
        assert bk_core._whole_rows_only()
        bk_core.BKCoreFunction.ROW_UNITS = "always"
        assert not bk_core._whole_rows_only()
        bk_core.BKCoreFunction.ROW_UNITS = "auto"
        # the timing reduction bench.py uses: the slowest rank defines the step time
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == float(world)
        if rank == 0:
            _, G_full = C.forward_f32(a.numpy(), b.contiguous().numpy(), c.contiguous().numpy(), 1j)
            np.save(os.path.join(out_dir, "ok.npy"), np.array([np.array_equal(G.numpy(), G_full)]))
    finally:
        dist.destroy_process_group()
    # results are on disk and the group is gone: end the rank here rather than in interpreter shutdown, where a rank
    # that has used gloo occasionally aborts in a library thread's destructor
    import sys
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(0)


def test_two_rank_sharded_equals_unsharded(tmp_path):
    world, B, N = 2, 7, 300        # uneven split: 4 + 3 rows
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, N, str(tmp_path)), nprocs=world, join=True)
    assert np.load(tmp_path / "ok.npy")[0]
