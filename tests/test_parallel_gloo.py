"""N > 1 path on CPU: two gloo ranks shard the chop tiles by rows, gather on rank 0 and blend.
The per-tile network / extract / blend are the oracle's CPU functions here (the HIP kernels need
a GPU); what is under test is innfer_amd.parallel: sharding, padded gather, ordering, chain
broadcast.  Result must equal the single-process oracle chop_forward bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from oracle import tiles as otiles
from innfer_amd import synth
from innfer_amd.parallel import ChopRunner, run_chain


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _extract(img, patch_shape, step, batch_first=False, tile_range=None):
    t = otiles.extract_patches_2d(img, patch_shape, step, batch_first=True).squeeze(0)
    if tile_range is not None:
        t = t[tile_range[0]:tile_range[0] + tile_range[1]]
    return t.unsqueeze(0)


def _plan(H, W, patch, step):
    return oracle.chop_geometry(H, W, patch, step)


def _worker(rank, world, port, h, w, q, shard='tiles'):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sd1 = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(synth.rrdbnet_shapes(nb=1, scale=1), 1).items()}
        sd2 = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(synth.rrdbnet_shapes(nb=1, scale=2), 2).items()}
        f1 = lambda t: oracle.rrdbnet_forward(sd1, t, nb=1, scale=1)
        f2 = lambda t: oracle.rrdbnet_forward(sd2, t, nb=1, scale=2)
        x = torch.from_numpy(synth.uniform((1, 3, h, w), 5))
        kw = dict(extract_fn=_extract, recompose_fn=otiles.recompose_tensor, plan_fn=_plan, tile_batch=2, shard=shard, profile=True)
        r1 = ChopRunner(f1, 1, **kw)
        r2 = ChopRunner(f2, 2, **kw)
        with torch.no_grad():
            y = r2(x)                                   # single stage
            z = run_chain([r1, r2], x)                  # chain 1x + 2x
        first, count = r2._share(*[len(v) for v in _plan(h, w, 200, 0.5)[1:]], world, rank)
        assert r2.last['tiles'] == count and r2.last['tiles_total'] >= count
        # real tiles only: a sender moves exactly its share, rank 0 receives everybody else's
        tile_bytes = 3 * (400 ** 2) * 4
        want = (r2.last['tiles_total'] - count if rank == 0 else count) * tile_bytes
        assert r2.last['exchange_bytes'] == want, (r2.last, want)
        if rank == 0:
            q.put((y.numpy(), z.numpy()))
        else:
            assert y is None and z is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


# 250x330: 2 tile rows x 3 columns.  shard='rows' with 3 ranks leaves the last one without tiles; shard='tiles' with 4 ranks
# gives 2,2,1,1 (ranges that split a tile row); 7 ranks: one rank has nothing to send
@pytest.mark.parametrize("h,w,world,shard", [(250, 330, 2, 'tiles'), (250, 330, 3, 'rows'), (250, 330, 4, 'tiles'), (250, 330, 7, 'tiles')])
def test_sharded_chop_equals_single_process(h, w, world, shard):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, h, w, q, shard)) for r in range(world)]
    for p in procs:
        p.start()
    y, z = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sd1 = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(synth.rrdbnet_shapes(nb=1, scale=1), 1).items()}
    sd2 = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(synth.rrdbnet_shapes(nb=1, scale=2), 2).items()}
    x = torch.from_numpy(synth.uniform((1, 3, h, w), 5))
    torch.set_num_threads(2)
    with torch.no_grad():
        ref_y = oracle.chop_forward(lambda t: oracle.rrdbnet_forward(sd2, t, nb=1, scale=2), x, 2)
        mid = oracle.chop_forward(lambda t: oracle.rrdbnet_forward(sd1, t, nb=1, scale=1), x, 1)
        ref_z = oracle.chop_forward(lambda t: oracle.rrdbnet_forward(sd2, t, nb=1, scale=2), mid, 2)
    # tiles are independent and the blend order is fixed, but oneDNN may pick different kernels for
    # batch-2 vs batch-1 convs: allow fp32 round-off, nothing more
    np.testing.assert_allclose(y, ref_y.numpy(), atol=2e-6, rtol=0)
    np.testing.assert_allclose(z, ref_z.numpy(), atol=4e-6, rtol=0)
