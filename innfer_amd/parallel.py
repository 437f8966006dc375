"""Multi-GPU chop_forward: tiles sharded over ranks, HR tiles gathered on rank 0.

The reference is single-process; the only part of its hot path that shards is
Model.chop_forward's tile list (run.py:186-197: every tile's forward is independent,
the blend utils.py:436-443 is the single cross-tile step).  One process per GPU
(torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests):

  * every rank holds the (small) LR frame and cuts ITS contiguous range of whole tile
    rows -- tiles are row-major, so the range is contiguous in the tile list and seams
    between ranks are horizontal bands only;
  * each rank pushes its tiles through the network in batches;
  * one gather of the raw HR tiles to rank 0 (padded to the largest share; RCCL lowers a
    gather to direct peer sends, which suits the fully connected 7-link xGMI topology:
    every peer pushes over its own link), then ONE blend kernel on rank 0;
  * model chains: the blended intermediate is broadcast before the next stage.

No collective sits inside the per-tile compute.
"""
import torch
import torch.distributed as dist


def shard_tile_rows(n_rows, n_cols, world, rank):
    """Contiguous block of whole tile rows for `rank`: (first_tile, n_tiles).
    Rows are dealt as evenly as possible, earlier ranks take the remainder
    (43 rows over 8 ranks -> 6,6,6,5,5,5,5,5)."""
    base, rem = divmod(n_rows, world)
    rows = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first * n_cols, rows * n_cols


class ChopRunner:
    """chop_forward (run.py:167-202) over `world` ranks.

    model_fn   : [n,C,ps,ps] -> [n,C,s*ps,s*ps] (an innfer_amd nn.Module on the GPU)
    extract_fn : (img, (ps,ps), [step,step], batch_first, tile_range) -> tiles, default HIP kernel
    recompose_fn: (tiles, H, W, step, scale) -> image, default HIP kernel
    plan_fn    : (H, W, patch, step) -> (ps, ys, xs), default the C ABI's innfer_chop_plan
    Returns the blended [1,C,sH,sW] tensor on rank 0 and None elsewhere
    (all ranks get it with broadcast_result=True, used between chained models).
    """

    def __init__(self, model_fn, scale, tile_batch=64, patch=200, step=0.5, group=None,
                 extract_fn=None, recompose_fn=None, plan_fn=None):
        self.model_fn, self.scale, self.tile_batch = model_fn, scale, tile_batch
        self.patch, self.step, self.group = patch, step, group
        if extract_fn is None or recompose_fn is None or plan_fn is None:
            from . import lib as L
            from .utils import utils as U
            extract_fn = extract_fn or U.extract_patches_2d
            recompose_fn = recompose_fn or U.recompose_tensor
            plan_fn = plan_fn or L.chop_plan
        self.extract_fn, self.recompose_fn, self.plan_fn = extract_fn, recompose_fn, plan_fn

    def _world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group), dist.get_rank(self.group)
        return 1, 0

    def __call__(self, data, broadcast_result=False):
        world, rank = self._world()
        _, C, H, W = data.shape
        ps, ys, xs = self.plan_fn(H, W, self.patch, self.step)
        n_rows, n_cols = len(ys), len(xs)
        first, count = shard_tile_rows(n_rows, n_cols, world, rank)
        P = ps * self.scale
        outs = []
        if count:
            tiles = self.extract_fn(data, (ps, ps), [self.step, self.step], batch_first=True,
                                    tile_range=(first, count)).squeeze(0)
            with torch.no_grad():
                for i in range(0, count, self.tile_batch):
                    outs.append(self.model_fn(tiles[i:i + self.tile_batch]))
        if world == 1:
            hr = torch.cat(outs, 0) if len(outs) != 1 else outs[0]
            return self.recompose_fn(hr, H, W, step=self.step, scale=self.scale)

        # ---- gather the HR tiles on rank 0 (shares padded to the largest one) ----
        max_count = shard_tile_rows(n_rows, n_cols, world, 0)[1]
        dtype = outs[0].dtype if outs else data.dtype
        out_c = outs[0].shape[1] if outs else getattr(self.model_fn, 'out_nc', C)    # a rank without tiles (more ranks than tile rows)
        send = torch.zeros((max_count, out_c, P, P), dtype=dtype, device=data.device)
        if count:
            send[:count] = torch.cat(outs, 0) if len(outs) != 1 else outs[0]
        del outs
        gathered = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
        dist.gather(send, gathered, dst=0, group=self.group)
        result = None
        if rank == 0:
            parts = [gathered[r][:shard_tile_rows(n_rows, n_cols, world, r)[1]] for r in range(world)]
            hr = torch.cat(parts, 0)
            del gathered, parts
            result = self.recompose_fn(hr, H, W, step=self.step, scale=self.scale)
        if broadcast_result:
            if rank != 0:
                result = torch.empty((1, out_c, H * self.scale, W * self.scale), dtype=dtype, device=data.device)
            dist.broadcast(result, src=0, group=self.group)
        return result


def run_chain(runners, data):
    """Model chain `a+b` (run.py:424-426): every stage is tile-sharded; the blended
    intermediate is broadcast so each rank can cut its own next-stage tiles."""
    x = data
    for i, r in enumerate(runners):
        x = r(x, broadcast_result=(i + 1 < len(runners)))
    return x
