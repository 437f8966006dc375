"""bench.py's N > 1 entry without a launcher (`python bench.py --gpus N`): the parent starts N fresh rank processes before
anything touches a GPU, the ranks rendezvous, time the contract's region (barriers, max over ranks) and rank 0 prints ONE
JSON line.  INNFER_BENCH_SELFTEST=1 replaces the HIP workload by a sleep so that this control flow runs on the CPU box;
the same flow with the real kernels is tests/test_gpu_sharded.py::test_bench_two_ranks_dry_run."""
import json
import os
import subprocess
import sys

import pytest

from innfer_amd.parallel import shard_tile_rows, shard_tiles

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=300):
    env = dict(os.environ, INNFER_BENCH_SELFTEST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_gpus_flag_starts_n_ranks_and_prints_one_line(n):
    r = _run(["--gpus", str(n), "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == n and line["ranks_seen"] == n
    assert line["steps"] == 3 and line["warmup"] == 1
    assert line["ms_per_step"] >= 2.0                    # the 2 ms sleep of every step is inside the timed region


def test_launcher_env_wins_over_the_flag():
    """Under torch.distributed.run the ranks already exist: bench.py must not spawn again (world size from the environment)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, INNFER_BENCH_SELFTEST="1", WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1000:] for o in outs]
    assert json.loads(outs[0][0].strip().splitlines()[-1])["n_gpus"] == 2
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]      # only rank 0 prints the line (gloo itself chats on stdout)


def test_a_failing_rank_fails_the_run():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "nonsense"])
    assert r.returncode != 0


def test_shard_tiles_is_an_even_contiguous_partition():
    for n, world in [(798, 8), (3268, 8), (6, 4), (3, 7), (0, 2), (798, 1)]:
        shares = [shard_tiles(n, world, r) for r in range(world)]
        assert shares[0][0] == 0 and sum(c for _, c in shares) == n
        for (f0, c0), (f1, _) in zip(shares, shares[1:]):
            assert f1 == f0 + c0
        counts = [c for _, c in shares]
        assert max(counts) - min(counts) <= 1 and counts == sorted(counts, reverse=True)
    assert [shard_tiles(798, 8, r)[1] for r in range(8)] == [100] * 6 + [99] * 2
    assert [shard_tile_rows(43, 76, 8, r)[1] // 76 for r in range(8)] == [6, 6, 6, 5, 5, 5, 5, 5]
    import innfer_amd.lib as L                                   # the C ABI's partition (innfer_shard_tiles, host code) is the same function
    for n, world in [(798, 8), (3268, 8), (5, 7), (0, 3)]:
        assert [L.shard_tiles(n, world, r) for r in range(world)] == [shard_tiles(n, world, r) for r in range(world)]


def test_tile_batches():
    from innfer_amd.parallel import MAX_TILE_BATCH, tile_batches
    assert tile_batches(798) == [266, 266, 266]
    assert tile_batches(100) == [100] and tile_batches(0) == []
    for n in (1, 63, 272, 273, 3268, 5000):
        b = tile_batches(n)
        assert sum(b) == n and max(b) <= MAX_TILE_BATCH and max(b) - min(b) <= 1 and len(b) == -(-n // MAX_TILE_BATCH)
    assert tile_batches(10, 4) == [4, 4, 2] and tile_batches(8, 64) == [8]


def test_tile_batches_are_sized_from_the_engine_and_survive_oom():
    """the chop batch follows the engine's workspace, not a constant.  engine_tile_cap takes the largest batch whose cost fits the
    budget (a 16x network: 2.9 GB per tile -> far fewer than 272 per launch); run_tile_batches halves the batch when the allocator runs out and
    still returns every tile's result in order."""
    import torch
    from innfer_amd.parallel import MAX_TILE_BATCH, engine_tile_cap, run_tile_batches, tile_batches

    class Fake:
        def __init__(self, per_tile, limit=None):
            self.per_tile, self.limit, self.calls, self.released = per_tile, limit, [], 0

        def tile_batch_bytes(self, b, ps, dtype):
            return 1000 + b * self.per_tile

        def release_workspace(self):
            self.released += 1

        def __call__(self, x):
            self.calls.append(x.shape[0])
            if self.limit and x.shape[0] > self.limit:
                raise torch.OutOfMemoryError("fake")
            return x * 2.0

    assert engine_tile_cap(Fake(10), 200, torch.float16, "cpu", budget=1e9) == MAX_TILE_BATCH
    assert engine_tile_cap(Fake(100), 200, torch.float16, "cpu", budget=1000 + 100 * 57 + 50) == 57
    assert engine_tile_cap(Fake(100), 200, torch.float16, "cpu", budget=5) == 1            # never zero: a single tile is tried (and may raise)
    assert engine_tile_cap(lambda t: t, 200, torch.float16, "cpu", budget=5) == MAX_TILE_BATCH      # a bare callable: nothing known about it
    assert tile_batches(100, None, 57) == [50, 50] and tile_batches(100, None, 1000) == [100] and tile_batches(7, 3, 2) == [3, 3, 1]
    tiles = torch.arange(23 * 2, dtype=torch.float32).reshape(23, 2, 1, 1)
    # (cap from the engine is MAX_TILE_BATCH here: budget unknown on the CPU -> patch the probe)
    import innfer_amd.parallel as P
    real = P.free_device_bytes
    P.free_device_bytes = lambda device: 1 << 40
    try:
        m = Fake(10, limit=5)
        y = run_tile_batches(m, tiles)
        assert torch.equal(y, tiles * 2.0) and m.released >= 1
        assert m.calls[0] == 23 and max(c for c in m.calls[-4:]) <= 5 and sum(c for c in m.calls if c <= 5) == 23
        got = {}
        run_tile_batches(Fake(10), tiles, tile_batch=4, sink=lambda i, t: got.__setitem__(i, t.clone()))
        assert sorted(got) == [0, 4, 8, 12, 16, 20] and torch.equal(torch.cat([got[i] for i in sorted(got)]), tiles * 2.0)
        with __import__("pytest").raises(torch.OutOfMemoryError):
            run_tile_batches(Fake(10, limit=0.5), tiles)          # even one tile does not fit: the error surfaces
    finally:
        P.free_device_bytes = real


def test_run_tile_batches_lands_results_in_the_tile_buffer():
    """with out= every batch's result belongs in its rows of ONE preallocated buffer -- a model_fn that takes `out=` (the nn.Module
    shells) writes there itself (no result tensor, no copy), any other callable's result is copied; both equal the concatenated form."""
    import torch
    from innfer_amd.parallel import run_tile_batches

    class Lands:
        _accepts_out = True
        calls = []

        def __call__(self, t, out=None):
            assert out is not None and out.shape[0] == t.shape[0] and out.is_contiguous()
            self.calls.append(out.data_ptr())
            out.copy_(t * 2)
            return out

    tiles = torch.arange(7 * 3 * 4 * 4, dtype=torch.float32).reshape(7, 3, 4, 4)
    ref = run_tile_batches(lambda t: t * 2, tiles, tile_batch=3)
    buf = torch.full((7, 3, 4, 4), -1.0)
    m = Lands()
    got = run_tile_batches(m, tiles, tile_batch=3, out=buf)
    assert got is buf and torch.equal(buf, ref)
    assert m.calls == [buf[0:3].data_ptr(), buf[3:6].data_ptr(), buf[6:7].data_ptr()]          # written in place, batch by batch
    buf2 = torch.full((7, 3, 4, 4), -1.0)
    assert run_tile_batches(lambda t: t * 2, tiles, tile_batch=3, out=buf2) is buf2 and torch.equal(buf2, ref)
    buf3 = torch.full((7, 3, 4, 4), -1.0)             # pick (a tuple-returning model): copied, even when the callable could land
    assert torch.equal(run_tile_batches(lambda t: (t, t * 2), tiles, tile_batch=2, pick=lambda y: y[1], out=buf3), ref)
    with __import__("pytest").raises(RuntimeError):
        run_tile_batches(lambda t: t[:, :2] * 2, tiles, tile_batch=3, out=torch.empty(7, 3, 4, 4))
    with pytest.raises(ValueError):          # sink= and out= are alternatives (out= used to win silently)
        run_tile_batches(lambda t: t * 2, tiles, tile_batch=3, sink=lambda i, y: None, out=torch.empty_like(ref))
