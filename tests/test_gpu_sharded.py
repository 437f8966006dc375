"""BASELINE config 4 on the HIP path: the tile-sharded chop / model-chain runner (innfer_amd.parallel) with the real kernels.

  * ChopRunner at world size 1 == Model(chop=True) bit for bit, and against golden G4 (the reference's Model.__call__);
  * run_chain([1x, 4x]) against the oracle's two chop_forwards (run.py:424-426 feeds one model's output to the next);
  * HIP extract_patches_2d(tile_range=...) == the slice of the full extraction (utils/utils.py:318-369);
  * R = 2, 3, 5 ranks emulated in one process (every rank's share computed separately, shares concatenated, one blend)
    == the unsharded result, bit for bit;
  * two real rank processes (gloo rendezvous, both on cuda:0 -- the box has one GPU) through ChopRunner / run_chain with the
    HIP kernels == the single-process result, bit for bit;
  * `python bench.py --gpus 2` (INNFER_BENCH_DRYRUN=1) starts two ranks and prints ONE line with n_gpus == 2 and the tile_sharded object.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _sd(shapes, seed=0):
    from innfer_amd import synth
    return {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed).items()}


def _net(dev, nb, scale, seed=0):
    from innfer_amd import synth
    from innfer_amd.architectures.RRDBNet_arch import RRDBNet
    sd = _sd(synth.rrdbnet_shapes(nb=nb, scale=scale), seed)
    net = RRDBNet(3, 3, 64, nb, upscale=scale)
    net.load_state_dict(sd, strict=True)
    return net.to(dev).eval(), sd


def test_chop_runner_world1_equals_model_and_golden(dev, golden, tmp_path):
    from innfer_amd import synth
    from innfer_amd.parallel import ChopRunner
    from innfer_amd.run import Model
    g = golden("g4_chop")
    for (nb, scale, h, w, tag) in [(2, 4, 250, 330, "x4_250x330"), (1, 2, 201, 640, "x2_201x640")]:
        sd = _sd(synth.rrdbnet_shapes(nb=nb, scale=scale))
        path = str(tmp_path / f"{scale}x_{tag}.pth")
        torch.save(sd, path)
        m = Model(path, arch="infer", scale=None, device="cuda", chop=True, tile_batch=4)
        x = torch.from_numpy(synth.uniform((1, 3, h, w), 40 + scale)).to(dev).half()
        y_model = m(x)
        for shard in ("tiles", "rows"):
            for tb in (1, 4, 64):
                y = ChopRunner(m.model, scale=scale, tile_batch=tb, shard=shard)(x)
                assert torch.equal(y, y_model), (tag, shard, tb)
        y = y_model.float().cpu()
        assert np.abs(y[0, :, ::8, ::8].numpy() - g[f"chop_{tag}_sub"]).max() < 1e-2
        assert np.abs(y[0, :, -32:, -32:].numpy() - g[f"chop_{tag}_crop_b"]).max() < 1e-2


def test_run_chain_vs_oracle(dev):
    """Model chain 1x + 4x (the `-m a+b` form of run.py:424-426) on a 230x310 frame: 2x3 tiles per stage."""
    import oracle
    from innfer_amd import synth
    from innfer_amd.parallel import ChopRunner, run_chain
    net1, sd1 = _net(dev, 2, 1, seed=3)
    net4, sd4 = _net(dev, 2, 4, seed=4)
    x = torch.from_numpy(synth.uniform((1, 3, 230, 310), 77))
    y = run_chain([ChopRunner(net1, 1, tile_batch=4), ChopRunner(net4, 4, tile_batch=4)], x.to(dev).half()).float().cpu()
    with torch.no_grad():
        mid = oracle.chop_forward(lambda t: oracle.rrdbnet_forward(sd1, t, nb=2, scale=1), x, 1)
        ref = oracle.chop_forward(lambda t: oracle.rrdbnet_forward(sd4, t, nb=2, scale=4), mid, 4)
    assert tuple(y.shape) == tuple(ref.shape) == (1, 3, 920, 1240)
    err = (y - ref).abs()
    assert err.max().item() < 1e-2 and err.mean().item() < 1e-3, (err.max().item(), err.mean().item())
    # >= 99 % of the final uint8 codes within +-1 of the fp32 path's
    from innfer_amd.utils import utils as U
    a = U.tensor2np(y.to(dev)).astype(np.int32)
    b = U.tensor2np(ref.to(dev)).astype(np.int32)
    assert (np.abs(a - b) <= 1).mean() >= 0.99


def test_extract_tile_range_is_a_slice(dev):
    from innfer_amd import synth
    from innfer_amd.utils import utils as U
    for h, w in [(250, 330), (431, 615), (200, 200)]:
        for dt in (torch.float16, torch.float32):
            img = torch.from_numpy(synth.uniform((1, 3, h, w), 9)).to(dev).to(dt)
            full = U.extract_patches_2d(img, (200, 200), [0.5, 0.5], batch_first=True).squeeze(0)
            n = full.shape[0]
            for b, c in [(0, n), (0, 1), (n - 1, 1), (1, n - 2), (n // 2, n - n // 2), (n, 0)]:
                if c < 0 or (c == 0 and b == n and n == 1):
                    continue
                part = U.extract_patches_2d(img, (200, 200), [0.5, 0.5], batch_first=True, tile_range=(b, c)).squeeze(0)
                assert torch.equal(part, full[b:b + c]), (h, w, b, c)
            with pytest.raises(ValueError):
                U.extract_patches_2d(img, (200, 200), [0.5, 0.5], batch_first=True, tile_range=(n - 1, 2))


@pytest.mark.parametrize("world", [2, 3, 5])
def test_emulated_ranks_equal_unsharded(dev, world):
    """Every rank's share through the HIP path on its own (extract sub-range -> batches -> raw HR tiles), shares put where the
    exchange would put them, one blend: identical bits to the unsharded run, for both partitions."""
    from innfer_amd import lib as L, synth
    from innfer_amd.parallel import ChopRunner, shard_tile_rows, shard_tiles
    from innfer_amd.utils import utils as U
    net, _ = _net(dev, 1, 2, seed=5)
    h, w = 431, 330
    x = torch.from_numpy(synth.uniform((1, 3, h, w), 21)).to(dev).half()
    ref = ChopRunner(net, 2, tile_batch=3)(x)
    ps, ys, xs = L.chop_plan(h, w, 200, 0.5)
    n = len(ys) * len(xs)
    for mode in ("tiles", "rows"):
        hr = torch.full((n, 3, 2 * ps, 2 * ps), float("nan"), dtype=torch.float16, device=dev)
        covered = 0
        for r in range(world):
            f, c = shard_tiles(n, world, r) if mode == "tiles" else shard_tile_rows(len(ys), len(xs), world, r)
            covered += c
            if not c:
                continue
            t = U.extract_patches_2d(x, (ps, ps), [0.5, 0.5], batch_first=True, tile_range=(f, c)).squeeze(0)
            for i in range(0, c, 2):
                hr[f + i:f + i + 2] = net(t[i:i + 2])
        assert covered == n
        y = U.recompose_tensor(hr, h, w, step=0.5, scale=2)
        assert torch.equal(y, ref), (mode, world)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


_RANK_SCRIPT = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["INNFER_REPO"])
from innfer_amd import synth
from innfer_amd.architectures.RRDBNet_arch import RRDBNet
from innfer_amd.parallel import ChopRunner, run_chain
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")
def net(nb, scale, seed):
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(synth.rrdbnet_shapes(nb=nb, scale=scale), seed).items()}
    m = RRDBNet(3, 3, 64, nb, upscale=scale); m.load_state_dict(sd, strict=True)
    return m.to(dev).eval()
n1, n2 = net(1, 1, 3), net(1, 2, 4)
x = torch.from_numpy(synth.uniform((1, 3, 431, 330), 21)).to(dev).half()
r1, r2 = ChopRunner(n1, 1, tile_batch=3, profile=True), ChopRunner(n2, 2, tile_batch=3, shard=os.environ["SHARD"], profile=True)
y = r2(x)
z = run_chain([r1, r2], x)
if rank == 0:
    np.savez(os.environ["OUT"], y=y.cpu().numpy(), z=z.cpu().numpy(), xbytes=r2.last["exchange_bytes"], mine=r2.last["tiles"], total=r2.last["tiles_total"])
else:
    assert y is None and z is None
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,shard", [(2, "tiles"), (3, "rows")])
def test_two_rank_processes_equal_single_process(dev, tmp_path, world, shard):
    from innfer_amd import synth
    from innfer_amd.parallel import ChopRunner, run_chain
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    out = str(tmp_path / "out.npz")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   INNFER_REPO=REPO, OUT=out, SHARD=shard, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    got = np.load(out)
    n1, _ = _net(dev, 1, 1, seed=3)
    n2, _ = _net(dev, 1, 2, seed=4)
    x = torch.from_numpy(synth.uniform((1, 3, 431, 330), 21)).to(dev).half()
    y = ChopRunner(n2, 2, tile_batch=3)(x)
    z = run_chain([ChopRunner(n1, 1, tile_batch=3), ChopRunner(n2, 2, tile_batch=3)], x)
    assert np.array_equal(got["y"], y.cpu().numpy())
    assert np.array_equal(got["z"], z.cpu().numpy())
    # only real tiles crossed: rank 0 received (total - its own) tiles of 3 x 400 x 400 fp16
    assert int(got["xbytes"]) == (int(got["total"]) - int(got["mine"])) * 3 * 400 * 400 * 2


def test_bench_two_ranks_dry_run(dev):
    """The driver's `python bench.py --gpus N` form for N = 2 on this one-GPU box: ONE line, n_gpus == 2, frame replicas as the
    headline (weak) and BASELINE config 4 sharded over both ranks in `tile_sharded` (strong)."""
    env = dict(os.environ, INNFER_BENCH_DRYRUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "INNFER_BENCH_SELFTEST"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "frame540",
                        "--sharded-steps", "1", "--no-power-probe", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "DRY RUN" in line["config"]["parallelism"]
    ts = line["tile_sharded"]
    assert ts["parallelism"] == "tile-dp2" and ts["scaling"] == "strong"
    # 798 tiles per stage, rank 1 owns 399: their raw outputs (1x stage: 3x200x200, 4x stage: 3x800x800, fp16) cross to rank 0
    assert ts["exchange_bytes_into_rank0"] == 399 * 3 * 2 * (200 ** 2 + 800 ** 2)
    # every rank reports the phases of the profiled pass: [rank][stage] with its own tile count
    pr = ts["per_rank_phases_ms"]
    assert len(pr) == 2 and all(len(stages) == 2 for stages in pr)
    assert [st["tiles"] for st in pr[0]] == [399, 399] and [st["tiles"] for st in pr[1]] == [399, 399]
    assert all({"compute_ms", "exchange_ms", "blend_ms", "bcast_ms"} <= set(st) for stages in pr for st in stages)
    rf = line["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == ("GB/s" if rf["bound"] == "hbm" else "TFLOP/s")
    assert 0 < rf["frac"] <= 1 and 0 < rf["frac_mfma"] <= 1 and 0 < rf["frac_hbm"] <= 1


def test_cabi_comm_single_rank(dev):
    """The C ABI's multi-GPU entry points (innfer_comm_*, innfer_gather_tiles) on the one GPU of this box: RCCL is bound at run time, a
    one-rank communicator is created on the current device, gather / broadcast of a one-rank communicator leave the buffer alone, and the
    host partition equals innfer_amd.parallel.shard_tiles.  (Two ranks need two GPUs: RCCL refuses two ranks on one device.)"""
    import ctypes as C
    import innfer_amd.lib as L
    from innfer_amd.parallel import shard_tiles
    for n, world in [(798, 8), (3268, 8), (5, 7), (0, 3)]:
        assert [L.shard_tiles(n, world, r) for r in range(world)] == [shard_tiles(n, world, r) for r in range(world)]
    torch.cuda.set_device(dev)
    buf = (C.c_char * L.COMM_ID_BYTES)()
    L.check(L.lib.innfer_comm_unique_id(buf))
    assert any(bytes(buf))
    h = C.c_void_p()
    L.check(L.lib.innfer_comm_init(C.byref(h), buf, 0, 1))
    assert L.lib.innfer_comm_rank(h) == 0 and L.lib.innfer_comm_size(h) == 1
    t = torch.arange(6 * 3 * 8 * 8, dtype=torch.float16, device=dev).reshape(6, 3, 8, 8)
    ref = t.clone()
    s = torch.cuda.current_stream(dev).cuda_stream
    L.check(L.lib.innfer_gather_tiles(h, t.data_ptr(), 3 * 8 * 8 * 2, 6, s))
    L.check(L.lib.innfer_comm_broadcast(h, t.data_ptr(), t.numel() * 2, 0, s))
    torch.cuda.synchronize()
    assert torch.equal(t, ref)
    with pytest.raises(ValueError):
        L.check(L.lib.innfer_comm_broadcast(h, t.data_ptr(), 16, 3, s))      # root outside the communicator
    L.lib.innfer_comm_destroy(h)


_NCCL_RANK_SCRIPT = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["INNFER_REPO"])
from innfer_amd import synth
from innfer_amd.architectures.RRDBNet_arch import RRDBNet
from innfer_amd.parallel import ChopRunner, run_chain
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)          # RCCL over xGMI: one process per GPU
def net(nb, scale, seed):
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(synth.rrdbnet_shapes(nb=nb, scale=scale), seed).items()}
    m = RRDBNet(3, 3, 64, nb, upscale=scale); m.load_state_dict(sd, strict=True)
    return m.to(dev).eval()
n1, n2 = net(1, 1, 3), net(1, 2, 4)
x = torch.from_numpy(synth.uniform((1, 3, 431, 330), 21)).to(dev).half()
res = {}
for transport in ("torch", "cabi"):
    r1 = ChopRunner(n1, 1, tile_batch=3, profile=True, transport=transport)
    r2 = ChopRunner(n2, 2, tile_batch=3, profile=True, transport=transport)
    y = r2(x)
    z = run_chain([r1, r2], x)
    if rank == 0:
        res.update({f"y_{transport}": y.cpu().numpy(), f"z_{transport}": z.cpu().numpy(), f"xbytes_{transport}": r2.last["exchange_bytes"],
                    "mine": r2.last["tiles"], "total": r2.last["tiles_total"]})
    else:
        assert y is None and z is None
if rank == 0:
    np.savez(os.environ["OUT"], **res)
dist.barrier()
dist.destroy_process_group()
'''


def test_first_contact_real_rccl_two_gpus(dev, tmp_path):
    """Two rank processes on TWO devices over the nccl (= RCCL) backend: ChopRunner with transport='torch' (batch_isend_irecv on RCCL,
    init_process_group(device_id=...)) and transport='cabi' (the library's own communicator: innfer_comm_init / innfer_gather_tiles /
    innfer_comm_broadcast) must both return the single-process result bit for bit, with exactly the other rank's tiles crossing.
    Skips on a one-GPU box; it is the first thing a multi-GPU box runs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    from innfer_amd import synth
    from innfer_amd.parallel import ChopRunner, run_chain
    script = tmp_path / "rank_nccl.py"
    script.write_text(_NCCL_RANK_SCRIPT)
    out = str(tmp_path / "out_nccl.npz")
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   INNFER_REPO=REPO, OUT=out, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    got = np.load(out)
    n1, _ = _net(dev, 1, 1, seed=3)
    n2, _ = _net(dev, 1, 2, seed=4)
    x = torch.from_numpy(synth.uniform((1, 3, 431, 330), 21)).to(dev).half()
    y = ChopRunner(n2, 2, tile_batch=3)(x).cpu().numpy()
    z = run_chain([ChopRunner(n1, 1, tile_batch=3), ChopRunner(n2, 2, tile_batch=3)], x).cpu().numpy()
    for transport in ("torch", "cabi"):
        assert np.array_equal(got[f"y_{transport}"], y), transport
        assert np.array_equal(got[f"z_{transport}"], z), transport
        assert int(got[f"xbytes_{transport}"]) == (int(got["total"]) - int(got["mine"])) * 3 * 400 * 400 * 2, transport


def test_bench_counts_gpus_without_touching_hip():
    """bench.py's parent process decides how many ranks it may start from the KFD topology (sysfs), not from a HIP call: a process that has
    initialised the GPU must never be replaced, and the parent stays clean.  On this box the count is >= 1 and agrees with torch's."""
    sys.path.insert(0, REPO)
    import bench
    n = bench.visible_gpu_count()
    assert n >= 1 and n == torch.cuda.device_count()


def test_key_addressed_engine_fp32_tiles_on_a_non_current_gpu(dev):
    """ParamEngineModule.tile_batch_bytes uploads the weights and builds the fp32 panels -- on the device the caller names, not on whatever GPU happens
    to be current; the forward that follows launches there with those pointers.  PAN in -no_fp16 mode sized and run on cuda:1 while cuda:0 is current, and a PPON forward on
    cuda:1 (its own forward override).  Skips on a one-GPU box."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from innfer_amd import synth
    from innfer_amd.architectures import get_network
    from innfer_amd.parallel import engine_tile_cap, run_tile_batches
    from innfer_amd.utils.defaults import get_network_G_config
    d1 = torch.device("cuda", 1)
    torch.cuda.set_device(0)
    for arch, kw in (("pan", {"nb": 2}), ("ppon", {"nb": 2})):
        net = get_network(get_network_G_config({"type": arch, **kw}, 4))
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 3).items()}, strict=True)
        net = net.to(d1).eval()
        tiles = torch.from_numpy(synth.uniform((3, 3, 24, 24), 4)).to(d1)            # float32: the fp32 mode
        cap = engine_tile_cap(net, 24, torch.float32, d1)
        assert cap >= 1 and net._weights_device == d1 and torch.cuda.current_device() == 0
        pick = (lambda y: y[2]) if arch == "ppon" else None
        y = run_tile_batches(net, tiles, 2, pick=pick)
        assert y.device == d1 and torch.isfinite(y).all()
        ref = get_network(get_network_G_config({"type": arch, **kw}, 4))
        ref.load_state_dict(net.state_dict(), strict=True)
        with torch.cuda.device(d1):
            r = ref.to(d1).eval()(tiles)
        assert torch.equal(y, r[2] if arch == "ppon" else r)
        with pytest.raises(NotImplementedError):                                      # one engine, one GPU: a second device is refused, not served with foreign pointers
            net.tile_batch_bytes(1, 24, torch.float32, device=torch.device("cuda", 0))
