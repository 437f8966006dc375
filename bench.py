#!/usr/bin/env python3
"""Headline benchmark: output MPix/s of 4x ESRGAN RRDBNet-23 (fp16) on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (`Model.__call__(chop=False)` semantics: the whole
generator forward through libinnfer_amd.so) over one synthetic frame that is already
resident in HBM.  Workloads:
    frame1080 (default)  1x3x1080x1920 -> 1x3x4320x7680, un-tiled   (BASELINE config 2)
    frame540             1x3x540x960   -> 1x3x2160x3840  (the "->4K" reading of the metric)
    chop8k               4320x7680 input through chop_forward (3268 tiles of 200^2, blend)
With N > 1 ranks every rank runs the same workload on its own frame (frame-level data
parallel replicas: an un-tiled frame cannot be split without halo exchange over the
~348-px receptive radius) -- weak scaling, no data-path collective; `chop8k` shards the
tile list over ranks and gathers HR tiles on rank 0 (innfer_amd/parallel.py).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     : dominant kernel (conv3x3_mfma, by summed time) -- algorithmic FLOPs per launch /
                 its average launch duration, both from HIP events around every launch of one forward
  cpu_baseline : the oracle (torch fp32 restatement of the reference) timed on the host cores on a
                 bounded sample (rank 0, N == 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F16_TFLOPS = 2516.6      # MI355X dense fp16/bf16 MFMA: 256 CU x 4096 FLOP/clk x 2.4 GHz
KIND_NAMES = {0: "first_conv_kernel"}
# rows per wave of the instantiation conv_launch picks for NT 16-channel tiles (csrc/conv3x3.hip), slab / NCHW output
RPW_OF_NT = {1: 4, 2: int(os.environ.get("INNFER_RPW32", "5")), 4: int(os.environ.get("INNFER_RPW64", "3"))}
PC = int(os.environ.get("INNFER_PC", "1"))      # slab-output convs run the producer / consumer kernel
PC_SHAPE = {(2, 0): (3, 2, 8 if PC == 2 else 4, 0), (4, 0): (2, 4, 4, 0), (1, 1): (3, 1, 4, 1)}


def kernel_key(k):
    """rocprof-style name of the instantiation conv_launch picks for launch kind k = 16*NT + out_mode."""
    nt, mode = k // 16, k % 16
    if PC and (nt, mode) in PC_SHAPE:
        return "conv3x3_pc<%d,%d,%d,%d>" % PC_SHAPE[(nt, mode)]
    return f"conv3x3_mfma<{RPW_OF_NT[nt]},{nt},{mode}>"


def kind_name(k):
    return "first_conv_kernel" if k == 0 else kernel_key(k)


def build_net(dev, nb=23, scale=4):
    import torch
    from innfer_amd import synth
    from innfer_amd.architectures.RRDBNet_arch import RRDBNet
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(synth.rrdbnet_shapes(nb=nb, scale=scale), 0).items()}
    net = RRDBNet(3, 3, 64, nb, upscale=scale)
    net.load_state_dict(sd, strict=True)
    return net.to(dev).eval(), sd


def timed_forward(net, x):
    """Per-launch HIP-event timing of one forward (plain schedule) through the C ABI."""
    import torch
    import innfer_amd.lib as L
    net._ensure_engine()
    N, _, H, W = x.shape
    s = L.lib.innfer_net_scale(net._handle)
    out = torch.empty((N, net.out_nc, H * s, W * s), dtype=x.dtype, device=x.device)
    need = L.lib.innfer_net_workspace_bytes(net._handle, N, H, W)
    if net._ws is None or net._ws.numel() < need:
        net._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
    cap = 4096
    ms, fl, kd, n = (C.c_float * cap)(), (C.c_double * cap)(), (C.c_int * cap)(), C.c_int()
    stream = torch.cuda.current_stream(x.device).cuda_stream
    L.check(L.lib.innfer_net_set_band_rows(net._handle, int(net.band_rows)))
    L.check(L.lib.innfer_net_forward_timed(net._handle, x.data_ptr(), L.F16, out.data_ptr(), L.F16, N, H, W,
                                           net._ws.data_ptr(), net._ws.numel(), stream, cap, ms, fl, kd, C.byref(n)))
    return [(kd[i], ms[i], fl[i]) for i in range(min(n.value, cap))]


def pmc_traffic(kind):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json, written by scripts/profile.sh + summarize_prof.py), or None."""
    try:
        t = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
        return t[kernel_key(kind)]["hbm_bytes_per_launch"]
    except Exception:
        return None


def roofline_from_launches(launches):
    agg = {}
    for k, ms, fl in launches:
        a = agg.setdefault(k, [0.0, 0.0, 0])
        a[0] += ms; a[1] += fl; a[2] += 1
    dom = max(agg, key=lambda k: agg[k][0])
    t_ms, flops, cnt = agg[dom]
    achieved = flops / (t_ms * 1e-3) / 1e12
    per_kernel = {kind_name(k): {"launches": v[2], "ms_total": round(v[0], 4), "avg_ms": round(v[0] / v[2], 5),
                                 "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 2)} for k, v in agg.items()}
    total_ms = sum(v[0] for v in agg.values())
    total_fl = sum(v[1] for v in agg.values())
    return {"bound": "mfma", "kernel": kind_name(dom), "launches": cnt,
            "avg_launch_ms": round(t_ms / cnt, 5), "flops_per_launch": flops / cnt,
            "achieved": round(achieved, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": pmc_traffic(dom),
            "traffic_note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB averaged over the kernel's "
                            "dispatches, separate rocprofv3 --pmc passes (profiles/traffic.json)",
            "all_kernels_tflops": round(total_fl / (total_ms * 1e-3) / 1e12, 2),
            "per_kernel": per_kernel}


def power_probe(step, seconds=2.5):
    """Package power and shader clock the GPU holds under THIS workload: `step` is enqueued back to back for ~`seconds`
    while a thread samples `rocm-smi` (about one sample per 0.4 s; the first second is discarded so that the power manager has
    settled).  Context for the roofline object only -- the peak in `roofline.peak` is the 2.4 GHz figure; never part of
    the timed region.  Returns None when rocm-smi is not there."""
    import re, shutil, subprocess, threading
    import torch
    if not shutil.which("rocm-smi"):
        return None
    samples, stop = [], threading.Event()

    def sampler():
        t0 = time.perf_counter()
        while not stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            except Exception:
                return
            clk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
            pw = re.search(r"Package Power \(W\): ([0-9.]+)", out)
            if clk and pw and time.perf_counter() - t0 > 1.0:
                samples.append((int(clk.group(1)), float(pw.group(1))))

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    stop.set()
    th.join(timeout=15)
    if not samples:
        return None
    cap = None
    try:
        m = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)",
                      subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout)
        cap = float(m.group(1)) if m else None
    except Exception:
        pass
    sclk = sum(c for c, _ in samples) / len(samples)
    return {"sclk_mhz": round(sclk, 0), "package_w": round(sum(w for _, w in samples) / len(samples), 0), "cap_w": cap,
            "samples": len(samples), "peak_at_sclk_tflops": round(PEAK_F16_TFLOPS * sclk / 2400.0, 1)}


def per_layer_table(launches, npix_lr):
    """stderr table: per distinct (kind, flops) launch class -> avg ms, TFLOP/s (diagnostics)."""
    rows = {}
    for k, ms, fl in launches:
        r = rows.setdefault((k, fl), [0.0, 0])
        r[0] += ms; r[1] += 1
    out = []
    for (k, fl), (ms, n) in sorted(rows.items()):
        out.append(f"  {kind_name(k):34s} n={n:3d} GFLOP={fl / 1e9:9.1f} avg_ms={ms / n:8.4f} TFLOP/s={fl / (ms / n * 1e-3) / 1e12:8.1f}")
    return "\n".join(out)


def usable_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline_child(budget_s=12.0):
    """Runs in a child process that never touches the GPU: the oracle (== the reference's -cpu
    fp32 path, pinned by golden vectors) on the host cores, bounded sample."""
    import torch
    import oracle
    from innfer_amd import synth
    cores = min(usable_cores(), 64)
    torch.set_num_threads(cores)
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(synth.rrdbnet_shapes(nb=23, scale=4), 0).items()}
    x = torch.from_numpy(synth.uniform((1, 3, 128, 128), 1))
    with torch.no_grad():
        oracle.rrdbnet_forward(sd, x, nb=23, scale=4)          # warm-up
        times = []
        t_all = time.perf_counter()
        while time.perf_counter() - t_all < budget_s and len(times) < 20:
            t0 = time.perf_counter()
            oracle.rrdbnet_forward(sd, x, nb=23, scale=4)
            times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    print(json.dumps({"value": round(512 * 512 / med / 1e6, 4), "unit": "output MPix/s", "cores": cores,
                      "kind": "port",
                      "sample": f"RRDBNet-23 4x fp32, 1x3x128x128 un-tiled (BASELINE config 1), median of "
                                f"{len(times)} runs, {med:.3f} s/run, torch {torch.__version__} CPU"}), flush=True)


def cpu_baseline(timeout_s=120):
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child"],
                           capture_output=True, text=True, timeout=timeout_s)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "output MPix/s", "cores": usable_cores(), "kind": "port",
                "sample": "cpu baseline child failed: " + r.stderr[-200:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "output MPix/s", "cores": usable_cores(), "kind": "port",
                "sample": f"cpu baseline child exceeded {timeout_s} s"}


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="frame1080", choices=["frame1080", "frame540", "chop8k", "chop4k", "chain4k"])
    ap.add_argument("--band-rows", type=int, default=int(os.environ.get("INNFER_BAND_ROWS", "0")))
    ap.add_argument("--tile-batch", type=int, default=0, help="chop workloads: tiles per network launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-power-probe", action="store_true")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        return cpu_baseline_child()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # INNFER_BENCH_DRYRUN=1: control-flow rehearsal of the N>1 path on a ONE-GPU box (every rank on cuda:0, gloo
    # rendezvous); never a measurement.
    dryrun = world > 1 and os.environ.get("INNFER_BENCH_DRYRUN") == "1"
    if dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if dryrun:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from innfer_amd import synth
    net, _ = build_net(dev)
    net.band_rows = args.band_rows

    if args.workload.startswith("frame"):
        H, W = (1080, 1920) if args.workload == "frame1080" else (540, 960)
        x = torch.from_numpy(synth.uniform((1, 3, H, W), 2 + rank)).to(dev).half()
        out_pix_per_rank = 16 * H * W

        def step():
            return net(x)
        cfg = {"workload": f"ESRGAN RRDBNet-23 4x fp16, 1x3x{H}x{W} -> 1x3x{4 * H}x{4 * W}, un-tiled "
                           f"(Model(chop=False)), one frame per rank", "band_rows": args.band_rows,
               "parallelism": (f"frame replicas x{world}" if world > 1 else "single GPU") + (" (DRY RUN: all ranks on one GPU)" if dryrun else "")}
    else:
        from innfer_amd import parallel
        H, W = (4320, 7680) if args.workload == "chop8k" else (2160, 3840)       # chop4k and chain4k: 4K input
        x = torch.from_numpy(synth.uniform((1, 3, H, W), 2)).to(dev).half()
        runner = parallel.ChopRunner(net, scale=4, tile_batch=args.tile_batch or 64)
        out_pix_per_rank = 16 * H * W / world       # unique output pixels of the ONE shared frame
        if args.workload == "chain4k":              # BASELINE config 4: model chain 1x + 4x (run.py:424-426)
            net1, _ = build_net(dev, scale=1)
            runner1 = parallel.ChopRunner(net1, scale=1, tile_batch=args.tile_batch or 64)

            def step():
                return parallel.run_chain([runner1, runner], x)
            what = "model chain RRDBNet-23 1x + RRDBNet-23 4x"
        else:
            def step():
                return runner(x)
            what = "ESRGAN RRDBNet-23 4x"
        cfg = {"workload": f"{what} fp16, {H}x{W} input through chop_forward (patch 200, step 0.5), "
                           f"tiles sharded over {world} rank(s), HR tiles gathered + blended on rank 0",
               "band_rows": args.band_rows, "parallelism": f"tile-dp{world}"}

    log('warmup')
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        y = step()
    e1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    t = torch.tensor([wall], dtype=torch.float64, device="cpu" if dryrun else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = t.item()
    log(f'timed {args.steps} steps in {wall:.3f} s')
    ev_ms = e0.elapsed_time(e1)
    del y

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        value = out_pix_per_rank * world * args.steps / wall / 1e6
        line = {"metric": "output MPix/s, 4x ESRGAN RRDB-23 1080p->4K", "value": round(value, 2),
                "unit": "MPix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f16", "data": "synthetic", "config": cfg,
                "hip_event_ms_per_step": round(ev_ms / args.steps, 3)}
        flops = net.flops(1, H, W) if args.workload.startswith("frame") else None
        if flops:
            line["model_tflops"] = round(flops * world * args.steps / wall / 1e12, 2)
            line["frac_of_mfma_peak"] = round(flops * args.steps / wall / 1e12 / PEAK_F16_TFLOPS, 4)
        if not args.no_roofline and args.workload.startswith("frame"):
            log('per-launch timing')
            timed_forward(net, x)
            launches = timed_forward(net, x)
            line["roofline"] = roofline_from_launches(launches)
            if world == 1 and not args.no_power_probe:
                log('power probe')
                pw = power_probe(step)
                if pw:
                    pw["frac_of_peak_at_sclk"] = round(line["roofline"]["achieved"] / pw["peak_at_sclk_tflops"], 4)
                    line["roofline"]["power"] = pw
            log("per-layer classes:\n" + per_layer_table(launches, H * W))
        net.release_workspace()
        if world == 1 and not args.no_cpu_baseline:
            log('cpu baseline')
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
