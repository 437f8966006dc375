#!/usr/bin/env python3
"""Headline benchmark: output MPix/s of 4x ESRGAN RRDBNet-23 (fp16) on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: this process starts N fresh rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (`Model.__call__(chop=False)` semantics: the whole
generator forward through libinnfer_amd.so) over one synthetic frame that is already
resident in HBM.  Workloads (`--workload`):
    frame1080 (default)  1x3x1080x1920 -> 1x3x4320x7680, un-tiled   (BASELINE config 2)
    frame540             1x3x540x960   -> 1x3x2160x3840  (the "->4K" reading of the metric)
    chop8k               4320x7680 input through chop_forward (3268 tiles of 200^2, blend)   (BASELINE config 3)
    chop4k               2160x3840 input through chop_forward (798 tiles)
    chain4k              model chain RRDBNet-23 1x + 4x on a 2160x3840 input, 798 tiles per stage   (BASELINE config 4)
frame*: with N > 1 ranks every rank runs the workload on its own frame (frame-level replicas: an un-tiled
frame cannot be split without halo exchange over the ~348-px receptive radius) -- weak scaling, no data-path
collective.  chop* / chain*: ONE shared frame, its tile list sharded over the ranks, raw HR tiles sent to rank 0
over RCCL and blended there (innfer_amd/parallel.py) -- strong scaling.
Whatever the headline workload, the line also carries `tile_sharded`: BASELINE config 4 (chain4k) timed the same
way (barrier + synchronise on both sides, max over ranks) on the SAME ranks, with the exchange bytes / ms of one
profiled pass -- the north_star's "tiles partition across the GPUs, reassembled with an RCCL gather".

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     : dominant kernel (by summed time) -- algorithmic FLOPs and bytes per launch / its average launch
                 duration, from HIP events around every launch of one forward; BOTH roofs per kernel, `bound` = the larger floor
  cpu_baseline : the oracle (torch fp32 restatement of the reference) timed on the host cores on a
                 bounded sample (rank 0, in a child process that never touches a GPU; at every N).
  chop8k       : BASELINE config 3 (8K input through chop_forward, 3268 tiles) -- 1 warm-up + 1 timed pass on the same ranks
  unet64       : BASELINE config 5 (pix2pix UNet_256 on 64x3x256x256), N == 1 only
  pan540       : PAN 4x on a 1x3x540x960 frame (north_star's third conv stack), N == 1 only (+ pan_chop1080: the CLI's tiled shape; pan540_fp32: the -no_fp16 mode)
  srresnet1080 : SRResNet-16 4x on the 1080p frame (north_star's second conv stack), N == 1 only
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F16_TFLOPS = 2516.6      # MI355X dense fp16/bf16 MFMA: 256 CU x 4096 FLOP/clk x 2.4 GHz
PEAK_HBM_GBS = 8000.0         # HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
# instantiation conv_launch picks for NT 16-channel output tiles and the output mode (csrc/conv3x3.hip)
PC_SHAPE = {(2, 0): (3, 2, 4, 0), (4, 0): (2, 4, 4, 0), (1, 1): (3, 1, 4, 1)}
MFMA_SHAPE = {1: 4, 2: 4, 4: 2}
WORKLOADS = ["frame1080", "frame540", "chop8k", "chop4k", "chain4k"]


def kernel_key(k):
    """Name of the instantiation family conv_launch picks for launch kind k (csrc/net.hip do_conv): 16*NT + out_mode, + 1000 the fp32-accurate form,
    + 2000 HR_conv0 + conv_last fused (rocprof: conv3x3_pc<2,4,4,0,..,TMF 4325887 = 0x4201FF>), + 3000 an up-conv as four 2x2-tap phases (TMF 6291967 =
    0x6001FF).  The plain 64-output row holds both the plain (TMF 4194815) and the residual-from-LDS (4456959) launches; profiles/traffic.json lists the
    instantiations by their TMF."""
    variant, k = k // 1000, k % 1000
    nt, mode = k // 16, k % 16
    if variant == 5:          # the last upconv_block -> HR_conv0 -> conv_last chained through LDS (csrc/hr_chain.hip)
        return "hr_chain_kernel<upconv+HR_conv0+conv_last>"
    if variant == 4:          # conv nf -> 4 nf with nn.PixelShuffle(2) as the store of conv3x3_pc<2,4,4,0,..,TMF 0xC001FF> (phase-major plane-order panels)
        return "conv3x3_pc<2,4,4,0>+pixelshuffle"
    if (nt, mode) in PC_SHAPE:
        name = "conv3x3_pc<%d,%d,%d,%d>" % PC_SHAPE[(nt, mode)]
    else:
        name = f"conv3x3_mfma<{MFMA_SHAPE[nt]},{nt},{mode}>"
    return name + {0: "", 1: "+split", 2: "+fused_tail", 3: "+upconv_phases"}[variant]


def kind_name(k):
    return "first_conv_mfma<4,1>" if k == 0 else kernel_key(k)          # (3 -> 64 on the matrix cores, csrc/conv_first.hip)


KIND_NOTES = {"hr_chain_kernel<upconv+HR_conv0+conv_last>": "the last upconv_block -> HR_conv0 -> conv_last as one kernel chained through LDS: the three layers' ALGORITHMIC FLOPs (nine taps each on the "
                                                            "HR grid; executed: 4/9 of the up-conv's x 1.2 for the recomputed halo), the 64-channel HR tensor never written",
              "+fused_tail": "HR_conv0 + conv_last in one launch (+ the rim pass): both convs' FLOPs, the 64-channel HR tensor neither written nor read",
              "+upconv_phases": "up-conv as four 2x2-tap phases on the LR grid (all four in one visit of a tile): FLOPs are the ALGORITHMIC ones of the nine-tap layer it replaces "
                       "(reference block.py:348-361); executed FLOPs = 4/9 of them, so `frac_mfma` here is not the matrix pipe's utilisation"}


XGMI_LINK_GBS = 153.0          # per link and direction (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, point to point)
XGMI_PAYLOAD_EFF = 0.8         # payload share of a large point-to-point transfer the model assumes (unmeasured on this pool: no multi-GPU box has run it)


def predict_tile_scaling(stages, out_channels=3, worlds=(1, 2, 4, 8)):
    """The strong-scaling table the first multi-GPU run is judged against, from ONE rank's measured phases.
    stages: per chained model {tiles_total, ms_per_tile (this rank's compute ms / its tiles), blend_ms, tile_bytes (one raw HR tile), bcast_bytes (the blended
    intermediate handed to the next model, 0 for the last)}.  Model (innfer_amd/parallel.py): the row-major tile list is split evenly, so the slowest rank computes
    ceil(tiles / N) tiles at the measured per-tile cost (tiles are independent: run.py:186-197); every other rank then sends its raw HR tiles to rank 0 point to point --
    its own xGMI link each (N <= 8: 7 links per GPU), all in one grouped send/recv, so the exchange lasts one peer's share / (link rate x payload efficiency);
    rank 0 blends alone (measured, does not shrink); between chained models the intermediate is broadcast: one link time."""
    import math
    link = XGMI_LINK_GBS * 1e9 * XGMI_PAYLOAD_EFF
    rows, t1 = [], None
    for n in worlds:
        comp = exch = blend = bcast = 0.0
        for st in stages:
            share = math.ceil(st["tiles_total"] / n)
            comp += share * st["ms_per_tile"]
            if n > 1:
                exch += (st["tiles_total"] // n) * st["tile_bytes"] / link * 1e3          # (a peer's share; rank 0 keeps its own)
                bcast += st["bcast_bytes"] / link * 1e3
            blend += st["blend_ms"]
        total = comp + exch + blend + bcast
        t1 = total if n == 1 else t1
        rows.append({"n_gpus": n, "ms_per_frame": round(total, 2), "compute_ms": round(comp, 2), "exchange_ms": round(exch, 2), "blend_ms": round(blend, 2),
                     "bcast_ms": round(bcast, 2), "speedup": round(t1 / total, 3), "efficiency": round(t1 / total / n, 3)})
    return rows


def stage_model(phases, ps, scales, last_bcast=False):
    """predict_tile_scaling's input from the profiled pass of one rank (ChopRunner.last per chained model)."""
    out = []
    for i, (p, sc) in enumerate(zip(phases, scales)):
        P = ps * sc
        out.append({"tiles_total": p["tiles_total"], "ms_per_tile": p["compute_ms"] / max(1, p["tiles"]), "blend_ms": p["blend_ms"],
                    "tile_bytes": 3 * P * P * 2, "bcast_bytes": 0 if (i == len(phases) - 1 and not last_bcast) else 3 * p.get("out_h", 0) * p.get("out_w", 0) * 2})
    return out


def build_net(dev, nb=23, scale=4):
    import torch
    from innfer_amd import synth
    from innfer_amd.architectures.RRDBNet_arch import RRDBNet
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(synth.rrdbnet_shapes(nb=nb, scale=scale), 0).items()}
    net = RRDBNet(3, 3, 64, nb, upscale=scale)
    net.load_state_dict(sd, strict=True)
    return net.to(dev).eval(), sd


def timed_forward(net, x):
    """Per-launch HIP-event timing of one forward (plain schedule) through the C ABI: [(kind, ms, flops, bytes)]."""
    import torch
    import innfer_amd.lib as L
    net._engine_on(x.device)
    N, _, H, W = x.shape
    s = L.lib.innfer_net_scale(net._handle)
    out = torch.empty((N, net.out_nc, H * s, W * s), dtype=x.dtype, device=x.device)
    need = L.lib.innfer_net_workspace_bytes(net._handle, N, H, W)
    if net._ws is None or net._ws.numel() < need:
        net._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
    cap = 4096
    ms, fl, by, kd, n = (C.c_float * cap)(), (C.c_double * cap)(), (C.c_double * cap)(), (C.c_int * cap)(), C.c_int()
    stream = torch.cuda.current_stream(x.device).cuda_stream
    L.check(L.lib.innfer_net_set_band_rows(net._handle, int(net.band_rows)))
    L.check(L.lib.innfer_net_set_fused_tail(net._handle, int(bool(net.fused_tail))))
    L.check(L.lib.innfer_net_set_hr_chain(net._handle, int(bool(net.hr_chain))))
    L.check(L.lib.innfer_net_set_upconv_phases(net._handle, int(net.upconv_phases)))
    L.check(L.lib.innfer_net_set_residual_lds(net._handle, int(net.residual_lds)))
    L.check(L.lib.innfer_net_forward_timed(net._handle, x.data_ptr(), L.F16, out.data_ptr(), L.F16, N, H, W,
                                           net._ws.data_ptr(), net._ws.numel(), stream, cap, ms, fl, by, kd, C.byref(n)))
    return [(kd[i], ms[i], fl[i], by[i]) for i in range(min(n.value, cap))]


def pmc_traffic(kind):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json, written by scripts/profile.sh + summarize_prof.py), or None."""
    try:
        t = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
        return t[kernel_key(kind)]["hbm_bytes_per_launch"]
    except Exception:
        return None


def two_roofs(flops, nbytes, ms):
    """Both floors of a kernel (or a sum of launches): MFMA time at the dense fp16 peak, HBM time at 8 TB/s for the
    ALGORITHMIC bytes; the binding roof is the larger floor."""
    t = ms * 1e-3
    t_mfma, t_hbm = flops / (PEAK_F16_TFLOPS * 1e12), nbytes / (PEAK_HBM_GBS * 1e9)
    return {"tflops": round(flops / t / 1e12, 2), "gbs": round(nbytes / t / 1e9, 1),
            "frac_mfma": round(t_mfma / t, 4), "frac_hbm": round(t_hbm / t, 4), "frac_binding": round(max(t_mfma, t_hbm) / t, 4),
            "bound": "mfma" if t_mfma >= t_hbm else "hbm", "flop_per_byte": round(flops / nbytes, 1)}


def roofline_from_launches(launches):
    agg = {}
    for k, ms, fl, by in launches:
        a = agg.setdefault(k, [0.0, 0.0, 0.0, 0])
        a[0] += ms; a[1] += fl; a[2] += by; a[3] += 1
    dom = max(agg, key=lambda k: agg[k][0])
    t_ms, flops, nbytes, cnt = agg[dom]
    r = two_roofs(flops, nbytes, t_ms)
    per_kernel = {}
    for k, v in agg.items():
        e = {"launches": v[3], "ms_total": round(v[0], 4), "avg_ms": round(v[0] / v[3], 5)}
        e.update(two_roofs(v[1], v[2], v[0]))
        for suffix, note in KIND_NOTES.items():
            if kind_name(k).endswith(suffix):
                e["note"] = note
        if kind_name(k).endswith("+upconv_phases"):          # the matrix pipe executes 4/9 of the nine-tap layer's algorithmic FLOPs
            e["frac_mfma_executed"] = round(e["frac_mfma"] * 4.0 / 9.0, 4)
        per_kernel[kind_name(k)] = e
    total_ms = sum(v[0] for v in agg.values())
    total_fl = sum(v[1] for v in agg.values())
    # the contract's roof for this path is the MFMA one (north_star ">= 0.5x MFMA roofline"): `bound` / `achieved` / `peak` / `frac`
    # are quoted against it for the dominant kernel -- `frac` = `frac_mfma_algorithmic` = algorithmic FLOPs / time / the 2.4 GHz dense peak, NOT a
    # utilisation of the binding resource; `binding_roof` is the two-roof rule's answer for the same launches (the larger floor) and `frac_binding`
    # the fraction against THAT roof (both under explicit names)
    out = {"bound": "mfma", "kernel": kind_name(dom), "launches": cnt, "avg_launch_ms": round(t_ms / cnt, 5),
           "flops_per_launch": flops / cnt, "bytes_per_launch": nbytes / cnt,
           "achieved": r["tflops"], "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": r["frac_mfma"], "frac_mfma_algorithmic": r["frac_mfma"],
           "binding_roof": r["bound"], "frac_binding": r["frac_binding"]}
    out.update({"frac_mfma": r["frac_mfma"], "frac_hbm": r["frac_hbm"], "tflops": r["tflops"], "gbs": r["gbs"],
                "flop_per_byte": r["flop_per_byte"], "ridge_flop_per_byte": round(PEAK_F16_TFLOPS * 1e3 / PEAK_HBM_GBS, 1),
                "traffic": pmc_traffic(dom),
                "traffic_note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB averaged over the kernel's "
                                "dispatches, separate rocprofv3 --pmc passes (profiles/traffic.json)",
                "all_kernels_tflops": round(total_fl / (total_ms * 1e-3) / 1e12, 2),
                "per_kernel": per_kernel})
    return out


def power_probe(step, seconds=2.5):
    """Package power and shader clock the GPU holds under THIS workload: `step` is enqueued back to back for ~`seconds`
    while a thread samples `rocm-smi` (about one sample per 0.4 s; the first second is discarded so that the power manager has
    settled).  Context for the roofline object only -- the peak in `roofline.peak` is the 2.4 GHz figure; never part of
    the timed region.  Returns None when rocm-smi is not there."""
    import re, shutil, threading
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


def per_layer_table(launches):
    """stderr table: per distinct (kind, flops) launch class -> avg ms, TFLOP/s, algorithmic GB/s (diagnostics)."""
    rows = {}
    for k, ms, fl, by in launches:
        r = rows.setdefault((k, fl, by), [0.0, 0])
        r[0] += ms; r[1] += 1
    out = []
    for (k, fl, by), (ms, n) in sorted(rows.items()):
        t = ms / n * 1e-3
        out.append(f"  {kind_name(k):26s} n={n:3d} GFLOP={fl / 1e9:8.1f} MB={by / 1e6:8.1f} avg_ms={ms / n:8.4f} "
                   f"TFLOP/s={fl / t / 1e12:7.1f} GB/s={by / t / 1e9:7.1f}")
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


def unet64_object(dev, reps=50, windows=7, warm_s=0.6):
    """BASELINE config 5: UnetGenerator(3, 3, 8 downs, ngf 64, BatchNorm on the statistics of each image -- run.py runs pix2pix with meval=False, one
    image at a time) on 64 x 3 x 256 x 256, fp16, synthetic weights.  Warm-up by TIME (>= `warm_s` seconds and >= 50 forwards: the first forwards
    after an engine is created / after the GPU idled have shown one-off dispatch stalls of 17-25 ms, profiles/r4/unet_stall.txt), then `windows`
    windows of `reps` forwards, each between HIP events on the launch stream; `ms_per_step` is the MEDIAN window, min / max beside it, and
    `hip_event_sum_ms` the sum of the per-launch HIP-event durations of one forward (what the kernels alone take)."""
    import torch
    from innfer_amd import synth
    from innfer_amd.architectures import get_network
    from innfer_amd.utils.defaults import get_network_G_config
    net = get_network(get_network_G_config("p2p_256", 1))
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).train()
    x = torch.from_numpy(synth.uniform((64, 3, 256, 256), 3, -1, 1)).to(dev).half()
    n_warm, t0 = 0, time.perf_counter()
    while n_warm < 50 or time.perf_counter() - t0 < warm_s:
        for _ in range(10):
            net(x)
        torch.cuda.synchronize()
        n_warm += 10
    win = []
    for _ in range(windows):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            net(x)
        e1.record()
        torch.cuda.synchronize()
        win.append(e0.elapsed_time(e1) / reps)
    ms = sorted(win)[len(win) // 2]
    fl = net.flops(64, 256, 256)                    # 12.1 GFLOP per image
    out = {"workload": "pix2pix UnetGenerator(3, 3, 8, ngf 64), train-mode BatchNorm per image, 64x3x256x256 -> 64x3x256x256 fp16 (BASELINE config 5)",
           "steps": reps, "windows": windows, "warmup": n_warm, "ms_per_step": round(ms, 4), "ms_per_step_min": round(min(win), 4),
           "ms_per_step_max": round(max(win), 4), "ms_per_step_windows": [round(w, 4) for w in win],
           "value": round(64 / ms * 1e3, 1), "unit": "img/s",
           "model_tflops": round(fl / ms / 1e9, 1), "frac_of_mfma_peak": round(fl / ms / 1e9 / PEAK_F16_TFLOPS, 4)}
    per = unet_per_kernel(net, x)
    if per:
        out["hip_event_sum_ms"] = round(sum(e["ms_total"] for e in per.values()), 4)
        out["per_kernel"] = per
    net.release_workspace()
    return out


def pan540_object(dev, reps=20, windows=7, warm_s=0.5):
    """SURVEY 8a row a12 / north_star "RRDB/SRResNet/PAN stacks": PAN 4x (nf 40, unf 24, 16 SCPA blocks, FSA self-attention; utils/defaults.py:78-89) on one
    1 x 3 x 540 x 960 frame, fp16, synthetic weights, un-tiled.  Timed like unet64 (warm-up by time, median of `windows` windows of `reps` forwards between HIP
    events); per-kernel two-roof entries from the library's launch timer.  Algorithmic work 488 952 MAC per input pixel (probed)."""
    import torch
    from innfer_amd import synth
    from innfer_amd.architectures import get_network
    from innfer_amd.utils.defaults import get_network_G_config
    net = get_network(get_network_G_config("pan", 4))
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    H, W = 540, 960
    x = torch.from_numpy(synth.uniform((1, 3, H, W), 3)).to(dev).half()
    n_warm, t0 = 0, time.perf_counter()
    while n_warm < 20 or time.perf_counter() - t0 < warm_s:
        for _ in range(5):
            net(x)
        torch.cuda.synchronize()
        n_warm += 5
    win = []
    for _ in range(windows):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            net(x)
        e1.record()
        torch.cuda.synchronize()
        win.append(e0.elapsed_time(e1) / reps)
    ms = sorted(win)[len(win) // 2]
    fl = 2.0 * 488952 * H * W
    out = {"workload": "PAN 4x (nf 40, unf 24, 16 SCPA blocks, FSA), 1x3x540x960 -> 1x3x2160x3840 fp16, un-tiled (SURVEY 8a row a12)",
           "steps": reps, "windows": windows, "warmup": n_warm, "ms_per_step": round(ms, 4), "ms_per_step_min": round(min(win), 4), "ms_per_step_max": round(max(win), 4),
           "value": round(16 * H * W / ms / 1e3, 1), "unit": "output MPix/s", "model_tflops": round(fl / ms / 1e9, 1),
           "frac_of_mfma_peak": round(fl / ms / 1e9 / PEAK_F16_TFLOPS, 4),
           "compulsory_hbm_note": "3 x 2 B in + 3 x 2 B x 16 out per input pixel = 0.05 GB per frame: this network is bound by its 40- / 24-channel intermediates, "
                                  "which the five-launch SCPA schedule moved 16 slab groups per block and the one-launch schedule moves 4 (DESIGN: PAN)"}
    per = unet_per_kernel(net, x)
    if per:
        out["hip_event_sum_ms"] = round(sum(e["ms_total"] for e in per.values()), 4)
        out["per_kernel"] = per
    net.release_workspace()
    # ... and PAN where the command line runs it: run.py:373-375 hard-codes chop=True, patch 200 for every SR architecture, so a 1080p frame
    # reaches PAN as 190 tiles of 200 x 200 (10 x 19, step 0.5) -> extract, batches through the network, overlap blend (Model.chop_forward, run.py:167-202)
    try:
        from innfer_amd import parallel
        Hc, Wc = 1080, 1920
        xc = torch.from_numpy(synth.uniform((1, 3, Hc, Wc), 4)).to(dev).half()
        runner = parallel.ChopRunner(net, scale=4)
        for _ in range(3):
            runner(xc)
        torch.cuda.synchronize()
        cw = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                runner(xc)
            e1.record()
            torch.cuda.synchronize()
            cw.append(e0.elapsed_time(e1) / 4)
        cms = sorted(cw)[len(cw) // 2]
        import innfer_amd.lib as L
        ps, ys, xs = L.chop_plan(Hc, Wc, 200, 0.5)
        ntile = len(ys) * len(xs)
        out["chop1080"] = {"workload": f"PAN 4x fp16, 1x3x{Hc}x{Wc} through chop_forward: {ntile} tiles of {ps}^2 (patch 200, step 0.5), extract + network + overlap blend -- the "
                                       "shape the reference's command line gives PAN (run.py:373-375)",
                           "tiles": ntile, "tile_batches": parallel.tile_batches(ntile, None, parallel.engine_tile_cap(net, ps, torch.float16, dev)),
                           "ms_per_frame": round(cms, 3), "ms_min": round(min(cw), 3), "ms_max": round(max(cw), 3), "value": round(16 * Hc * Wc / cms / 1e3, 1),
                           "unit": "unique-output MPix/s", "model_tflops": round(2.0 * 488952 * ntile * ps * ps / cms / 1e9, 1)}
        net.release_workspace()
    except Exception as e:                              # a side object must never cost the rest of the line
        out["chop1080"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    # ... and the same frame as a float32 tensor = the reference's -no_fp16 mode (run.py:345,421-422; SURVEY 8c "fp32 path <= 1e-4"): since round 6 the SCPA blocks, the
    # HR side and the attention run on (hi, lo) fp16 operand pairs of the matrix cores (csrc/pan_scpa_split.hip, conv3x3_pc SPLIT, pan_attention_mfma<true>)
    try:
        xf = x.float()
        for _ in range(5):
            net(xf)
        torch.cuda.synchronize()
        fw = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                net(xf)
            e1.record()
            torch.cuda.synchronize()
            fw.append(e0.elapsed_time(e1) / 10)
        fms = sorted(fw)[len(fw) // 2]
        out["fp32"] = {"workload": "PAN 4x, 1x3x540x960 float32 -> 1x3x2160x3840 float32, un-tiled: the -no_fp16 mode, <= 1e-4 of the fp32 reference (tests/test_gpu_fp32_mode.py)",
                       "ms_per_step": round(fms, 4), "ms_per_step_min": round(min(fw), 4), "ms_per_step_max": round(max(fw), 4), "value": round(16 * H * W / fms / 1e3, 1),
                       "unit": "output MPix/s", "dtype": "f32 as fp16 (hi, lo) pairs", "x_fp16_engine": round(fms / ms, 2)}
        per32 = unet_per_kernel(net, xf)
        if per32:
            out["fp32"]["per_kernel"] = per32
        net.release_workspace()
    except Exception as e:
        out["fp32"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def srresnet1080_object(dev, reps=10, windows=7, warm_s=0.5):
    """SURVEY 8a row a11 / north_star "RRDB/SRResNet/PAN stacks": SRResNet-16 4x as utils/defaults.py:53-67 builds it (no norm, ReLU, CNA, pixelshuffle,
    res_scale 1; SRResNet_arch.py:15-91) on one 1 x 3 x 1080 x 1920 frame, fp16, synthetic weights, un-tiled.  Timed like unet64 (warm-up by time, median
    of `windows` windows of `reps` forwards between HIP events); per-kernel two-roof entries from innfer_net_forward_timed (the RRDBNet engine's timer).
    Algorithmic work 2 572 992 MAC per input pixel (probed) = net.flops()."""
    import torch
    from innfer_amd import synth
    from innfer_amd.architectures import get_network
    from innfer_amd.utils.defaults import get_network_G_config
    net = get_network(get_network_G_config("srgan", 4))
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    H, W = 1080, 1920
    x = torch.from_numpy(synth.uniform((1, 3, H, W), 3)).to(dev).half()
    n_warm, t0 = 0, time.perf_counter()
    while n_warm < 10 or time.perf_counter() - t0 < warm_s:
        for _ in range(5):
            net(x)
        torch.cuda.synchronize()
        n_warm += 5
    win = []
    for _ in range(windows):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            net(x)
        e1.record()
        torch.cuda.synchronize()
        win.append(e0.elapsed_time(e1) / reps)
    ms = sorted(win)[len(win) // 2]
    fl = net.flops(1, H, W)
    out = {"workload": "SRResNet-16 4x (nf 64, no norm, ReLU, CNA, pixelshuffle: utils/defaults.py:53-67), 1x3x1080x1920 -> 1x3x4320x7680 fp16, un-tiled (SURVEY 8a row a11)",
           "steps": reps, "windows": windows, "warmup": n_warm, "ms_per_step": round(ms, 4), "ms_per_step_min": round(min(win), 4), "ms_per_step_max": round(max(win), 4),
           "value": round(16 * H * W / ms / 1e3, 1), "unit": "output MPix/s", "model_tflops": round(fl / ms / 1e9, 1),
           "frac_of_mfma_peak": round(fl / ms / 1e9 / PEAK_F16_TFLOPS, 4)}
    timed_forward(net, x)
    r = roofline_from_launches(timed_forward(net, x))
    out["hip_event_sum_ms"] = round(sum(e["ms_total"] for e in r["per_kernel"].values()), 4)
    out["dominant_kernel"] = {k: r[k] for k in ("kernel", "launches", "avg_launch_ms", "tflops", "gbs", "frac_mfma", "frac_hbm", "binding_roof", "frac_binding")}
    out["per_kernel"] = r["per_kernel"]
    net.release_workspace()
    return out


def unet_per_kernel(net, x):
    """Per-kernel two-roof entries of one UNet forward from the library's launch timer (innfer_timer_*), or None when the library has none."""
    import innfer_amd.lib as L
    if not hasattr(L, "timed_launches"):
        return None
    launches = L.timed_launches(lambda: net(x))
    agg = {}
    for name, ms, fl, by in launches:
        a = agg.setdefault(name, [0.0, 0.0, 0.0, 0])
        a[0] += ms; a[1] += fl; a[2] += by; a[3] += 1
    per = {}
    for name, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        e = {"launches": v[3], "ms_total": round(v[0], 4)}
        if v[1] > 0 and v[2] > 0:
            e.update(two_roofs(v[1], v[2], v[0]))
        elif v[2] > 0:
            e.update({"gbs": round(v[2] / (v[0] * 1e-3) / 1e9, 1), "frac_hbm": round(v[2] / (PEAK_HBM_GBS * 1e9) / (v[0] * 1e-3), 4), "bound": "hbm"})
        per[name] = e
    return per


# ----------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: the parent starts one fresh process per rank BEFORE anything touches the GPU
# (never re-exec a process that has initialised HIP), relays rank 0's JSON line and fails if a rank fails.
def visible_gpu_count():
    """GPUs this process would see, WITHOUT any HIP / torch.cuda call in the parent that spawns the ranks: GPU nodes of the KFD topology in
    sysfs (simd_count > 0), cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES.  Falls back to
    torch.cuda.device_count() (which does not create a context on this image) when sysfs has no topology."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(l.split() for l in open(f).read().splitlines() if len(l.split()) == 2)
            n += int(props.get("simd_count", "0")) > 0
        except Exception:
            pass
    if n == 0:
        import torch
        return torch.cuda.device_count()
    # a container / lease that exposes a subset of the node's GPUs still shows the whole topology in sysfs: a GPU is usable only through its
    # DRM render node, so the count is cut down to the render nodes this process may open
    nodes = glob.glob("/dev/dri/renderD*")
    usable = sum(os.access(d, os.R_OK | os.W_OK) for d in nodes)
    if usable:                                      # (none accessible at all: not a statement about GPUs -- leave the topology count alone)
        n = min(n, usable)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def spawn_ranks(args, argv):
    import socket
    n = args.gpus
    dry = os.environ.get("INNFER_BENCH_DRYRUN") == "1" or os.environ.get("INNFER_BENCH_SELFTEST") == "1"
    if not dry:
        have = visible_gpu_count()
        if have < n:
            log(f"--gpus {n} but only {have} GPU(s) visible (INNFER_BENCH_DRYRUN=1 rehearses the N > 1 control flow on one GPU)")
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")            # rank 0's stdout (the JSON line); the other ranks' stdout goes to stderr
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else sys.stderr))
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):      # a rank died: the others would wait in a collective for ever
            time.sleep(2.0)
            for i, p in enumerate(procs):
                if p.poll() is None:
                    p.kill()                            # exactly the PIDs started above
                    rcs[i] = p.wait()
                else:
                    rcs[i] = p.returncode
            break
        time.sleep(0.1)
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    if any(rcs):
        log(f"rank exit codes {rcs}")
        return 1
    return 0


def timed_steps(step, steps, warmup, world, sync, barrier, max_over_ranks):
    """The contract's timed region: W untimed steps, then exactly K steps bracketed by barrier + synchronise, MAX over ranks."""
    for _ in range(warmup):
        step()
    sync()
    if world > 1:
        barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        y = step()
    sync()
    if world > 1:
        barrier()
    sync()
    wall = time.perf_counter() - t0
    del y
    return max_over_ranks(wall)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="frame1080", choices=WORKLOADS)
    ap.add_argument("--band-rows", type=int, default=0)
    ap.add_argument("--tile-batch", type=int, default=0, help="chop workloads: tiles per network launch (0 = innfer_amd.parallel.tile_batches: evenly sized launches of <= 272 tiles)")
    ap.add_argument("--sharded-steps", type=int, default=-1, help="timed passes of the tile_sharded (chain4k) measurement; 0 = skip; default 2 on one GPU, 5 on N > 1 "
                    "(the strong-scaling table of the first multi-GPU run comes from these objects: VERDICT r4 item 6a)")
    ap.add_argument("--fp32", action="store_true", help="frame workloads: a float32 frame = the fp32-accurate engine (the reference's -no_fp16 mode); no roofline object")
    ap.add_argument("--no-hr-chain", action="store_true", help="A/B: the last up-conv and HR_conv0 + conv_last as two launches (innfer_net_set_hr_chain 0)")
    ap.add_argument("--no-fused-tail", action="store_true", help="A/B: HR_conv0 and conv_last as two launches (innfer_net_set_fused_tail 0)")
    ap.add_argument("--no-upconv-phases", action="store_true", help="A/B: the up-convs as nine taps on the HR grid (innfer_net_set_upconv_phases 0)")
    ap.add_argument("--upconv-phase-visits", action="store_true", help="A/B: the phase up-convs with one phase per visit of a tile (innfer_net_set_upconv_phases 2; default: all four phases in one visit)")
    ap.add_argument("--residual-lds", type=int, default=1, choices=[0, 1, 2], help="A/B: innfer_net_set_residual_lds -- the dense block's residual from the conv's own LDS stages: 0 never, 1 the RRDB-end blocks (default), 2 every block")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the chop8k (BASELINE config 3) and unet64 (config 5) objects of the line")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-power-probe", action="store_true")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        return cpu_baseline_child()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args, sys.argv[1:])

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.sharded_steps < 0:
        args.sharded_steps = 5 if world > 1 else 2
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"--gpus {args.gpus} but the launcher started {world} rank(s): reporting n_gpus = {world}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # INNFER_BENCH_SELFTEST=1: the N-rank control flow only (spawn, rendezvous, barriers, max over ranks, ONE line from rank 0)
    #   with a sleep as the "step": runs without a GPU, covered by tests/test_bench_ranks_cpu.py.  Never a measurement.
    # INNFER_BENCH_DRYRUN=1: the real HIP workload with every rank on cuda:0 and a gloo rendezvous -- a rehearsal of the
    #   N > 1 path on a ONE-GPU box.  Never a measurement either.
    selftest = os.environ.get("INNFER_BENCH_SELFTEST") == "1"
    dryrun = world > 1 and os.environ.get("INNFER_BENCH_DRYRUN") == "1"
    if world > 1:
        if selftest or dryrun:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def barrier():
        dist.barrier()

    def max_over_ranks(v):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device="cpu" if (selftest or dryrun) else torch.device("cuda", local_rank))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()

    if selftest:
        wall = timed_steps(lambda: time.sleep(0.002), args.steps, args.warmup, world, lambda: None, barrier, max_over_ranks)
        seen = torch.tensor([1.0])
        if world > 1:
            dist.all_reduce(seen)
        if rank == 0:
            print(json.dumps({"metric": "selftest", "value": round(world * args.steps / wall, 2), "unit": "steps/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall * 1e3 / args.steps, 3),
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none",
                              "data": "selftest: no GPU work, control flow only", "ranks_seen": int(seen.item()),
                              "config": {"workload": "selftest"}}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def sync():
        torch.cuda.synchronize()

    from innfer_amd import parallel, synth
    net, _ = build_net(dev)
    net.band_rows = args.band_rows
    net.fused_tail = not args.no_fused_tail
    net.hr_chain = not args.no_hr_chain
    net.upconv_phases = 0 if args.no_upconv_phases else (2 if args.upconv_phase_visits else 1)
    net.residual_lds = args.residual_lds
    tag = " (DRY RUN: all ranks on one GPU, gloo)" if dryrun else ""

    def chop_setup(workload, profile=False):
        """(step, H, W, description, runners) of a tile-sharded workload over the current ranks."""
        H, W = (4320, 7680) if workload == "chop8k" else (2160, 3840)
        x = torch.from_numpy(synth.uniform((1, 3, H, W), 2)).to(dev).half()
        r4 = parallel.ChopRunner(net, scale=4, tile_batch=args.tile_batch or None, profile=profile)
        if workload == "chain4k":                   # BASELINE config 4: model chain 1x + 4x (run.py:424-426)
            net1, _ = build_net(dev, scale=1)
            r1 = parallel.ChopRunner(net1, scale=1, tile_batch=args.tile_batch or None, profile=profile)
            return (lambda: parallel.run_chain([r1, r4], x)), H, W, "model chain RRDBNet-23 1x + RRDBNet-23 4x", [r1, r4]
        return (lambda: r4(x)), H, W, "ESRGAN RRDBNet-23 4x", [r4]

    def chop_flops(runners, H, W):
        """Algorithmic FLOPs of one pass: every tile of every stage counts (the reference's tiling is part of the contract)."""
        import innfer_amd.lib as L
        ps, ys, xs = L.chop_plan(H, W, 200, 0.5)
        return sum(r.model_fn.flops(len(ys) * len(xs), ps, ps) for r in runners), len(ys) * len(xs)

    frame = args.workload.startswith("frame")
    if frame:
        H, W = (1080, 1920) if args.workload == "frame1080" else (540, 960)
        x = torch.from_numpy(synth.uniform((1, 3, H, W), 2 + rank)).to(dev)
        x = x.float() if args.fp32 else x.half()
        out_pix = 16 * H * W * world                # every rank produces its own frame

        def step():
            return net(x)
        cfg = {"workload": f"ESRGAN RRDBNet-23 4x {'fp32-accurate (fp16 hi/lo pairs, -no_fp16)' if args.fp32 else 'fp16'}, 1x3x{H}x{W} -> 1x3x{4 * H}x{4 * W}, un-tiled "
                           f"(Model(chop=False)), one frame per rank", "band_rows": args.band_rows,
               "parallelism": (f"frame replicas x{world}" if world > 1 else "single GPU") + tag}
        scaling = "weak"
        metric = "output MPix/s, 4x ESRGAN RRDB-23 1080p->4K"
    else:
        step, H, W, what, main_runners = chop_setup(args.workload)
        out_pix = 16 * H * W                        # unique output pixels of the ONE shared frame
        cfg = {"workload": f"{what} fp16, {H}x{W} input through chop_forward (patch 200, step 0.5), "
                           f"tile list sharded over {world} rank(s), raw HR tiles sent to rank 0 and blended there",
               "band_rows": args.band_rows, "parallelism": f"tile-dp{world}" + tag}
        scaling = "strong"
        metric = {"chop8k": "output MPix/s, 4x ESRGAN RRDB-23, 8K input through chop_forward",
                  "chop4k": "output MPix/s, 4x ESRGAN RRDB-23, 4K input through chop_forward",
                  "chain4k": "output MPix/s, model chain 1x + 4x ESRGAN RRDB-23, 4K input through chop_forward"}[args.workload]

    log('warmup + timed region')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n_done = [0]

    def step_ev():                                  # HIP events around the K timed steps only (torch's current stream = the launch stream)
        if n_done[0] == args.warmup:
            e0.record()
        n_done[0] += 1
        return step()
    wall = timed_steps(step_ev, args.steps, args.warmup, world, sync, barrier, max_over_ranks)
    e1.record()
    sync()
    log(f'timed {args.steps} steps in {wall:.3f} s')
    ev_ms = e0.elapsed_time(e1)

    line = None
    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        line = {"metric": metric, "value": round(out_pix * args.steps / wall / 1e6, 2),
                "unit": "MPix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": scaling,
                "vs_baseline": None, "dtype": "f32 as fp16 (hi, lo) pairs" if (frame and args.fp32) else "f16", "data": "synthetic", "config": cfg,
                "hip_event_ms_per_step": round(ev_ms / args.steps, 3)}
        if not frame:
            flops, _ = chop_flops(main_runners, H, W)
            line["model_tflops"] = round(flops * args.steps / wall / 1e12, 2)
            line["frac_of_mfma_peak_all_gpus"] = round(flops * args.steps / wall / 1e12 / (PEAK_F16_TFLOPS * world), 4)
        if frame:
            flops = net.flops(1, H, W)
            line["model_tflops"] = round(flops * world * args.steps / wall / 1e12, 2)
            line["frac_of_mfma_peak"] = round(flops * args.steps / wall / 1e12 / PEAK_F16_TFLOPS, 4)
            if not args.no_roofline and not args.fp32:
                log('per-launch timing')
                timed_forward(net, x)
                launches = timed_forward(net, x)
                line["roofline"] = roofline_from_launches(launches)
                if world == 1 and not args.no_power_probe:
                    log('power probe')
                    pw = power_probe(step)
                    if pw:
                        pw["frac_of_peak_at_sclk"] = round(line["roofline"]["tflops"] / pw["peak_at_sclk_tflops"], 4)
                        line["roofline"]["power"] = pw
                        # a box-normalised companion of `value`: the boxes of the pool hold different shader clocks under the same 1400 W cap (1.62 ..
                        # 1.76 GHz seen; +-3 % of frame time), more than most kernel changes move -- the same step rescaled to a 1700 MHz box.  A comparison aid, never `value`.
                        line["box_normalised"] = {"sclk_mhz": pw["sclk_mhz"], "ms_per_step_at_1700mhz": round(line["ms_per_step"] * pw["sclk_mhz"] / 1700.0, 3),
                                                  "value_at_1700mhz": round(line["value"] * 1700.0 / pw["sclk_mhz"], 2),
                                                  "note": "ms_per_step x sclk / 1700 MHz (the frame is matrix-pipe / power bound: time scales with 1 / sclk); sclk = mean of the rocm-smi samples of the power probe"}
                log("per-layer classes:\n" + per_layer_table(launches))
    if frame:
        del x
    net.release_workspace()
    torch.cuda.empty_cache()

    def gather_phases(phases):
        """[rank][stage] -> {tiles, compute_ms, exchange_ms, blend_ms, bcast_ms, ...} of the profiled pass, from every rank (rank 0 keeps the list)."""
        mine = [{k: (round(v, 2) if isinstance(v, float) else v) for k, v in p.items()} for p in phases]
        if world == 1:
            return [mine]
        box = [None] * world
        dist.all_gather_object(box, mine)
        return box

    # ---- BASELINE config 4 on the same ranks: chain 1x + 4x, 4K input, tile list sharded, RCCL exchange ----
    if args.sharded_steps > 0 and args.workload != "chain4k":
        log('tile-sharded chain4k')
        cstep, cH, cW, cwhat, runners = chop_setup("chain4k", profile=False)
        cwall = timed_steps(cstep, args.sharded_steps, 1, world, sync, barrier, max_over_ranks)
        for r in runners:
            r.profile = True                        # one more pass with a synchronise around every phase: where the time goes
        cstep()
        sync()
        phases = [dict(r.last) for r in runners]
        all_phases = gather_phases(phases)
        if world > 1:
            barrier()
        if rank == 0:
            ms = cwall * 1e3 / args.sharded_steps
            cfl, ntile = chop_flops(runners, cH, cW)            # 1060 + 1144 TFLOP per frame
            tf = cfl / 1e12
            line["tile_sharded"] = {
                "workload": f"{cwhat} fp16, {cH}x{cW} input, {ntile} tiles of 200^2 per stage (BASELINE config 4), tile list sharded "
                            f"over {world} rank(s), raw HR tiles to rank 0, blend there, intermediate broadcast" + tag,
                "parallelism": f"tile-dp{world}", "scaling": "strong", "steps": args.sharded_steps, "warmup": 1,
                "ms_per_frame": round(ms, 2), "value": round(16 * cH * cW / (ms * 1e-3) / 1e6, 2), "unit": "MPix/s",
                "model_tflops": round(tf / (ms * 1e-3), 1),
                "frac_of_mfma_peak_all_gpus": round(tf / (ms * 1e-3) / (PEAK_F16_TFLOPS * world), 4),
                "rank0_phases_ms": [{k: (round(v, 2) if isinstance(v, float) else v) for k, v in p.items()} for p in phases],
                "per_rank_phases_ms": all_phases,
                "exchange_bytes_into_rank0": sum(p.get("exchange_bytes", 0) for p in phases),
                "exchange_ms": round(sum(p.get("exchange_ms", 0.0) for p in phases), 2)}
            # the prediction this (or the first multi-GPU) run is checked against: from THIS rank's per-tile cost and blend time (predict_tile_scaling)
            for p_, sc_ in zip(phases, (1, 4)):
                p_["out_h"], p_["out_w"] = cH * sc_, cW * sc_
            pred = predict_tile_scaling(stage_model(phases, 200, (1, 4)))
            line["tile_sharded"]["predicted"] = pred
            line["tile_sharded"]["predicted_ms"] = next((r["ms_per_frame"] for r in pred if r["n_gpus"] == world), None)
            line["tile_sharded"]["prediction_model"] = (f"ceil(tiles/N) x this rank's ms per tile + one peer's raw HR tiles over one xGMI link ({XGMI_LINK_GBS:.0f} GB/s x "
                                                        f"{XGMI_PAYLOAD_EFF}) + rank 0's blend + the intermediate's broadcast (one link time); DESIGN.md section 8")
        del cstep, runners
        torch.cuda.empty_cache()

    # ---- BASELINE config 3 on the same ranks: 8K input through chop_forward (3268 tiles), one warm-up + one timed pass ----
    def chop8k_object():
        ksteps = 5 if world > 1 else 1
        kstep, kH, kW, kwhat, krunners = chop_setup("chop8k")
        kwall = timed_steps(kstep, ksteps, 1, world, sync, barrier, max_over_ranks) / ksteps
        # one more pass with a synchronise around every phase, every rank reporting (never timed)
        for r in krunners:
            r.profile = True
        kstep()
        sync()
        kmine = [dict(r.last) for r in krunners]
        kphases = gather_phases(kmine)
        if world > 1:
            barrier()
        out = None
        if rank == 0:
            kfl, ntile = chop_flops(krunners, kH, kW)           # 4686.8 TFLOP (the redundant tile FLOPs of the reference's tiling count)
            out = {
                "workload": f"{kwhat} fp16, {kH}x{kW} input through chop_forward, {ntile} tiles of 200^2 (BASELINE config 3), tile list sharded over "
                            f"{world} rank(s), blend on rank 0" + tag,
                "steps": ksteps, "warmup": 1, "per_rank_phases_ms": kphases, "s_per_frame": round(kwall, 3), "value": round(16 * kH * kW / kwall / 1e6, 2), "unit": "unique-output MPix/s",
                "model_tflops": round(kfl / kwall / 1e12, 1), "frac_of_mfma_peak_all_gpus": round(kfl / kwall / 1e12 / (PEAK_F16_TFLOPS * world), 4),
                "tile_batches": parallel.tile_batches(parallel.shard_tiles(ntile, world, 0)[1], args.tile_batch or None,
                                                      parallel.engine_tile_cap(net, 200, torch.float16, dev))}
            kpred = predict_tile_scaling(stage_model(kmine, 200, (4,)))
            out["predicted"] = kpred
            out["predicted_ms"] = next((r["ms_per_frame"] for r in kpred if r["n_gpus"] == world), None)
        del kstep, krunners
        return out

    if not args.no_extras and args.workload != "chop8k":
        log('chop8k')
        if world == 1:
            try:
                line["chop8k"] = chop8k_object()
            except Exception as e:                  # a side object must never cost the headline line (one rank: nobody else waits in a collective)
                line["chop8k"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        else:                                       # every rank runs the same collective sequence: an exception is fatal for the job on purpose
            obj = chop8k_object()
            if rank == 0:
                line["chop8k"] = obj
        net.release_workspace()
        torch.cuda.empty_cache()

    # ---- BASELINE config 5: pix2pix UNet_256, 64 x 3 x 256 x 256 (one GPU) ----
    if not args.no_extras and world == 1:
        log('unet64')
        try:
            line["unet64"] = unet64_object(dev)
        except Exception as e:                      # a side object must never cost the headline line
            line["unet64"] = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- north_star's third stack: PAN 4x on a 540 x 960 frame (one GPU) ----
    if not args.no_extras and world == 1:
        log('pan540')
        try:
            line["pan540"] = pan540_object(dev)
            line["pan_chop1080"] = line["pan540"].pop("chop1080", None)
            line["pan540_fp32"] = line["pan540"].pop("fp32", None)
        except Exception as e:                      # a side object must never cost the headline line
            line["pan540"] = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- north_star's second stack: SRResNet-16 4x on the 1080p frame (one GPU) ----
    if not args.no_extras and world == 1:
        log('srresnet1080')
        try:
            line["srresnet1080"] = srresnet1080_object(dev)
        except Exception as e:                      # a side object must never cost the headline line
            line["srresnet1080"] = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        if not args.no_cpu_baseline:
            log('cpu baseline')
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
