"""What the chip sustains on a plain fp16 GEMM (hipBLASLt via torch.matmul) under its 1400 W cap: the practical MFMA ceiling
the conv kernels are compared with in DESIGN.md section 4 (library call used as a yardstick only, not on the product path)."""
import subprocess, threading, time, torch
dev = torch.device("cuda:0")
def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = [l for l in out.splitlines() if "sclk" in l]
    pw = [l for l in out.splitlines() if "Package Power" in l]
    return (sclk[0].split("(")[-1].rstrip(")") if sclk else "?"), (pw[0].split(":")[-1].strip() if pw else "?")
for (m, n, k) in [(8192, 8192, 8192), (16384, 8192, 8192), (32768, 4096, 1728)]:
    a = torch.randn(m, k, device=dev, dtype=torch.float16) * 0.05
    b = torch.randn(k, n, device=dev, dtype=torch.float16) * 0.05
    for _ in range(5): torch.matmul(a, b)
    torch.cuda.synchronize()
    reps = max(20, int(6.0 / (2.0 * m * n * k / 1.2e15)))        # ~6 s so the power manager settles
    t0 = time.time(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    box = []
    th = threading.Timer(3.0, lambda: box.append(smi()))        # sampled mid-run (the enqueue loop blocks on the queue depth)
    th.start()
    e0.record()
    for i in range(reps): torch.matmul(a, b)
    e1.record()
    torch.cuda.synchronize(); th.join(); s = box[0]
    ms = e0.elapsed_time(e1) / reps
    print(f"matmul f16 {m}x{n}x{k}: {ms*1e3:8.1f} us  {2.0*m*n*k/ms/1e9:7.1f} TFLOP/s  sclk {s[0]}  power {s[1]} W  ({reps} reps)", flush=True)
