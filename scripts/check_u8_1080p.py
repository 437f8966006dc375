import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from innfer_amd import run as R, synth
from innfer_amd.utils import utils as U
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(synth.rrdbnet_shapes(nb=23, scale=4), 0).items()}
torch.save(sd, '/tmp/4x_big.pth')
for chop in (True, False):
    m = R.Model('/tmp/4x_big.pth', 'infer', 4, chop=chop)
    img = synth.image_u8(1080, 1920, 3, 5)
    got = m.run_u8(img)
    want = U.tensor2np(m(U.np2tensor(img, dtype=torch.float16)))
    torch.cuda.synchronize()
    t0 = time.perf_counter(); got = m.run_u8(img); t1 = time.perf_counter()
    want = U.tensor2np(m(U.np2tensor(img, dtype=torch.float16))); t2 = time.perf_counter()
    print('chop' if chop else 'frame', got.shape, 'equal', np.array_equal(got, want), f'fused {1e3*(t1-t0):.1f} ms  separate {1e3*(t2-t1):.1f} ms (host numpy in, host numpy out)')
