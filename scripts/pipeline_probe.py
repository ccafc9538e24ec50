"""where does the input pipeline's time go? host batching / H2D / device kernel, timed separately"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dpot_amd.data import DeviceBatcher, resize_pad_window
B, Traw = 32, 20
pool = [np.random.rand(64, 64, Traw, 1).astype(np.float32) for _ in range(64)]
db = DeviceBatcher(B, 128, 10, 1, 4, max_raw_floats_per_sample=64 * 64 * Traw)
samples = pool[:B]; starts = [3] * B
for _ in range(3):
    db.submit(samples, starts); db.get(); db.release()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    db.submit(samples, starts); db.get(); db.release()
t_host = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 20
devs = [torch.from_numpy(s).cuda() for s in samples]
for _ in range(3):
    resize_pad_window(devs, starts, 128, 10, 1, 4)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    resize_pad_window(devs, starts, 128, 10, 1, 4)
e1.record(); e1.synchronize()
print(f"submit+get host time {t_host*1e3:.3f} ms/batch, incl. device drain {t_all*1e3:.3f} ms/batch, "
      f"device kernel {e0.elapsed_time(e1)/20*1e3:.1f} us/batch (B={B}, 64x64x{Traw}x1 -> 128x128x11x4)")
