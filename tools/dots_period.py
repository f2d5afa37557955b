"""pyrhe_engine_loo_dots_multi (pass 2's one-read kernel) at a fixed number of store bytes and different ROW LENGTHS: the same kernel,
the same matrix work per byte -- only the distance between the rows a wave reads changes (a row of n_pad floats is its own 2-MiB
page at N = 500k).  Round 4: is the kernel bound by address translation / DRAM pages rather than by bytes or matrix work?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrhe_amd import engine
from pyrhe_amd.engine import Row
dev = torch.device("cuda:0")
nx, n_sub, nys, nyf = 8, 10, 9, 10
total_bytes = 4 << 30
for period in (65536, 131072, 262144, 500736, 1001472):
    batch = max(2, total_bytes // (nx * n_sub * period * 4))
    eng = engine.SketchEngine(period, period, max_entries=32, device=dev)
    store = torch.randn((batch, nx, n_sub * period), dtype=torch.float32, device=dev)
    ys = torch.randn((nys, n_sub * period), dtype=torch.float64, device=dev)
    yf = torch.randn((nyf, period), dtype=torch.float64, device=dev)
    x = [Row(None, store[0, a], None, store.stride(0)) for a in range(nx)]
    y_sub, y_flat = [Row(ys[b]) for b in range(nys)], [Row(yf[m]) for m in range(nyf)]
    for f32 in (1, 0):
        eng.set_option("dots_f32", f32)
        for _ in range(2):
            eng.loo_dots_multi(x, y_sub, y_flat, True, period, n_sub, batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.loo_dots_multi(x, y_sub, y_flat, True, period, n_sub, batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"period {period:8d} ({period * 4 / 2**20:5.2f} MiB rows) batch {batch:4d} f32 {f32}: {dt * 1e3:7.2f} ms  {store.numel() * 4 / dt / 1e12:5.2f} TB/s")
    eng.close()
    del store, ys, yf
