"""Host -> HBM rate of engine.upload_packed (pinned 1 GiB staging pieces) + the in-place recode, on a 8 GB synthetic .bed body."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrhe_amd import engine as E
dev = torch.device("cuda:0")
N, m = 500_000, 64_000
rb = (N + 3) // 4
host = np.random.default_rng(0).integers(0, 256, size=(m, rb), dtype=np.uint8)
for rep in range(2):
    t0 = time.perf_counter(); g = E.upload_packed(host, dev); torch.cuda.synchronize(); t1 = time.perf_counter()
    E.recode_rows(g, "value"); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"upload {host.nbytes/1e9:.1f} GB: {t1-t0:.2f} s = {host.nbytes/1e9/(t1-t0):.1f} GB/s;  recode {g.numel()/1e9/(t2-t1):.0f} GB/s in place")
    del g
