"""Host -> HBM rate of engine.upload_packed (two pinned 256 MiB staging pieces) on an 8 GB synthetic .bed body: plain, followed by
the in-place recode, and with the recode of every piece overlapped with the transfer of the next (``recode="value"``, what
models.py does)."""
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
    ref = g
    t3 = time.perf_counter(); g2 = E.upload_packed(host, dev, recode="value"); torch.cuda.synchronize(); t4 = time.perf_counter()
    same = bool(torch.equal(ref, g2))
    print(f"upload {host.nbytes/1e9:.1f} GB: {t1-t0:.2f} s = {host.nbytes/1e9/(t1-t0):.1f} GB/s;  recode afterwards {(t2-t1)*1e3:.1f} ms "
          f"({g.numel()/1e9/(t2-t1):.0f} GB/s in place);  upload with the recode overlapped {t4-t3:.2f} s = {host.nbytes/1e9/(t4-t3):.1f} GB/s, same bytes: {same}")
    del g, g2, ref
