"""Host -> HBM rate of engine.upload_packed on an 8 GB synthetic .bed body: the host-side copy into the pinned staging pieces
with 1 / 2 / 4 / 8 / 16 threads, rows staged with and without the device pitch (one contiguous copy per piece against a 2-D
copy), 2 or 3 staging pieces; then the default form followed by the in-place recode, and with the recode of every piece
overlapped with the transfer of the next (``recode="value"``, what models.py does)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrhe_amd import engine as E
dev = torch.device("cuda:0")
N, m = 500_000, 64_000
rb = (N + 3) // 4
host = np.random.default_rng(0).integers(0, 256, size=(m, rb), dtype=np.uint8)
print("cpus available:", len(os.sched_getaffinity(0)))
ref = E.upload_packed(host, dev, threads=1, n_stage=2, pitched=False)            # (also warms the pinned allocator)
for threads, n_stage, pitched in [(1, 2, False), (1, 2, True), (2, 3, True), (4, 3, True), (8, 3, True), (8, 3, False), (8, 2, True), (16, 3, True), (16, 4, True)]:
    t0 = time.perf_counter(); g = E.upload_packed(host, dev, threads=threads, n_stage=n_stage, pitched=pitched); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"threads {threads:2d} staging pieces {n_stage} pitched {int(pitched)}: {t1-t0:.2f} s = {host.nbytes/1e9/(t1-t0):.1f} GB/s, same bytes: {bool(torch.equal(ref, g))}")
    del g
for rep in range(2):
    t0 = time.perf_counter(); g = E.upload_packed(host, dev); torch.cuda.synchronize(); t1 = time.perf_counter()
    E.recode_rows(g, "value"); torch.cuda.synchronize(); t2 = time.perf_counter()
    t3 = time.perf_counter(); g2 = E.upload_packed(host, dev, recode="value"); torch.cuda.synchronize(); t4 = time.perf_counter()
    same = bool(torch.equal(g, g2))
    print(f"default: upload {host.nbytes/1e9:.1f} GB: {t1-t0:.2f} s = {host.nbytes/1e9/(t1-t0):.1f} GB/s;  recode afterwards {(t2-t1)*1e3:.1f} ms "
          f"({g.numel()/1e9/(t2-t1):.0f} GB/s in place);  upload with the recode overlapped {t4-t3:.2f} s = {host.nbytes/1e9/(t4-t3):.1f} GB/s, same bytes: {same}")
    del g, g2
# a memory-mapped file (page cache warm after the first read): what models.py hands over
path = "/tmp/upload_rate.bed"
host[: m // 4].tofile(path)
mm = np.memmap(path, dtype=np.uint8, mode="r", shape=(m // 4, rb))
for rep in range(2):
    t0 = time.perf_counter(); g = E.upload_packed(mm, dev, recode="value"); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"memory-mapped file, {mm.nbytes/1e9:.1f} GB: {t1-t0:.2f} s = {mm.nbytes/1e9/(t1-t0):.1f} GB/s")
    del g
os.remove(path)
