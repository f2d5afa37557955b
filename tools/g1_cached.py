"""GEMM 1 / GEMM 2 with HBM taken out: the same 10 016-entry launch, but every entry points into 32 (L2-resident) or
512 (Infinity-Cache-resident) distinct rows -- how fast is the instruction stream + LDS + barrier structure alone?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrhe_amd import engine as E
dev = torch.device("cuda:0")
N, m, k = 500_000, 10016, 10
geno = E.synth_genotypes(m, N, 1, 0, dev)
for prec in ("i8x3", "f16"):
    eng = E.SketchEngine(N, N, max_entries=m, precision=prec, device=dev)
    keep = eng.keep_vector([])
    Z = torch.zeros((k, eng.n_pad), device=dev); Z[:, :N] = torch.randn((k, N), device=dev)
    y = torch.zeros((1, eng.n_pad), device=dev); y[:, :N] = torch.randn((1, N), device=dev)
    eng.set_rhs(Z, y, keep)
    for distinct in (m, 512, 32):
        ent = (torch.arange(m, device=dev, dtype=torch.int32) % distinct).contiguous()
        sk = torch.empty((1, k, eng.n_pad), device=dev); nm = torch.empty((1, 1), dtype=torch.float64, device=dev)
        for _ in range(3):
            eng.sketch_block(geno, ent, [0, m], out_sketch=sk, out_norm=nm)
        torch.cuda.synchronize()
        eng.profile(True)
        for _ in range(10):
            eng.sketch_block(geno, ent, [0, m], out_sketch=sk, out_norm=nm)
        torch.cuda.synchronize()
        p = eng.profile_read(); eng.profile(False)
        print(f"{prec} distinct rows {distinct:6d}: gemm1 {p['gemm1_ms']/10*1e3:7.1f} us  gemm2 {p['gemm2_ms']/10*1e3:7.1f} us  mid {p['mid_ms']/10*1e3:6.1f} us")
    eng.close()
