import torch, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from pyrhe_amd import engine
dev = torch.device("cuda:0")
n, m, k = 20000, 3000, 10
packed = engine.recode_rows(engine.synth_genotypes(m, n, 1, 0, dev), "value")
ent = torch.arange(m, dtype=torch.int32, device=dev)
free0 = None
for it in range(60):
    eng = engine.SketchEngine(n, n, max_entries=m, precision="i8x3", device=dev)
    eng.set_coding("value")
    keep = eng.keep_vector([])
    Z = torch.randn((k, eng.n_pad), device=dev); y = torch.randn((1, eng.n_pad), device=dev)
    eng.set_rhs(Z, y, keep)
    sk, nm, _ = eng.sketch_block(packed, ent, [0, 1000, 2000, m])
    a, na, d, nd = eng.sketch_block_dom(packed, ent, [0, m])
    from pyrhe_amd.engine import Row
    x = [Row(None, sk[0, c], None, sk.stride(0)) for c in range(4)]
    ys = [Row(torch.randn(eng.n_pad, dtype=torch.float64, device=dev)) for _ in range(3)]
    eng.loo_dots_multi(x, ys, ys[:2], True, eng.n_pad, 1, 3)
    torch.cuda.synchronize()
    eng.close()
    del eng, sk, nm, a, na, d, nd, Z, y, keep, x, ys
    torch.cuda.empty_cache()
    free = torch.cuda.mem_get_info()[0]
    if it == 5: free0 = free
print("free after 6 iterations:", free0, " after 60:", free, " delta MB:", (free0 - free) / 1e6)
