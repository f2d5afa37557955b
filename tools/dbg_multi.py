import sys, runpy
sys.path.insert(0, ".")
from pyrhe_amd import engine
orig = engine.SketchEngine.multi_ok
def dbg(x, y_sub, y_flat, period):
    ok = orig(x, y_sub, y_flat, period)
    def blk_ok(r):
        return (r.tot is None, r.blk is not None, None if r.blk is None else r.blk.data_ptr() % 16, r.stride % 4, None if r.mul is None else r.mul.data_ptr() % 16)
    print("[multi_ok]", ok, "nx", len(x), "ny_sub", len(y_sub), "ny_flat", len(y_flat), "period%32", period % 32, [blk_ok(r) for r in x][:3], file=sys.stderr)
    return ok
engine.SketchEngine.multi_ok = staticmethod(dbg)
sys.argv = ["bench.py", "--dom", "--n-bin", "8", "--n-cov", "10", "--n-indv", "200000", "--n-snp", "40000", "--jn", "4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-alt", "--no-cross-check"]
runpy.run_path("bench.py", run_name="__main__")
