#!/usr/bin/env python3
"""gpurun_out/<tag>_final (scratch) -> profiles/<tag>_* (tracked): the summaries of one round that the judge reads.  Run from the
repo root after tools/collect.sh <tag>:   python3 tools/collect_copy.py r04 <commit the GPU runs were made at>"""
import csv, glob, json, os, re, shutil, sys

rnd = sys.argv[1]
commit = sys.argv[2] if len(sys.argv) > 2 else "HEAD"
src, dst = f"gpurun_out/{rnd}_final", "profiles"

# 1. the default bench line (plain run) and the same command under rocprofv3 --kernel-trace --stats
shutil.copy(f"{src}/bench_plain.json", f"{dst}/{rnd}_bench.json")
shutil.copy(f"{src}/bench_trace.json", f"{dst}/{rnd}_bench_traced.json")
shutil.copy(glob.glob(f"{src}/trace/**/*kernel_stats.csv", recursive=True)[0], f"{dst}/{rnd}_kernel_stats.csv")
for tag in ("c2", "c4", "c5", "dom"):
    shutil.copy(glob.glob(f"{src}/prof_{tag}/**/*kernel_stats.csv", recursive=True)[0], f"{dst}/{rnd}_{tag}_kernel_stats.csv")
    shutil.copy(f"{src}/prof_{tag}/bench.json", f"{dst}/{rnd}_{tag}_bench_traced.json")
if os.path.exists(f"{src}/dom_c4.json"):
    shutil.copy(f"{src}/dom_c4.json", f"{dst}/{rnd}_dom_8bin_cov_bench.json")

# 2. PMC summaries (one section per separate --pmc pass) and the traffic ratios bench.py quotes, per shape
sections = [s for s in ("fetch", "write", "sq1", "c4_fetch", "c4_write", "c4_sq1", "c5_fetch", "c5_write") if os.path.exists(f"{src}/pmc_{s}.summary.txt")]
with open(f"{dst}/{rnd}_pmc_summary.txt", "w") as out:
    out.write(f"# rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --n-snp 100000 --jn 10 --steps 1 --warmup 1 [config] at {commit}\n"
              "# one separate run per section (tools/collect.sh); mean per dispatch (tools/pmc_summary.py); c4 = --n-bin 8 --n-cov 10, c5 = --genie G+GxE\n")
    for sec in sections:
        out.write(f"== pmc_{sec}\n" + open(f"{src}/pmc_{sec}.summary.txt").read())
N, SNPS, PASSES = 500_000, 100_000, 2
alg = SNPS * ((N + 3) // 4) * PASSES


def counters(path, counter):
    """{short kernel name: (launches, mean)} of one pmc summary file"""
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            m = re.search(r"(gemm[12]\w*_kernel)<(\d+)", line)
            cur = f"{m.group(1)}<{m.group(2)}>" if m else None
        elif cur:
            m = re.search(counter + r"\s+n=\s*(\d+)\s+mean=([0-9.e+]+)", line)
            if m:
                n0, v0 = out.get(cur, (0, 0.0))            # instantiations that differ in later template arguments: pooled
                n1, v1 = int(m.group(1)), float(m.group(2))
                out[cur] = (n0 + n1, (n0 * v0 + n1 * v1) / (n0 + n1))
    return out


traffic = {"_comment": "HBM bytes per algorithmic byte of the GEMM kernels of three configurations: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 B summed "
                       "over the launches of the PMC run (MI355X_MICROARCH.md: FETCH_SIZE counts half of a wide streaming read on gfx950), divided by "
                       f"the packed genotype bytes those launches covered; separate --pmc runs, profiles/{rnd}_pmc_summary.txt",
           "commit": commit, "n_indv": N, "n_snp": SNPS, "jn": 10, "block_snps": 10000, "shapes": {}}
for shape, pre in (("1bin", ""), ("8bin+10cov", "c4_"), ("G+GxE", "c5_")):
    try:
        f, w = counters(f"{src}/pmc_{pre}fetch.summary.txt", "FETCH_SIZE"), counters(f"{src}/pmc_{pre}write.summary.txt", "WRITE_SIZE")
    except FileNotFoundError:
        continue
    traffic["shapes"][shape] = {"kernels": {k: {"launches": f[k][0], "fetch_size_mean": f[k][1], "write_size_mean": w.get(k, (0, 0.0))[1],
                                                "bytes_per_algorithmic_byte": round((2 * f[k][1] * 1024 * f[k][0] + w.get(k, (0, 0.0))[1] * 1024 * w.get(k, (0, 0))[0]) / alg, 4)}
                                            for k in f}}
json.dump(traffic, open(f"{dst}/{rnd}_traffic.json", "w"), indent=2)

# 3. BASELINE configurations at HEAD
cfg = {"commit": commit, "command": "tools/run_configs.sh (bench.py --no-cpu-baseline --no-alt --no-cross-check, one GPU)"}
for tag, f in (("C2", "c2"), ("C3", "c3"), ("C4", "c4"), ("C5", "c5"), ("C3_f16x2", "c3_f16x2"), ("C3_f16", "c3_f16")):
    d = json.load(open(f"{src}/{f}.json"))
    r = d["roofline"]
    cfg[tag] = {"workload": d["config"]["workload"], "precision": d["config"]["precision"], "M_snps_per_s": round(d["value"] / 1e6, 2),
                "ms_per_pass": round(d["ms_per_step"], 2), "pass_frac": r["pass_frac"], "gemm1_ms": r["gemm1_ms_per_step"],
                "gemm2_ms": r["gemm2_ms_per_step"], "mid_ms": r["mid_ms_per_step"], "kernels": r["kernels"],
                "covariates_as": d["config"].get("covariates_as"), "end_to_end": d["config"]["end_to_end"]}
json.dump(cfg, open(f"{dst}/{rnd}_configs.json", "w"), indent=1)

# 4. the cross-check sweep
shutil.copy(f"{src}/sweep_summary.txt", f"{dst}/{rnd}_verify_sweep.txt")
print("copied; traffic:", json.dumps({s: {k: v["bytes_per_algorithmic_byte"] for k, v in t["kernels"].items()} for s, t in traffic["shapes"].items()}))
