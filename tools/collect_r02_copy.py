#!/usr/bin/env python3
"""gpurun_out/ (scratch) -> profiles/r02_* (tracked): the round-2 summaries the judge reads.  Run from the repo root after
tools/collect_r02.sh and the r2 sessions; argument = the commit the GPU runs were made at."""
import csv, glob, json, os, re, shutil, sys

commit = sys.argv[1] if len(sys.argv) > 1 else "HEAD"
src, r2, dst = "gpurun_out/r02_final", "gpurun_out/r2", "profiles"

# 1. the default bench line (plain run) and the same command under rocprofv3 --kernel-trace --stats
shutil.copy(f"{src}/bench_plain.json", f"{dst}/r02_bench.json")
shutil.copy(f"{src}/bench_trace.json", f"{dst}/r02_bench_traced.json")
stats = glob.glob(f"{src}/trace/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(stats, f"{dst}/r02_kernel_stats.csv")

# 2. PMC summaries (one section per separate --pmc pass) and the traffic ratios bench.py quotes
with open(f"{dst}/r02_pmc_summary.txt", "w") as out:
    out.write(f"# rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --n-snp 100000 --jn 10 --steps 1 --warmup 1 ... at {commit}\n"
              "# one separate run per section (tools/collect_r02.sh); mean per dispatch (tools/pmc_summary.py)\n")
    for sec in ("fetch", "write", "sq1", "sq2"):
        out.write(f"== pmc_{sec}\n" + open(f"{src}/pmc_{sec}.summary.txt").read())
txt = open(f"{dst}/r02_pmc_summary.txt").read()
N, SNPS, PASSES = 500_000, 100_000, 2
alg = SNPS * ((N + 3) // 4) * PASSES


def grab(section, kern, counter):
    sec = txt.split("== " + section)[1].split("== ")[0]
    blk = sec.split(kern)[1]
    m = re.search(counter + r"\s+n=\s*(\d+)\s+mean=([0-9.e+]+)", blk)
    return int(m.group(1)), float(m.group(2))


traffic = {"_comment": "HBM bytes per algorithmic byte of the two headline kernels: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 B summed over the "
                       "launches of the PMC run (MI355X_MICROARCH.md: FETCH_SIZE counts half of a wide streaming read on gfx950), "
                       "divided by the packed genotype bytes those launches covered; separate --pmc runs, profiles/r02_pmc_summary.txt",
           "commit": commit, "n_indv": N, "n_snp": SNPS, "jn": 10, "block_snps": 10000, "kernels": {}}
for key, names in {"gemm1_i8_kernel": ["void gemm1_i8w_kernel<3, false, true", "void gemm1_i8_kernel<3, false, 0, true"],
                   "gemm2_i8p_kernel": ["void gemm2_i8p_kernel<1, false, true"]}.items():
    name = next(n for n in names if n in txt)          # the headline's GEMM 1 is gemm1_i8w_kernel since round 2
    nf, f = grab("pmc_fetch", name, "FETCH_SIZE")
    nw, w = grab("pmc_write", name, "WRITE_SIZE")
    traffic["kernels"][key] = {"kernel": name.replace("void ", "") + ">", "launches": nf, "fetch_size_mean": f, "write_size_mean": w,
                               "bytes_per_algorithmic_byte": round((2 * f * 1024 * nf + w * 1024 * nw) / alg, 4)}
json.dump(traffic, open(f"{dst}/r02_traffic.json", "w"), indent=2)

# 3. BASELINE configurations at HEAD
cfg = {"commit": commit, "command": "tools/run_configs.sh (bench.py --no-cpu-baseline --no-alt --no-cross-check, one GPU)"}
for tag, f in (("C2", "c2"), ("C3", "c3"), ("C4", "c4"), ("C5", "c5"), ("C3_f16x2", "c3_f16x2"), ("C3_f16", "c3_f16")):
    d = json.load(open(f"{src}/{f}.json"))
    r = d["roofline"]
    cfg[tag] = {"workload": d["config"]["workload"], "precision": d["config"]["precision"], "M_snps_per_s": round(d["value"] / 1e6, 2),
                "ms_per_pass": round(d["ms_per_step"], 2), "pass_frac": r["pass_frac"], "gemm1_ms": r["gemm1_ms_per_step"],
                "gemm2_ms": r["gemm2_ms_per_step"], "mid_ms": r["mid_ms_per_step"], "kernels": r["kernels"],
                "end_to_end": d["config"]["end_to_end"]}
json.dump(cfg, open(f"{dst}/r02_configs.json", "w"), indent=1)

# 4. sweep, skeleton, layout patterns, GEMM 1 diagnostics, the 20-column kernel's counters
shutil.copy(f"{src}/sweep_summary.txt", f"{dst}/r02_verify_sweep.txt")
shutil.copy(f"{r2}/fused_skel2.txt", f"{dst}/r02_fused_skeleton.txt")
shutil.copy(f"{r2}/pattern2.txt", f"{dst}/r02_layout_patterns.txt")
shutil.copy(f"{r2}/pmc_c5.txt", f"{dst}/r02_pmc_c5_gxe.txt")
with open(f"{dst}/r02_gemm1_diagnostics.txt", "w") as out:
    out.write("# bench.py --steps 4 --warmup 1 --no-alt --no-cpu-baseline [--opt ...] at the headline shape; ms per pass: whole pass, GEMM 1, GEMM 2\n"
              "# diag_* rows: diagnostic library (PYRHE_LIB=libpyrhe_sketch_diag.so), WRONG results by construction, timing only\n")
    for f in sorted(glob.glob(f"{r2}/knob_*.json") + glob.glob(f"{r2}/diag_*.json") + glob.glob(f"{r2}/deep_*.json")):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        r = d["roofline"]
        out.write(f"{os.path.basename(f)[:-5]:40s} {d['config']['workload'][:60]:60s} {d['ms_per_step']:7.2f} {r['gemm1_ms_per_step']:7.2f} {r['gemm2_ms_per_step']:7.2f}  {d['config'].get('options')}\n")
print("copied; traffic:", json.dumps(traffic["kernels"]))
