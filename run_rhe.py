#!/usr/bin/env python3
"""Command line of the MI355X build, flag-for-flag the reference's ``run_rhe.py`` (``run_rhe.py:160-201``):
same options, ``--config`` ini section ``[PyRHE_Config]``, banner, option echo, trait loop and ``Runtime:`` line.

Deliberate, documented deviations:
  * ``--device`` defaults to ``cuda`` (the sketch has no CPU path; ``--device cpu`` is an error);
  * ``--seed`` is parsed as an integer (the reference hands NumPy a string and fails);
  * ``--num_workers`` is accepted and ignored: parallelism is one process per GPU
    (``python -m torch.distributed.run --nproc-per-node N run_rhe.py ...``), blocks sharded across ranks;
  * new: ``--precision {f16x2,f16,f16e}``, ``--compat {reference,fixed}``, ``--recompute_pass2``;
"""
import argparse
import configparser
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def parse_config(config_path, section):
    cfg = configparser.ConfigParser()
    cfg.read(config_path)
    return dict(cfg.items(section))


def coerce(value, default):
    """ini strings -> the type of the flag's default; ``none`` -> None (run_rhe.py:18-26)."""
    if value.lower() == "none":
        return None
    if isinstance(default, bool):
        return value.lower() in ("true", "1", "yes")
    if isinstance(default, int):
        return int(value)
    return value


def build_parser():
    p = argparse.ArgumentParser(description="PyRHE on MI355X")
    p.add_argument("--model", type=str, default="rhe", choices=["rhe", "genie", "rhe_dom"])
    p.add_argument("--genie_model", type=str, default="G+GxE+NxE", choices=["G", "G+GxE", "G+GxE+NxE"])
    p.add_argument("--streaming", action="store_true", help="use streaming version")
    p.add_argument("--trace", "-tr", action="store_true", help="get the trace estimate")
    p.add_argument("--trace_dir", type=str, default="", help="directory to save the trace information")
    p.add_argument("--benchmark_runtime", action="store_true", help="benchmark the runtime")
    p.add_argument("--genotype", "-g", type=str, help="genotype file path")
    p.add_argument("--phenotype", "-p", type=str, default=None, help="phenotype file path")
    p.add_argument("--covariate", "-c", type=str, default=None, help="Covariate file path")
    p.add_argument("--cov_one_hot_conversion", action="store_true")
    p.add_argument("--categorical_threshhold", type=int, default=100)
    p.add_argument("--env", "-e", type=str, default=None, help="Environment file path")
    p.add_argument("--annotation", "-annot", type=str, default=None, help="Annotation file path")
    p.add_argument("--num_vec", "-k", type=int, default=10, help="The number of random vectors (10 is recommended).")
    p.add_argument("--num_bin", "-b", type=int, default=8, help="Number of bins")
    p.add_argument("--num_workers", type=int, default=8, help="accepted for compatibility; ignored")
    p.add_argument("--num_block", "-jn", type=int, default=100, help="The number of jackknife blocks.")
    p.add_argument("--seed", "-s", default=None, help="Random seed")
    p.add_argument("--device", type=str, default="cuda", help="device to use (cuda only)")
    p.add_argument("--cuda_num", type=int, default=None, help="GPU index")
    p.add_argument("--output", "-o", type=str, default="test.out", help="output of the file")
    p.add_argument("--geno_impute_method", type=str, default="binary", choices=["binary", "mean"])
    p.add_argument("--cov_impute_method", type=str, default="ignore", choices=["ignore", "mean"])
    p.add_argument("--samp_prev", default=None)
    p.add_argument("--pop_prev", default=None)
    p.add_argument("--suppress", action="store_true", help="do not print out the outputs to stdout, log file only")
    p.add_argument("--debug", action="store_true", help="debug mode")
    p.add_argument("--debug_output", type=str, default="test")
    p.add_argument("--config", type=str, help="Configuration file path")
    # MI355X build
    p.add_argument("--precision", type=str, default="i8x3", choices=["i8x3", "f16x2", "f16", "f16e", "i8f16"])
    p.add_argument("--compat", type=str, default="reference", choices=["reference", "fixed"])
    p.add_argument("--recompute_pass2", action="store_true")
    return p


def main(args):
    import torch
    import torch.distributed as dist
    from pyrhe_amd.logger import Logger
    from pyrhe_amd.models import GENIE, RHE, RHE_DOM, StreamingGENIE, StreamingRHE, StreamingRHE_DOM

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    log = Logger(output_file=args.output if rank == 0 else None, suppress=args.suppress or rank != 0, debug_mode=args.debug)
    for line in ("##################################", "#                                #",
                 "#          PyRHE (v1.0.0)        #", "#                                #",
                 "##################################"):
        log._log(line)
    log._log("\n")
    options = {"-g (genotype)": args.genotype, "-annot (annotation)": args.annotation, "-p (phenotype)": args.phenotype,
               "-c (covariates)": args.covariate, "-o (output)": args.output, "-k (# random vectors)": args.num_vec,
               "-jn (# jackknife blocks)": args.num_block, "--num_workers": args.num_workers, "--device": args.device,
               "--geno_impute_method": args.geno_impute_method, "--cov_impute_method": args.cov_impute_method}
    log._log("Active essential options:")
    for flag, val in options.items():
        log._log(f"\t{flag} {val}")
    log._log("\n")
    log._debug(args)
    if (args.samp_prev is not None) != (args.pop_prev is not None):
        raise ValueError("Must set both or neither of --samp-prev and --pop-prev.")
    params = dict(model=args.model, geno_file=args.genotype, annot_file=args.annotation, pheno_file=args.phenotype,
                  cov_file=args.covariate, num_jack=args.num_block, num_bin=args.num_bin, num_random_vec=args.num_vec,
                  geno_impute_method=args.geno_impute_method, cov_impute_method=args.cov_impute_method,
                  cov_one_hot_conversion=args.cov_one_hot_conversion, categorical_threshhold=args.categorical_threshhold,
                  device=args.device, cuda_num=args.cuda_num, multiprocessing=args.num_workers > 1,
                  num_workers=args.num_workers, seed=None if args.seed is None else int(args.seed),
                  get_trace=args.trace, trace_dir=args.trace_dir, samp_prev=args.samp_prev, pop_prev=args.pop_prev, log=log,
                  precision=args.precision, compat=args.compat, recompute_pass2=args.recompute_pass2)
    if args.model == "rhe":
        model = (StreamingRHE if args.streaming else RHE)(**params)
    elif args.model == "genie":
        params.update(env_file=args.env, genie_model=args.genie_model)
        model = (StreamingGENIE if args.streaming else GENIE)(**params)
    elif args.model == "rhe_dom":
        model = (StreamingRHE_DOM if args.streaming else RHE_DOM)(**params)
    else:
        raise ValueError("Unsupported Model")
    results, runtime = {}, 0.0
    for trait in range(model.num_traits):
        t0 = time.time()
        res = model(trait=trait)
        runtime = time.time() - t0
        results[f"Trait{trait}"] = {**res, "runtime": runtime}
    log._log("Runtime: ", runtime)
    log._save_log()
    if world > 1:
        dist.destroy_process_group()
    return results


if __name__ == "__main__":
    parser = build_parser()
    args = parser.parse_args()
    if args.config:
        cfg = parse_config(args.config, "PyRHE_Config")
        for key, default in vars(args).items():
            if key in cfg:
                setattr(args, key, coerce(cfg[key], default))
    main(args)
