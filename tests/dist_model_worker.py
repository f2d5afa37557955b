"""Worker of test_distributed_gpu.py::test_two_ranks_without_a_seed_agree: launched by torch.distributed.run, every rank builds
the SAME model class from the same files with seed=None and no annotation file (so the seed, the generated annotation and the
probe vectors all have to be agreed between the ranks) and writes what it drew and estimated."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    work, prefix, pheno = sys.argv[1:4]
    os.chdir(work)                                      # generated_annot lands here
    dist.init_process_group(os.environ.get("PYRHE_DIST_BACKEND", "gloo"))
    from pyrhe_amd import models
    from pyrhe_amd.logger import Logger
    m = models.StreamingRHE(geno_file=prefix, annot_file=None, num_bin=3, pheno_file=pheno, num_jack=6, num_random_vec=10,
                            seed=None, device="cuda", cuda_num=0, log=Logger(suppress=True, debug_mode=False))
    res = m(trait=0)
    out = {"rank": dist.get_rank(), "seed": m.seed, "zsum": float(np.abs(m.all_zb).sum()), "len_bin": [int(v) for v in m.len_bin],
           "sigma": [float(v) for v in res["sigma_ests_total"]], "se": [float(v) for v in res["sig_errs"]]}
    with open(os.path.join(work, f"rank{dist.get_rank()}.json"), "w") as f:
        json.dump(out, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
