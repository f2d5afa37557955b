#!/bin/bash
# two ranks on one GPU (gloo), every leg of bench.py enabled (plink-coded pass, cross-check, f16 pass): no leg may leave the ranks out of step
mkdir -p gpurun_out/r3
export PYRHE_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --n-indv 40000 --n-snp 40000 --jn 10 --steps 2 --warmup 1 --n-bin 4 --n-cov 5 --no-cpu-baseline > gpurun_out/r3/two_rank_all_legs.json 2> gpurun_out/r3/two_rank_all_legs.err; echo "rc=$?"
tail -c 1800 gpurun_out/r3/two_rank_all_legs.json; tail -5 gpurun_out/r3/two_rank_all_legs.err
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29656 bench.py --gpus 4 --n-indv 40000 --n-snp 40000 --jn 10 --steps 2 --warmup 1 --genie G+GxE --no-cpu-baseline > gpurun_out/r3/four_rank_all_legs.json 2> gpurun_out/r3/four_rank_all_legs.err; echo "rc=$?"
tail -c 900 gpurun_out/r3/four_rank_all_legs.json; tail -5 gpurun_out/r3/four_rank_all_legs.err
