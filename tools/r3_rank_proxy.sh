#!/bin/bash
# What ONE rank of an 8-GPU job does per pass, measured on one GPU: the shard of rank 0 at G = 8 (13 of the 100 blocks of the
# headline cohort = 130 000 SNPs) under the driver's launch line with RCCL (one rank, every collective executed).  The pass time
# against 13 % of the single-GPU pass shows the fixed per-pass cost that limits strong scaling.
mkdir -p gpurun_out/r3
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["ms_per_step"],2), "ms per pass | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], "| e2e", d["config"]["end_to_end"], d["config"].get("dist"))'
for cfg in "" "--n-bin 8 --n-cov 10" "--genie G+GxE"; do
  python bench.py --n-snp 130000 --jn 13 --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-cross-check $cfg 2>/dev/null | tail -1 | python -c "$fmt" "13 blocks, no process group [$cfg]"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --force-collectives --n-snp 130000 --jn 13 --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-cross-check $cfg 2>/dev/null | tail -1 | python -c "$fmt" "13 blocks, RCCL one rank      [$cfg]"
done
