#!/bin/bash
# Round 6, GPU session AI: what an N-GPU bench run executes, on one GPU (force-dist), and the torchrun launch line of the driver at N = 1.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06ai
mkdir -p $OUT
cd $R
timeout 900 python tools/check_force_dist.py 20 > $OUT/force_dist_check.log 2>&1; echo "force-dist rc=$?"; tail -1 $OUT/force_dist_check.log | cut -c1-600
cp gpurun_out/force_dist_check.json $OUT/ 2>/dev/null
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras > $OUT/torchrun_n1.log 2> $OUT/torchrun_n1.err; echo "torchrun rc=$?"; tail -1 $OUT/torchrun_n1.log | cut -c1-400
