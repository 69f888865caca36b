#!/bin/bash
mkdir -p gpurun_out
for P in 0 12000 30000; do
  DACO_SPARSE_PAD_LDS=$P timeout 300 python tools/run_headline_kernel.py 5 64 2048 1000 scan_sparse 2>/dev/null | sed "s/^{/{\"pad\": $P, /"
done > gpurun_out/occ.jsonl
for P in 0 7000 15000 28000; do
  DACO_SPARSE_PAD_LDS=$P timeout 300 python tools/run_headline_kernel.py 8 48 512 500 scan_sparse 2>/dev/null | sed "s/^{/{\"pad\": $P, /"
done >> gpurun_out/occ.jsonl
