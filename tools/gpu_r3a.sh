#!/bin/bash
cd $GRAFT_REPO_ROOT
for T in 512 1024; do DACO_CVRP_LS_THREADS=$T timeout 200 python tools/measure_cvrp_ls.py 2>&1 | tail -1 | cut -c1-200; done
