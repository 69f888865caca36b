#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/exp_nls_relabel.py 2>&1 | grep labelling
