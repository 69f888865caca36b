#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_10_layout_knob.py -x -q -m gpu 2>&1 | tail -5
