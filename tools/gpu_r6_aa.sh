#!/bin/bash
# Round 6, GPU session AA: ACO.run on head rows through a one-instance colony: parity with the plain loop, the reference's call patterns timed.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06aa
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_00_tsp.py tests/test_gpu_14_surface.py tests/test_gpu_11_scan_sparse.py -m gpu -q --timeout 300 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_07_net.py -m gpu -q --timeout 300 -x -k "notebook" > $OUT/pytest_notebook.log 2>&1; tail -2 $OUT/pytest_notebook.log
timeout 600 python tools/time_reference_calls.py 100 2>&1 | grep -v amdgpu.ids | tee $OUT/reference_calls.txt
