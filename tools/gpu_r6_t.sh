#!/bin/bash
# Round 6, GPU session T: the colony iteration without the int64 paths tensor: parity, A/B at the headline shape and at one instance.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06t
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_11_scan_sparse.py tests/test_gpu_00_tsp.py tests/test_gpu_12_streams.py -m gpu -q --timeout 300 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log | cut -c1-400
timeout 300 python tools/ab_compact_tours.py 64 2>/dev/null | tee $OUT/ab_compact_b64.json
timeout 300 python tools/ab_compact_tours.py 1 2>/dev/null | tee $OUT/ab_compact_b1.json
timeout 300 python tools/ab_compact_tours.py 8 2>/dev/null | tee $OUT/ab_compact_b8.json
