for pad in 0 4096 8192 13000 22000 35000 60000; do
  DACO_SCAN32_LDS_PAD=$pad python bench.py --no-cpu --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pad', $pad, 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'ms_step', round(d['ms_per_step'],4))"
done
