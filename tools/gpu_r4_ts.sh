#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/two_stream_experiment.py 300 > gpurun_out/two_stream.jsonl 2>gpurun_out/two_stream.err
