#!/usr/bin/env python3
"""Which kernels wait for their global loads one at a time?  Compiles the library's sources to gfx950 assembly (hipcc -S, no GPU
needed) and prints, per kernel, the number of vector-memory loads and of full waits (`s_waitcnt vmcnt(0)`): a kernel whose waits are
about as many as its loads issues a load, waits for it, uses it -- a runtime-trip-count loop of `x = g[i]; lds[j] = x;`, or a load
inside an `if (e < E)` block, compiles to exactly that, and every iteration is then a whole memory round trip.  Round 6's last
session found gnn_t_node_lin_bwd (34 / 34), gnn_t_bwd_stats, the pheromone update's chunk loader and the LDS-heads table copy this
way; the cure is the same everywhere: clamped indices, all loads into registers first, the bounds test on the
values.  usage: tools/scan_serialized_loads.py [file.hip ...]   (default: every .hip under deepaco_amd/csrc)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "deepaco_amd", "csrc", "*.hip")))
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-x", "hip", "-S", "--cuda-device-only"]
for src in srcs:
    with tempfile.NamedTemporaryFile(suffix=".s") as out:
        r = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-o", out.name, src], capture_output=True, text=True)
        if r.returncode:
            print(f"{src}: hipcc failed\n{r.stderr[-400:]}")
            continue
        kern, stats = None, {}
        for line in open(out.name):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                kern = m.group(1)
                stats[kern] = [0, 0]
            elif kern:
                if "s_endpgm" in line:
                    kern = None
                elif re.search(r"\s(global|buffer|scratch)_load", line):
                    stats[kern][0] += 1
                elif "s_waitcnt vmcnt(0)" in line:
                    stats[kern][1] += 1
    for k, (loads, waits) in stats.items():
        if loads >= 4 and waits * 2 >= loads:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:110]
            print(f"{os.path.basename(src):28s} loads {loads:4d}  full waits {waits:4d}  {name}")
