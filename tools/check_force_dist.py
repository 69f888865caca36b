#!/usr/bin/env python3
"""What an N-GPU bench run executes, on the one GPU a builder has: the default bench.py line against
`bench.py --gpus 1 --force-dist` (RCCL process group of one rank: the barrier / max-over-ranks path, the same colony and sampler)
and against `--shard ants --force-dist` (the ant-sharded colony: same sampler as N = 1 since round 6, one all-gather per iteration
that degenerates at world size 1).  Writes gpurun_out/force_dist_check.json and fails if the instance-sharded value differs from the
plain run by more than 2 %.   usage: tools/check_force_dist.py [steps=20]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "20"


def run(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "3", "--no-extras", "--no-cpu", "--min-seconds", "0"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not lines:
        raise SystemExit(f"no record from {' '.join(cmd)}:\n{out.stderr[-2000:]}")
    return json.loads(lines[-1])


plain = [run([]) for _ in range(2)]
forced = [run(["--gpus", "1", "--force-dist"]) for _ in range(2)]
ants = run(["--gpus", "1", "--force-dist", "--shard", "ants"])
best = lambda rs: max(r["value"] for r in rs)
ratio = best(forced) / best(plain)
rec = {"plain_values": [r["value"] for r in plain], "force_dist_values": [r["value"] for r in forced], "ratio": ratio,
       "within_2_percent": abs(ratio - 1) <= 0.02,
       "samplers": {"plain": plain[0]["config"]["sampler"], "force_dist": forced[0]["config"]["sampler"], "ant_sharded": ants["config"]["sampler"]},
       "ant_sharded_value": ants["value"], "ant_sharded_over_plain": ants["value"] / best(plain),
       "rccl": forced[0].get("rccl"), "ant_sharded_rccl": ants.get("rccl")}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "force_dist_check.json"), "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps(rec))
assert rec["samplers"]["plain"] == rec["samplers"]["force_dist"] == rec["samplers"]["ant_sharded"], rec["samplers"]
assert rec["within_2_percent"], ratio
