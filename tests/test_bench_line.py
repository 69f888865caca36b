"""bench.py's last stdout line is a compact record (a 28 KB line was not parsed by the driver).

The full record of a real run (profiles/r04_bench_default.json, 28 KB) goes through bench.headline_record: the result must
stay below 4 KB, be strict JSON (no NaN / Infinity) and carry the contract's keys with roofline and cpu_baseline."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _full():
    return json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))


def test_headline_record_is_small_and_complete():
    txt = bench.headline_record(bench._finite(_full()))
    assert len(txt) <= bench.HEADLINE_MAX_BYTES == 4096
    assert "\n" not in txt
    rec = json.loads(txt, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rec["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(rec["cpu_baseline"])
    assert "workload" in rec["config"] and "model" not in rec["config"]
    assert abs(rec["roofline"]["frac"] - rec["roofline"]["achieved"] / rec["roofline"]["peak"]) < 1e-4


def test_headline_record_survives_a_bloated_input():
    full = _full()
    full["extras"] = {f"entry_{i}": {"value": float(i), "note": "x" * 5000} for i in range(300)}
    full["cpu_baseline"]["sample"] = "y" * 10000
    txt = bench.headline_record(bench._finite(full))
    assert len(txt) <= 4096
    rec = json.loads(txt)
    assert rec["roofline"] and rec["cpu_baseline"]


def test_non_finite_numbers_become_null():
    full = _full()
    full["value"] = float("nan")
    full["best_cost_gap"]["gap"] = float("inf")
    rec = json.loads(bench.headline_record(bench._finite(full)))
    assert rec["value"] is None


def test_roofline_counts_true_row_bytes():
    """frac is computed from the 4 n bytes a step needs, not from the padded row the kernel streams."""
    rf = bench.roofline_rows(500, 512, 64, "scan", 1.2)
    assert rf["row_bytes_per_launch"] == 64 * 512 * 499 * 4.0 * 500
    assert rf["padded_row_floats"] == 512
