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


def test_gap_statistics_and_the_equal_or_better_rule():
    """equal_or_better only when the interval's upper end is at or below +0.25 %; wins are counted."""
    import random
    rnd = random.Random(3)
    cpu = [[20.0 + rnd.gauss(0, 0.2) for _ in range(64)] for _ in range(3)]
    same = [[c + rnd.gauss(0, 0.05) for c in cpu[0]] for _ in range(3)]
    g = bench.gap_statistics(same, cpu)
    assert g["equal_or_better"] is True and abs(g["gap"]) < 0.002 and g["ci95"][0] < g["mean_paired_relative_difference"] < g["ci95"][1]
    assert 0 <= g["gpu_better_or_equal_on"] <= 64
    worse = [[c * 1.005 for c in run] for run in same]                      # +0.5 %: not equal, however tight the interval
    assert bench.gap_statistics(worse, cpu)["equal_or_better"] is False
    wide = bench.gap_statistics([[c + rnd.gauss(0, 1.0) for c in cpu[0]][:4]], [cpu[0][:4]])   # four noisy instances: no verdict of equality
    assert wide["equal_or_better"] is False
    assert bench.t975(63) == bench.T975[60] and bench.t975(15) == 2.131 and 1.96 < bench.t975(500) < 1.98


def test_headline_record_carries_the_three_gaps():
    full = _full()
    row = {"gap": 0.001, "ci95": [-0.001, 0.003], "equal_or_better": False, "gpu_better_or_equal_on": 14}
    full["best_cost_gap"] = dict(row, instances=32, iterations=20, gpu_seeds=3, sampler="scan_sparse", gpu_mean_best=17.0, cpu_mean_best=17.0,
                                 samplers={"scan_sparse": row, "scan": row, "race": row}, note="x" * 500)
    rec = json.loads(bench.headline_record(bench._finite(full)))
    assert set(rec["best_cost_gap"]["samplers"]) == {"scan_sparse", "scan", "race"}
    assert rec["best_cost_gap"]["samplers"]["race"] == [0.001, -0.001, 0.003, False, 14]
    assert bench.gap_samplers("scan_sparse") == ["scan_sparse", "scan", "race"] and bench.gap_samplers("scan") == ["scan", "race"]
