"""Host logic of sampler='auto' (engine.resolve_sampler / auto_head_k): which kernel family a colony's next construction runs.
No GPU: the rule reads the heuristic's rows only."""
import warnings

import torch

from deepaco_amd import engine


def _ksparse(n, k, seed=0):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(n, 2, generator=g)
    d = (c[:, None] - c).norm(dim=-1)
    d[torch.arange(n), torch.arange(n)] = 1e9
    _, idx = torch.topk(d, k=k, dim=1, largest=False)
    return d, torch.full_like(d, 1e-10).scatter_(1, idx, torch.rand(n, k, generator=g) + 0.05)


def test_head_size_from_the_heuristics_rows():
    d, h40 = _ksparse(300, 40)
    assert engine.auto_head_k(h40) == 62                      # the network's k-sparse output: 40 live entries per row -> the largest
    #                                                           head that leaves a slot free (a few ants then keep the rows in LDS)
    _, h100 = _ksparse(300, 100)
    assert engine.auto_head_k(h100) == 127
    assert engine.auto_head_k(1 / d) is None                  # plain 1/d: a fifth of a row's mass is in the tail
    _, h200 = _ksparse(300, 200)
    assert engine.auto_head_k(h200) is None
    _, h50 = _ksparse(500, 50)                                # TSP-500, k = 50: 13 lanes of 24 bytes x 500 rows fit a CU's LDS -> 51
    assert engine.auto_head_k(h50) == 51
    heavy = h40.clone()
    heavy[:, :] = torch.where(heavy > 1e-9, heavy, torch.full_like(heavy, 2e-4))   # 1 % of a row's mass outside its 40 live entries:
    assert engine.auto_head_k(heavy) == 63                    # not "practically all" in 62 -> the 0.98 rule's 63
    assert engine.auto_head_k(_ksparse(100, 10)[1]) is None   # sizes the head kernels do not cover
    flat = h40.clone()
    flat[:10] = 1e-10                                         # a few flat rows (all live entries below the floor) do not veto
    assert engine.auto_head_k(flat) == 62
    flat[:40] = 1e-10                                         # more than one row in twenty does
    assert engine.auto_head_k(flat) is None


def test_resolution_rules():
    d, h = _ksparse(300, 40)
    cache = {}
    assert engine.resolve_sampler("scan", 300, 30, h, cache) == ("scan", 30)          # explicit choices pass through
    assert engine.resolve_sampler("race", 300, None, h, cache) == ("race", None)
    assert engine.resolve_sampler("auto", 300, 30, 1 / d, cache) == ("scan_sparse", 30)      # after sparsify(k): its k
    assert engine.resolve_sampler("auto", 300, None, h, cache) == ("scan_sparse", 62)
    assert cache["auto_head"][0] is h                                                  # (tested once per heuristic object)
    assert engine.resolve_sampler("auto", 300, None, 1 / d, cache) == ("scan", None)
    assert engine.resolve_sampler("auto", 100, 10, h[:100, :100], {}) == ("scan", None)
    assert engine.resolve_sampler("auto", 2000, 100, h, {}) == ("scan", None)
    engine._warned_sparse_range = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert engine.resolve_sampler("scan_sparse", 100, 10, h, {}) == ("scan", None)
        assert engine.resolve_sampler("scan_sparse", 100, 10, h, {}) == ("scan", None)
    assert len([x for x in w if "scan_sparse" in str(x.message)]) == 1                # said once
    assert engine.resolve_sampler("scan_sparse", 300, 20, h, {}) == ("scan_sparse", 20)


def test_the_sorted_top_values_are_handed_to_the_head_table_once():
    """resolve_sampler leaves the rows' 127 largest values (which its concentration test sorted) in the colony's cache;
    the head table takes them instead of a torch.topk of its own, and they are dropped after that."""
    d, h = _ksparse(300, 40)
    k, top = engine.auto_head_k(h, want_top=True)
    assert k == 62 and top.shape == (300, 127)
    assert torch.equal(top, torch.topk(h, 127, dim=-1).values) and bool((top[:, 1:] <= top[:, :-1]).all())
    assert engine.auto_head_k(1 / d, want_top=True) == (None, None)
    assert engine.auto_head_k(h[:100, :100], want_top=True) == (None, None)
    cache = {}
    assert engine.resolve_sampler("auto", 300, None, h, cache) == ("scan_sparse", 62)
    assert engine.take_auto_top(cache, 1 / d) is None          # another heuristic object: nothing to take (and nothing kept)
    assert "auto_top" not in cache
    cache = {}
    engine.resolve_sampler("auto", 300, None, h, cache)
    got = engine.take_auto_top(cache, h)
    assert torch.equal(got, top)
    assert engine.take_auto_top(cache, h) is None and engine.take_auto_top(None, h) is None
    assert engine.resolve_sampler("auto", 300, None, h, cache) == ("scan_sparse", 62)      # (cached verdict: no second test)
    assert "auto_top" not in cache
