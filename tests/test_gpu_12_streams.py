"""StreamedTSP (engine): the instances of a BatchedTSP run as several colonies on their own HIP streams -- SURVEY.md 8(e)'s
partitioning inside one GPU (independent instances, no exchange).  It must be the SAME computation: pheromone, best costs and best
tours of one BatchedTSP over all instances, bit for bit, for every sampler the colony loop offers."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def instances(B, n, seed):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(B, n, 2, generator=g)
    d = (c[:, :, None] - c[:, None]).norm(dim=-1)
    i = torch.arange(n)
    d[:, i, i] = 1e9
    return d.to(dev())


@pytest.mark.parametrize("sampler,parts,B,n,kw", [("scan", 3, 6, 200, {}), ("scan_sparse", 4, 8, 300, {}), ("race", 2, 5, 150, {}),
                                                 ("scan", 4, 4, 100, dict(elitist=True)), ("scan", 2, 4, 120, dict(min_max=True))])
def test_streamed_colonies_equal_one_batched_colony(sampler, parts, B, n, kw):
    from deepaco_amd import engine
    d = instances(B, n, 3 * n + B)
    one = engine.BatchedTSP(d, n_ants=48, seed=17, sampler=sampler, **kw)
    many = engine.StreamedTSP(d, parts=parts, n_ants=48, seed=17, sampler=sampler, **kw)
    one.sparsify(n // 10)
    many.sparsify(n // 10)
    for _ in range(5):
        one.step()
        many.step()
    torch.cuda.synchronize()
    assert torch.equal(one.pheromone, many.pheromone)
    assert torch.equal(one.lowest_cost, many.lowest_cost)
    assert torch.equal(one.shortest_path, many.shortest_path)
    assert many.iteration == one.iteration == 5


def test_streamed_colony_run_and_uneven_split():
    from deepaco_amd import engine
    d = instances(7, 160, 5)                                   # 7 instances over 3 streams: 2 + 2 + 3
    many = engine.StreamedTSP(d, parts=3, n_ants=32, seed=4)
    many.sparsify(16)
    low = many.run(6)
    one = engine.BatchedTSP(d, n_ants=32, seed=4)
    one.sparsify(16)
    assert torch.equal(low, one.run(6)) and low.shape == (7,)
    assert many.shortest_path.sort(dim=1).values.tolist() == [list(range(160))] * 7


def test_reading_state_between_steps_without_a_device_sync():
    """`low = many.lowest_cost; many.step()` -- the gather is queued on the current stream, the next step rewrites
    the parts' state in place on their own streams.  The parts wait for the current stream at the start of a step, so every
    gathered copy is the state of ITS iteration (compared with one BatchedTSP read at the same points; no synchronize in between)."""
    from deepaco_amd import engine
    d = instances(8, 400, 11)
    one = engine.BatchedTSP(d, n_ants=128, seed=9)
    many = engine.StreamedTSP(d, parts=4, n_ants=128, seed=9)
    one.sparsify(40)
    many.sparsify(40)
    got, want = [], []
    big = torch.rand(4096, 4096, device=dev())
    for _ in range(6):
        one.step()
        want.append((one.lowest_cost.clone(), one.pheromone.clone(), one.shortest_path.clone()))
    for _ in range(6):
        many.step()
        for _ in range(4):
            big = big @ big * 1e-3                                     # keeps the current stream busy: the gathers below queue up behind it
        got.append((many.lowest_cost, many.pheromone, many.shortest_path))
    torch.cuda.synchronize()
    for (gl, gp, gs), (wl, wp, ws) in zip(got, want):
        assert torch.equal(gl, wl) and torch.equal(gp, wp) and torch.equal(gs, ws)
