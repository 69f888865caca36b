"""`python utils.py` of a problem directory writes its evaluation sets (deepaco_amd/datasets.py; the reference's utils.py
script tails).  CPU: the files exist where the loaders look for them, hold the record layout the loaders unpack, and are
reproducible; the op / bpp sets equal what the reference's scripts write (checked against the reference in the build
container: they are pure torch.rand / gen_instance streams under the stated seeds)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_bpp_and_op_sets_round_trip_through_the_loaders(tmp_path, monkeypatch):
    from deepaco_amd.datasets import write_datasets
    from deepaco_amd.bpp import utils as bpp_utils
    from deepaco_amd.op import utils as op_utils
    root = tmp_path / "data"
    root.mkdir()
    (tmp_path / "run").mkdir()
    monkeypatch.chdir(tmp_path / "run")                      # the reference's scripts run from the problem directory
    w = write_datasets("bpp", bpp_utils, root="../data")
    assert [os.path.basename(p) for p in w] == ["testDataset-120.pt"]
    d = torch.load(w[0])
    assert d.shape == (100, 121)                           # n items + the dummy node (bpp/utils.py:9-13)
    torch.manual_seed(123456)
    assert torch.equal(d[0], bpp_utils.gen_instance(120, "cpu"))            # first record = first draw under the seed
    w2 = write_datasets("op", op_utils, root="../data", sizes=(100,))
    assert sorted(os.path.basename(p) for p in w2) == ["testDataset-100.pt", "valDataset-100.pt"]
    val = torch.load(os.path.join("..", "data", "op", "valDataset-100.pt"))
    torch.manual_seed(12345)
    assert val.shape == (30, 100, 2) and torch.equal(val, torch.rand(size=(30, 100, 2)))
    again = write_datasets("bpp", bpp_utils, root="../data")
    assert torch.equal(torch.load(again[0]), d)


def test_mkp_records_hold_prize_and_weights():
    from deepaco_amd.datasets import _SPECS
    from deepaco_amd.mkp import utils as mkp_utils
    torch.manual_seed(12345)
    rec = _SPECS["mkp"][0][5](mkp_utils, 50)
    assert rec.shape == (50, 6) and bool((rec[:, 0] >= 0).all()) and bool((rec[:, 0] < 1).all())
