"""`python utils.py` in a problem directory writes that problem's evaluation sets under ../data/<problem>/, as the
reference's utils.py files do when run as scripts (cvrp/utils.py:55-67, cvrp_nls/utils.py:113-124, op/utils.py:60-70,
pctsp/utils.py:75-85, sop/utils.py:85-98, smtwtp/utils.py:52-65, bpp/utils.py:45-56, mkp/utils.py:60-82; tsp/ and
tsp_nls/ ship their sets and generate nothing).  One table instead of eight script tails: per problem the seed policy,
the sizes, how one instance becomes one record, and the container format -- so that the files hold what the reference's
load_*_dataset functions (and the drop-in's) expect.  Instances come from the problem's own utils module (the drop-in's
generators consume torch's RNG in the reference's order: tests/test_utils_siblings.py, g9 fixtures)."""
import os
import pickle

import torch


def _stack(records):
    return torch.stack(records)


# problem -> list of (file pattern, seed, reseed per size?, sizes, instances per size, record(utils, n), container)
_SPECS = {
    "cvrp": [("testDataset-{n}.pt", 123456, False, (20, 100, 500), 100,
              lambda u, n: (lambda dem, dist: torch.cat((dem.unsqueeze(0), dist), dim=0))(*u.gen_instance(n, "cpu")), _stack)],
    "cvrp_nls": [("testDataset-{n}.pt", 123456, True, (100, 500, 1000, 2000), 100,
                  lambda u, n: (lambda dem, dist, pos: torch.vstack([dem, pos.T, dist]))(*u.gen_instance(n, "cpu", True)), _stack)],
    "op": [("valDataset-{n}.pt", 12345, False, (100, 200, 300), 1, lambda u, n: torch.rand(size=(30, n, 2)), lambda r: r[0]),
           ("testDataset-{n}.pt", 123456, False, (100, 200, 300), 1, lambda u, n: torch.rand(size=(100, n, 2)), lambda r: r[0])],
    "pctsp": [("testDataset-{n}.pt", 123456, False, (20, 100, 500), 100,
               lambda u, n: (lambda d, pr, pe: torch.cat([d, pr.unsqueeze(0), pe.unsqueeze(0)], dim=0))(*u.gen_inst(n, "cpu")), _stack)],
    "sop": [("test{n}.pkl", 123456, False, (20, 50, 100), 100, lambda u, n: list(u.training_instance_gen(n, "cpu")), list)],
    "smtwtp": [("test{n}.pkl", 123456, False, (50, 100, 500), 100, lambda u, n: list(u.instance_gen(n, "cpu")), list)],
    "bpp": [("testDataset-{n}.pt", 123456, False, (120,), 100, lambda u, n: u.gen_instance(n, "cpu"), _stack)],
    "mkp": [("valDataset-{n}.pt", 12345, False, (50,), 100,
             lambda u, n: (lambda prize, w: torch.cat((prize.unsqueeze(1), w), dim=1))(*u.gen_instance(n, 5, "cpu")), _stack),
            ("testDataset-{n}.pt", 123456, False, (50,), 100,
             lambda u, n: (lambda prize, w: torch.cat((prize.unsqueeze(1), w), dim=1))(*u.gen_instance(n, 5, "cpu")), _stack)],
}


def write_datasets(problem, utils_module, root="../data", sizes=None):
    """Generate and save every evaluation set of `problem` under <root>/<problem>/ (created if missing; <root> must exist,
    as in the reference).  `sizes` restricts the problem sizes (tests).  Returns the paths written."""
    out_dir = os.path.join(root, problem)
    os.makedirs(out_dir, exist_ok=True)
    written = []
    for pattern, seed, per_size, all_sizes, count, record, container in _SPECS[problem]:
        if not per_size:
            torch.manual_seed(seed)
        for n in all_sizes:
            if per_size:
                torch.manual_seed(seed)
            records = [record(utils_module, n) for _ in range(count)]       # (always generated: later sizes see the same RNG state)
            if sizes is not None and n not in sizes:
                continue
            data = container(records)
            path = os.path.join(out_dir, pattern.format(n=n))
            if path.endswith(".pkl"):
                with open(path, "wb") as f:
                    pickle.dump(data, f)
            else:
                torch.save(data, path)
            written.append(path)
    return written
