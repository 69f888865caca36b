"""Drop-in counterpart of the reference's tsp_nls/ directory (aco.py, two_opt.py)."""
