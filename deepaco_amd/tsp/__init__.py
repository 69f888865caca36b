"""Drop-in counterpart of the reference's tsp/ directory (aco.py)."""
