"""Drop-in counterpart of the reference's cvrp/ directory (aco.py)."""
