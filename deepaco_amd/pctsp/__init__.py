"""Drop-in counterpart of the reference's pctsp/ directory (aco.py)."""
