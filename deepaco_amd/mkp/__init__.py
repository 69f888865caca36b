"""Drop-in counterpart of the reference's mkp/ directory (aco.py)."""
