"""Drop-in counterpart of the reference's op/ directory (aco.py)."""
