"""Drop-in counterpart of the reference's smtwtp/ directory (aco.py)."""
