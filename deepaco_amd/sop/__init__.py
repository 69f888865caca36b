"""Drop-in counterpart of the reference's sop/ directory (aco.py)."""
