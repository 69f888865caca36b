"""Drop-in counterpart of the reference's bpp/ directory (aco.py)."""
