"""ACO with the class surface of the reference's mkp/aco.py (`from aco import ACO`), on MI355X.
The implementation lives in deepaco_amd/siblings.py (class MKP)."""
import os
import sys

try:
    from deepaco_amd.siblings import MKP as ACO  # noqa: F401
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd.siblings import MKP as ACO  # noqa: F401
