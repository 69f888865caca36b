"""ACO with the class surface of the reference's sop/aco.py (`from aco import ACO`), on MI355X.
The implementation lives in deepaco_amd/siblings.py (class SOP)."""
import os
import sys

try:
    from deepaco_amd.siblings import SOP as ACO  # noqa: F401
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd.siblings import SOP as ACO  # noqa: F401
