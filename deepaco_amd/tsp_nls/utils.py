"""tsp_nls/utils.py surface: same graph construction as tsp/, with the optional start-node feature."""
import os
import sys

try:
    from deepaco_amd.tsp.utils import gen_distance_matrix, gen_pyg_data, load_val_dataset, load_test_dataset  # noqa: F401
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd.tsp.utils import gen_distance_matrix, gen_pyg_data, load_val_dataset, load_test_dataset  # noqa: F401
