"""ctypes binding of libdeepaco_hip.so (include/deepaco_hip.h).  Fails loudly if absent."""
import ctypes as C
import os

import torch  # noqa: F401  -- must be loaded first: libdeepaco_hip.so then binds to the HIP runtime
#                              (libamdhip64.so.7) torch already mapped, so streams/pointers are shared

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DACO_LIB_PATH: another build of the same library -- the A/B sessions under tools/ measure two builds in one process tree)
LIB_PATH = os.environ.get("DACO_LIB_PATH") or os.path.join(_HERE, "lib", "libdeepaco_hip.so")

RACE_NOISE, RACE_PHILOX, SCAN, SCAN_WAVE = 0, 1, 2, 3
MAX_NODES = 4096

_lib = None

_vp, _i, _l, _f, _u64, _u32, _sz = (C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint64, C.c_uint32,
                                     C.c_size_t)

# name -> (restype, argtypes); must list every symbol include/deepaco_hip.h declares
SIGNATURES = {
    "daco_version": (_i, []),
    "daco_last_error": (C.c_char_p, []),
    "daco_vec_for_n": (_i, [_i]),
    "daco_ld_for_n": (_i, [_i]),
    "daco_tsp_sample_workspace_bytes": (_sz, [_i, _i, _i]),
    "daco_tsp_sample": (_i, [_vp, _i, _i, _i, _vp, _l, _vp, _l, _f, _f, _i, _i, _vp, _i, _vp, _u64, _u64, _vp,
                             _u32, _i, _vp, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _sz, _vp, _vp]),
    "daco_tsp_sparse_workspace_bytes": (_sz, [_i, _i, _i]),
    "daco_tsp_sparse_workspace_bytes_general": (_sz, [_i, _i, _i]),
    "daco_tsp_sparse_tours_offset": (_sz, [_i, _i, _i]),
    "daco_tsp_sample_sparse": (_i, [_vp, _i, _i, _i, _vp, _l, _vp, _l, _f, _f, _vp, _i, _vp, _i, _u64, _u64, _vp, _u32, _i, _vp, _vp,
                                    _vp, _l, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "daco_tsp_sample_race_head": (_i, [_vp, _i, _i, _i, _vp, _l, _vp, _l, _f, _f, _vp, _i, _vp, _i, _u64, _u64, _vp, _u32, _i, _vp, _vp,
                                       _vp, _l, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "daco_tsp_sample_heads": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _l, _vp, _l, _f, _f, _vp, _i, _vp, _i, _u64, _u64, _vp, _u32, _i, _vp, _vp,
                                   _vp, _l, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "daco_pheromone_update_heads": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _f, _i, _vp, _vp, _f, _vp, _vp, _vp, _sz,
                                         _vp, _l, _f, _f, _vp, _i, _i, _i, _vp, _sz]),
    "daco_allreduce_delta_tau": (_i, [_vp, _vp, _vp, _sz]),
    "daco_tour_costs": (_i, [_vp, _i, _i, _i, _i, _vp, _l, _vp, _i, _vp]),
    "daco_pheromone_update_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "daco_pheromone_update": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _f, _i, _i, _vp, _vp, _f, _vp, _vp, _i, _vp,
                                   _sz]),
    "daco_prob_matrix": (_i, [_vp, _i, _i, _vp, _l, _vp, _l, _f, _f, _i, _vp, _sz]),
    "daco_pick_move": (_i, [_vp, _i, _i, _i, _vp, _sz, _i, _vp, _vp, _vp, _u64, _u64, _u32, _i, _vp, _vp, _vp, _vp]),
    "daco_directed_table_bytes": (_sz, [_i, _i, _i]),
    "daco_track_best": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f]),
    "daco_track_best_tours16": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f]),
    "daco_cvrp_sample": (_i, [_vp, _i, _i, _i, _vp, _l, _vp, _l, _f, _f, _vp, _f, _i, _vp, _i, _u64, _u64, _vp, _u32, _i, _i,
                              _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _sz, _vp, C.c_double, _vp, _vp]),
    "daco_sample_backward": (_i, [_vp, _i, _i, _i, _i, _vp, _l, _vp, _l, _f, _f, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp,
                                  C.c_double]),
    "daco_gnn_param_floats": (_sz, [_i]),
    "daco_gnn_workspace_bytes": (_sz, [_i, _i]),
    "daco_gnn_forward": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "daco_gnn_train_workspace_bytes": (_sz, [_i, _i, _i]),
    "daco_gnn_train_forward": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "daco_gnn_train_backward": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz]),
    "daco_cvrp_local_search": (_i, [_vp, _i, _i, _i, _i, _vp, _l, _vp, _f, _vp, _i, _vp, _vp]),
    "daco_hgs_table_bytes": (_sz, [_i, _i]),
    "daco_hgs_prepare": (_i, [_vp, _i, _i, _vp, _l, _i, _vp]),
    "daco_hgs_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "daco_hgs_local_search": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _i, _vp, _vp, _vp, _vp, _sz]),
    "daco_sibling_workspace_bytes": (_sz, [_i, _i, _i]),
    "daco_sibling_sample": (_i, [_vp, _i, _i, _i, _i, _vp, _l, _vp, _l, _f, _f, _vp, _vp, _l, _f, _vp, _i, _i, _vp, _vp,
                                 _i, _u64, _u64, _u32, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "daco_sibling_backward": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _l, _vp, _l, _f, _f, _vp, _vp, _l, _f, _vp, _i, _vp, _vp,
                                   _vp, _vp, _vp]),
    "daco_tsp_knn_graph": (_i, [_vp, _i, _i, _i, _vp, _f, _vp, _vp, _vp, _vp]),
    "daco_tsp_knn_graph_csr": (_i, [_vp, _i, _i, _i, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "daco_heu_matrix": (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _f, _vp, _vp]),
    "daco_two_opt": (_i, [_vp, _i, _i, _i, _vp, _vp, _l, _vp, _l, _vp]),
    "daco_two_opt_tables_bytes": (_sz, [_i, _i]),
    "daco_two_opt_prepare": (_i, [_vp, _i, _i, _vp, _l, _vp, _sz]),
    "daco_two_opt_nbr": (_i, [_vp, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _l, _vp]),
    "daco_two_opt_auto": (_i, [_vp, _i, _i, _i, _vp, _vp, _l, _vp, _vp, _vp, _l, _vp]),
    "daco_tsp_nls": (_i, [_vp, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _l, _vp, _vp, _vp, _l, _i, _l, _vp, _vp, _vp]),
}


ABI_VERSION = 126          # include/deepaco_hip.h DACO_VERSION this table was written against


class DacoError(RuntimeError):
    pass


def lib():
    """Load the shared library (once).  No fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DacoError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(make -C deepaco_amd/csrc, or __graft_entry__.build()). deepaco_amd has no CPU fallback.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        if h.daco_version() != ABI_VERSION:     # a stale .so would be called with the wrong argument lists
            raise DacoError(f"{LIB_PATH} has ABI version {h.daco_version()}, this package binds {ABI_VERSION}: "
                            "rebuild it (make -C deepaco_amd/csrc)")
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().daco_last_error().decode("utf-8", "replace")
        if rc == -2:
            raise ValueError(f"{what}: {msg}")
        raise DacoError(f"{what} failed (code {rc}): {msg}")
