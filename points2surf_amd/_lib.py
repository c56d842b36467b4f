"""ctypes binding of libp2s_hip.so (include/p2s_hip.h).  There is NO fallback: if the library is
missing or a call fails, an exception is raised."""
import ctypes
import os

from .weights import ModelCfg, WeightOffsets

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('P2S_LIB_PATH') or os.path.join(HERE, 'libp2s_hip.so')   # override: development builds

P2S_OK = 0
P2S_ECAPACITY = -4

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_int64 = ctypes.c_int64


class Counters(ctypes.Structure):
    """mirror of ``p2s_counters``"""
    _fields_ = [('ms_chain_stn', ctypes.c_double), ('ms_stn_head', ctypes.c_double),
                ('ms_chain_main', ctypes.c_double), ('ms_decoder', ctypes.c_double),
                ('ms_knn', ctypes.c_double), ('ms_subsample', ctypes.c_double), ('ms_grid', ctypes.c_double),
                ('queries', ctypes.c_int64), ('launches_chain', ctypes.c_int64),
                ('ms_chain_qstn', ctypes.c_double), ('fallback_queries', ctypes.c_int64),
                ('reserved', ctypes.c_double * 6)]


# name -> (restype, argtypes): every symbol include/p2s_hip.h declares
PROTOTYPES = {
    'p2s_abi_version': (c_int, []),
    'p2s_last_error': (ctypes.c_char_p, []),
    'p2s_device_count': (c_int, []),
    'p2s_release_scratch': (c_int, [c_int]),
    'p2s_model_create': (c_int, [ctypes.POINTER(ModelCfg), c_void_p, ctypes.c_size_t,
                                 ctypes.POINTER(WeightOffsets), c_int, ctypes.POINTER(c_void_p)]),
    'p2s_model_destroy': (c_int, [c_void_p]),
    'p2s_encode_decode': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                  c_void_p]),
    'p2s_encode_features': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'p2s_cloud_create': (c_int, [c_void_p, c_int, c_int, c_void_p, ctypes.POINTER(c_void_p)]),
    'p2s_cloud_destroy': (c_int, [c_void_p]),
    'p2s_cloud_num_points': (c_int, [c_void_p]),
    'p2s_cloud_index_export': (c_int, [c_void_p, ctypes.POINTER(ctypes.c_int32), c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    'p2s_query_grid': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, ctypes.POINTER(c_int64), c_void_p]),
    'p2s_knn_patch': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'p2s_knn_patch_set': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    'p2s_rng_create': (c_int, [ctypes.c_uint32, c_int, ctypes.POINTER(c_void_p)]),
    'p2s_rng_destroy': (c_int, [c_void_p]),
    'p2s_rng_get_state': (c_int, [c_void_p, c_void_p, ctypes.POINTER(ctypes.c_int32), c_void_p]),
    'p2s_rng_set_state': (c_int, [c_void_p, c_void_p, ctypes.c_int32, c_void_p]),
    'p2s_rng_set_jump_tables': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int]),
    'p2s_rng_check': (c_int, [c_void_p, c_void_p]),
    'p2s_subsample_uniform': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    'p2s_subsample_weighted': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    'p2s_subsample_fixed': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, ctypes.c_uint32, c_void_p, c_void_p, c_void_p]),
    'p2s_subsample_shuffle_pad': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    'p2s_patch_from_ids': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    'p2s_gather_points': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    'p2s_infer_shape': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int, c_void_p,
                                c_void_p, ctypes.POINTER(c_int64), c_void_p]),
    'p2s_infer_shape_ball': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int, c_void_p,
                                     c_void_p, ctypes.POINTER(c_int64), c_void_p]),
    'p2s_kd_order_host': (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, ctypes.POINTER(ctypes.c_int32)]),
    'p2s_ball_count': (c_int, [c_void_p, c_void_p, c_int64, ctypes.c_double, c_void_p, c_void_p]),
    'p2s_ball_patch': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, ctypes.c_double, c_int, c_int, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p]),
    'p2s_infer_queries': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    'p2s_random_rotations': (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    'p2s_rotate_points': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    'p2s_debug_fault_chunk': (c_int, [c_void_p, c_int]),
    'p2s_sdf_volume': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, ctypes.c_float, c_int, c_int, c_void_p,
                               ctypes.POINTER(ctypes.c_int32), c_void_p]),
    'p2s_marching_cubes': (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, ctypes.POINTER(c_int64),
                                   ctypes.POINTER(c_int64), c_int, c_int, ctypes.POINTER(c_int), c_int, c_void_p]),
    'p2s_mc_cell': (c_int, [c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), c_void_p]),
    'p2s_rng_random_sample': (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    'p2s_mesh_sample_surface': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                        ctypes.POINTER(ctypes.c_double), c_int, c_void_p]),
    'p2s_points_remove_close': (c_int, [c_void_p, c_int64, ctypes.c_double, c_int64, c_void_p, ctypes.POINTER(c_int64), c_int,
                                        c_void_p]),
    'p2s_nn_distance_stats': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, ctypes.POINTER(ctypes.c_double),
                                      ctypes.POINTER(ctypes.c_double), c_void_p]),
    'p2s_set_profiling': (c_int, [c_void_p, c_int]),
    'p2s_get_counters': (c_int, [c_void_p, ctypes.POINTER(Counters)]),
    'p2s_model_capture_logits': (c_int, [c_void_p, c_void_p, c_int64]),
    'p2s_write_txt_f32': (c_int, [ctypes.c_char_p, c_void_p, c_int64]),
    'p2s_write_query_vis_ply': (c_int, [ctypes.c_char_p, c_void_p, c_void_p, c_int64]),
    'p2s_write_coff_samples': (c_int, [ctypes.c_char_p, c_void_p, c_void_p, c_int64]),
}


class P2SError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('libp2s_hip error %d: %s' % (code, msg))
        self.code = code


_lib = None


def load():
    """Load the HIP engine.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            'libp2s_hip.so not found at %s -- build it with `python -m points2surf_amd.build` '
            '(hipcc --offload-arch=gfx950); there is no CPU/PyTorch fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)        # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, allow=()):
    if rc != P2S_OK and rc not in allow:
        msg = load().p2s_last_error()
        raise P2SError(rc, msg.decode('utf-8', 'replace') if msg else '')
    return rc
