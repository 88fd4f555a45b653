"""ctypes binding of libtfx_hip.so (include/tfx.h).  There is NO fallback: if the HIP library is
missing or fails to load, importing the env raises - the step path is the GPU kernels or nothing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TFX_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libtfx_hip.so"))

MAX_ARCH = 64
ACTION_BUFFER, ACTION_BROADCAST, ACTION_CYCLE, ACTION_GREEDY = 0, 1, 2, 3
SPAWN_NONE, SPAWN_COUNTS, SPAWN_PERIODIC = 0, 1, 2
ABI_VERSION = 11


class TfxConfig(C.Structure):
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("capacity", C.c_int32), ("n_envs", C.c_int32),
                ("planes", C.c_int32), ("length", C.c_float), ("rate", C.c_float),
                ("car_v", C.c_float), ("car_l", C.c_float), ("car_a", C.c_float),
                ("car_delta", C.c_float), ("car_v0", C.c_float), ("car_b", C.c_float),
                ("car_T", C.c_float), ("car_s0", C.c_float),
                ("yellow_ticks", C.c_int32), ("thresh", C.c_float), ("detect_dist", C.c_float),
                ("overflow_penalty", C.c_float), ("eps", C.c_float),
                ("learn_switch", C.c_int32), ("validate", C.c_int32), ("entry_spec", C.c_uint32),
                ("env_id_offset", C.c_int32), ("layout", C.c_int32),
                ("n_archetypes", C.c_int32), ("arch", (C.c_float * 8) * MAX_ARCH)]


class TfxBuffers(C.Structure):
    _fields_ = [("xv", C.c_void_p), ("w", C.c_void_p), ("leading", C.c_void_p), ("lastcar", C.c_void_p),
                ("obs", C.c_void_p), ("rewards", C.c_void_p), ("waiting", C.c_void_p),
                ("passed_dst", C.c_void_p), ("done_tick", C.c_void_p),
                ("trip_times", C.c_void_p), ("n_trips", C.c_void_p), ("trip_cap", C.c_int32)]


class TfxError(RuntimeError):
    pass


_lib = None

_PROTOS = {
    "tfx_abi_version": (C.c_int, []),
    "tfx_last_error": (C.c_char_p, []),
    "tfx_create": (C.c_int, [C.POINTER(TfxConfig), C.POINTER(C.c_void_p)]),
    "tfx_destroy": (C.c_int, [C.c_void_p]),
    "tfx_dims": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_int32)] * 4),
    "tfx_tables": (C.c_int, [C.c_void_p] + [C.c_void_p] * 4),
    "tfx_bind_buffers": (C.c_int, [C.c_void_p, C.POINTER(TfxBuffers)]),
    "tfx_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "tfx_reset_envs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tfx_refresh": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tfx_set_actions": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]),
    "tfx_set_spawns": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]),
    "tfx_set_poisson": (C.c_int, [C.c_void_p, C.c_double, C.c_uint64, C.c_void_p, C.c_int32]),
    "tfx_set_regular": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_uint64]),
    "tfx_set_spawn_archetypes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "tfx_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "tfx_move_cars": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tfx_advance_finished_cars": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tfx_agent_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tfx_remi": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tfx_cars_on_roads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "tfx_done": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "tfx_get_tick": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "tfx_set_tick": (C.c_int, [C.c_void_p, C.c_int32]),
    "tfx_vehicle_updates": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]),
    "tfx_reset_counters": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tfx_profile": (C.c_int, [C.c_void_p, C.c_int32]),
    "tfx_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "tfx_xv_pairs": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "tfx_export_ring": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tfx_import_ring": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tfx_fastdiv_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    "tfx_launch_info": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_int32)] * 3),
    "tfx_arrivals_replay": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "tfx_fused_ticks": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "tfx_pair_ticks": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "tfx_tail_ticks": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "tfx_slow_pairs": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]),
    "tfx_split_ticks": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "tfx_step_kernel": (C.c_char_p, [C.c_void_p]),
    "tfx_debug_fail_after": (C.c_int, [C.c_void_p, C.c_int32]),
}


def lib():
    """The loaded library; raises TfxError if it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TfxError("HIP library not found at %s - build it with `python -c 'import "
                           "__graft_entry__ as g; g.build()'` (or make -C traffic-env_amd/csrc). "
                           "There is no CPU fallback for the env step." % LIB_PATH)
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as exc:
            raise TfxError("cannot load %s: %s" % (LIB_PATH, exc))
        for name, (res, argtypes) in _PROTOS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = argtypes
        if handle.tfx_abi_version() != ABI_VERSION:
            raise TfxError("libtfx_hip.so ABI %d != binding ABI %d" % (handle.tfx_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise TfxError("tfx error %d: %s" % (rc, lib().tfx_last_error().decode("utf-8", "replace")))
    return rc
