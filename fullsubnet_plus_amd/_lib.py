"""ctypes binding of include/fsnp.h + include/fsnp_debug.h (cffi is not installed in this image; ctypes is stdlib)."""
import ctypes
import os

from . import _build

c_i32, c_i64, c_f32p, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.POINTER(ctypes.c_float), ctypes.c_void_p


class FsnpConfig(ctypes.Structure):
    """struct fsnp_config (include/fsnp.h)."""
    _fields_ = [
        ("num_freqs", c_i32), ("look_ahead", c_i32), ("sb_num_neighbors", c_i32), ("fb_num_neighbors", c_i32),
        ("tcn_hidden", c_i32), ("num_tcn_blocks", c_i32), ("sb_hidden", c_i32), ("output_size", c_i32),
        ("norm_type", c_i32), ("fb_act", c_i32), ("sb_act", c_i32), ("kersize", c_i32 * 3),
        ("num_groups_in_drop_band", c_i32), ("attention", c_i32), ("model", c_i32), ("sequence_model", c_i32),
        ("subband_num", c_i32),
    ]


NORM_TYPES = {"offline_laplace_norm": 0, "cumulative_laplace_norm": 1, "offline_gaussian_norm": 2,
              "cumulative_layer_norm": 3}
ACTIVATIONS = {None: 0, False: 0, "": 0, "ReLU": 1, "ReLU6": 2, "Tanh": 3}
ATTENTION = {"TSSE": 0, "SE": 1, "ECA": 2, "CBAM": 3}
SEQUENCE_MODELS = {"LSTM": 0, "GRU": 1, "TCN": 2}
MODE_FULL, MODE_PARITY = 0, 1
NUM_COSTS = 27           # FSNP_NUM_COSTS: values of the planner's flat cost table (fsnp_get_costs)
MODEL_FULLSUBNET_PLUS, MODEL_FULLSUBNET = 0, 1
BOX_PROBE_VALUES = 9     # FSNP_BOX_PROBE_VALUES (include/fsnp_debug.h)

# every symbol include/fsnp.h and include/fsnp_debug.h declare: name -> (restype, argtypes)
SYMBOLS = {
    "fsnp_create": (c_i32, [ctypes.POINTER(FsnpConfig), ctypes.POINTER(c_vp)]),
    "fsnp_destroy": (None, [c_vp]),
    "fsnp_set_weight": (c_i32, [c_vp, ctypes.c_char_p, c_vp, c_i64]),
    "fsnp_commit_weights": (c_i32, [c_vp]),
    "fsnp_num_weights": (c_i32, [c_vp]),
    "fsnp_weight_info": (c_i32, [c_vp, c_i32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_i64)]),
    "fsnp_workspace_bytes": (ctypes.c_size_t, [c_vp, c_i32, c_i32, c_i32]),
    "fsnp_forward": (c_i32, [c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(c_i64 * 3 * 3), c_vp, c_i32, c_i32, c_i32,
                             c_i32, c_i32, c_vp]),
    "fsnp_forward_complex": (c_i32, [c_vp, c_vp, ctypes.POINTER(c_i64 * 3), c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "fsnp_stft": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_vp]),
    "fsnp_istft": (c_i32, [c_vp, c_vp, ctypes.POINTER(c_i64 * 3), c_vp, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "fsnp_enhance_wave": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "fsnp_apply_cirm": (c_i32, [c_vp, c_vp, ctypes.POINTER(c_i64 * 3), c_vp, ctypes.POINTER(c_i64 * 3), c_i32, c_i32,
                                c_i32, c_vp]),
    "fsnp_norm": (c_i32, [c_i32, c_vp, ctypes.POINTER(c_i64 * 4), c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "fsnp_unfold": (c_i32, [c_vp, ctypes.POINTER(c_i64 * 4), c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "fsnp_channel_attention": (c_i32, [c_vp, c_i32, c_vp, ctypes.POINTER(c_i64 * 3), c_vp, c_i32, c_i32, c_vp]),
    "fsnp_fullband_model": (c_i32, [c_vp, c_i32, c_vp, ctypes.POINTER(c_i64 * 3), c_vp, c_i32, c_i32, c_vp]),
    "fsnp_lstm2_fc": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "fsnp_read_stage": (c_i32, [c_vp, ctypes.c_char_p, c_vp, c_i64]),
    "fsnp_set_timing": (c_i32, [c_vp, c_i32]),
    "fsnp_get_timing": (c_i32, [c_vp, ctypes.POINTER(ctypes.c_double * 4), ctypes.POINTER(c_i64 * 4), c_i32]),
    "fsnp_describe_plan": (c_i32, [c_vp, c_i32, c_i32, ctypes.POINTER(c_i32), c_i32]),
    "fsnp_describe_plan_ex": (c_i32, [c_vp, c_i32, c_i32, ctypes.POINTER(c_i32), c_i32]),
    "fsnp_reserve": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "fsnp_dump_config": (ctypes.c_int64, [c_vp, ctypes.c_char_p, ctypes.c_int64]),
    "fsnp_get_costs": (c_i32, [c_vp, ctypes.POINTER(ctypes.c_double * NUM_COSTS), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "fsnp_debug_plan_rows": (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i32, ctypes.c_double, ctypes.POINTER(c_i32), c_i32]),
    "fsnp_measure_costs": (c_i32, [c_vp, ctypes.POINTER(ctypes.c_double * NUM_COSTS)]),
    "fsnp_debug_set_costs": (c_i32, [c_vp, ctypes.POINTER(ctypes.c_double), c_i32]),
    "fsnp_debug_plan_rows2": (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i32, ctypes.c_double, c_i32, ctypes.POINTER(ctypes.c_double),
                              ctypes.POINTER(c_i32), c_i32]),
    "fsnp_forward_flops": (ctypes.c_double, [c_vp, c_i32, c_i32, c_i32]),
    "fsnp_lstm_flops": (ctypes.c_double, [c_vp, c_i64, c_i32]),
    "fsnp_debug_lstm_profile": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_i64]),
    "fsnp_debug_pp_profile": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_i64]),
    "fsnp_set_precision": (c_i32, [c_vp, c_i32]),
    "fsnp_check_errors": (c_i32, [c_vp]),
    "fsnp_poll_errors": (c_i32, [c_vp]),
    "fsnp_set_pipeline": (c_i32, [c_vp, c_i32]),
    "fsnp_flush": (c_i32, [c_vp, c_vp]),
    "fsnp_watch_weights": (c_i32, [c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_i64), c_i32, c_i32, c_vp]),
    "fsnp_set_verify": (c_i32, [c_vp, c_i32]),
    "fsnp_set_verify_sample": (c_i32, [c_vp, c_i32]),
    "fsnp_verify_count": (c_i64, [c_vp]),
    "fsnp_debug_verify_sample_stats": (c_i32, [c_vp, ctypes.POINTER(c_i64 * 3)]),
    "fsnp_debug_corrupt_exchange": (c_i32, [c_vp, c_i32]),
    "fsnp_abi_version": (c_i32, []),
    "fsnp_config_size": (c_i32, []),
    "fsnp_debug_set_lstm_coop": (c_i32, [c_vp, c_i32]),
    "fsnp_debug_set_gemm_dma": (c_i32, [c_vp, c_i32]),
    "fsnp_debug_inject_error": (c_i32, [c_vp]),
    "fsnp_debug_set_chaos": (c_i32, [c_vp, c_i32]),
    "fsnp_debug_set_lstm_waves": (c_i32, [c_vp, c_i32]),
    "fsnp_debug_set_num_cus": (c_i32, [c_vp, c_i32]),
    "fsnp_debug_lstm_pack": (c_i32, [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64]),
    "fsnp_debug_lstm_coop_pack": (c_i32, [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64]),
    "fsnp_debug_lstm_coopw_pack": (c_i32, [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64]),
    "fsnp_debug_lstm_hpw_pack": (c_i32, [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64]),
    "fsnp_debug_lstm_fbv_pack": (c_i32, [c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64]),
    "fsnp_debug_box_probe": (c_i32, [ctypes.c_double, ctypes.POINTER(ctypes.c_double * BOX_PROBE_VALUES), c_vp]),
    "fsnp_debug_launch_clock": (c_i32, [c_vp, ctypes.POINTER(ctypes.c_double * 7)]),
    "fsnp_last_error": (ctypes.c_char_p, []),
    "fsnp_version": (ctypes.c_char_p, []),
}

ABI_VERSION = 10         # FSNP_ABI_VERSION of the include/fsnp.h these signatures were written against

_lib = None


def load(build_if_missing=True):
    """dlopen libfsnp_hip.so (building it first if the sources are newer) and type every entry point.
    Raises if the extension cannot be built/loaded - the product path never degrades to a CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if build_if_missing and (not os.path.exists(path) or _build.is_stale()):
        # a stale library is never bound silently: its entry points may no longer match the signatures above
        if not os.access(_build.HERE, os.W_OK):
            raise RuntimeError(f"{path} is {'stale' if os.path.exists(path) else 'missing'} and {_build.HERE} is read-only: "
                               "rebuild with `python -m fullsubnet_plus_amd._build`")
        _build.build()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -m fullsubnet_plus_amd._build`")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.fsnp_abi_version() != ABI_VERSION or lib.fsnp_config_size() != ctypes.sizeof(FsnpConfig):
        raise RuntimeError(f"{path}: ABI mismatch (library: version {lib.fsnp_abi_version()}, fsnp_config {lib.fsnp_config_size()} "
                           f"bytes; binding: version {ABI_VERSION}, {ctypes.sizeof(FsnpConfig)} bytes) - rebuild the library")
    _lib = lib
    return lib


def last_error():
    return load().fsnp_last_error().decode(errors="replace")


ERR_TIMEOUT, ERR_STALE_WEIGHTS, ERR_VERIFY = 5, 6, 7      # device-side conditions noticed by a later call (include/fsnp.h)


class FsnpError(RuntimeError):
    """RuntimeError that carries the C ABI's return code."""

    def __init__(self, msg, code):
        super().__init__(msg)
        self.code = code


def check(rc, what):
    if rc != 0:
        raise FsnpError(f"{what} failed (code {rc}): {last_error()}", rc)
