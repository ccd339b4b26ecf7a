"""ctypes binding of libwfl.so (C ABI declared in include/wfl.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C gtn_applications_amd/csrc`.
There is NO fallback: if the shared object is missing or a symbol cannot be resolved the import
fails loudly -- the product path never routes around the HIP extension.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_uint8, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WFL_LIB_PATH") or os.path.join(_HERE, "libwfl.so")  # (override: A/B builds of the kernels)

WFL_OK = 0
ERR_INVALID, ERR_UNSUPPORTED, ERR_RUNTIME = 1, 2, 3
EPSILON = -1
SEMIRING_LOG, SEMIRING_TROPICAL = 0, 1
DENSE_MAIN, DENSE_REPAIR, DENSE_REDUCE, DENSE_ALL = 1, 2, 4, 7  # parts of wfl_dense_forward_parts / wfl_dense_grad_parts
CTC_WS_REJECTED, CTC_WS_STATUS, CTC_WS_LOG2Z, CTC_WS_ZRANGE, CTC_WS_DEBUG, CTC_WS_CLOCK = 0, 1, 2, 3, 4, 5
DENSE_WS_FLAGS = 0  # wfl_dense_workspace_field (include/wfl.h: WFL_DENSE_WS_FLAGS)
CONV_SPIKE, CONV_BLANK_OPTIONAL = 1, 2


class WflError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libwfl error {code}: {message}")
        self.code = code


class WflUnsupported(WflError):
    pass


class CtcCall(ctypes.Structure):
    """Mirror of `wfl_ctc_call` (include/wfl.h)."""

    _fields_ = [("n_labels", c_int64), ("host_state", c_void_p)]


class LatticeDesc(ctypes.Structure):
    """Mirror of `wfl_lattice_desc` (include/wfl.h) -- field order and types must match."""

    _fields_ = (
        [(n, c_int32) for n in ("B", "max_states", "max_arcs", "max_eps", "max_labels", "max_levels")]
        + [(n, c_int64) for n in ("total_states", "total_arcs", "total_eps", "total_labels")]
        + [("shared", c_int32)]
        + [
            (n, c_int64)
            for n in (
                "state_off", "arc_off", "eps_off", "lab_off", "lvl_off", "in_ptr", "out_ptr", "out_arc",
                "ein_ptr", "eout_ptr", "eout_arc", "arc_src", "arc_dst", "arc_slot", "arc_lab", "arc_wid",
                "eps_src", "eps_dst", "eps_wid", "labels", "lvl_ptr", "arc_orig", "eps_orig", "slot_ptr", "slot_arc", "int_words",
                "arc_w", "eps_w", "start_w", "accept_w", "float_words",
            )
        ]
    )


def _load():
    # torch wheels bundle their own HIP runtime (torch/lib/libamdhip64.so).  Import torch first so
    # that libwfl.so binds to THAT runtime instance: two HIP runtimes in one process do not share
    # devices, streams or allocations ("no ROCm-capable device is detected" at the first launch).
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C gtn_applications_amd/csrc`. There is no CPU fallback."
        )
    return ctypes.CDLL(LIB_PATH)


lib = _load()

_P = c_void_p  # device pointers and opaque handles travel as void*
_SIGS = {
    "wfl_last_error": (c_char_p, []),
    "wfl_version": (c_int, []),
    "wfl_free": (None, [_P]),
    # host graph library
    "wfl_graph_new": (_P, []),
    "wfl_graph_free": (None, [_P]),
    "wfl_graph_clone": (_P, [_P]),
    "wfl_graph_add_node": (c_int, [_P, c_int, c_int]),
    "wfl_graph_add_arc": (c_int, [_P, c_int, c_int, c_int, c_int, c_float]),
    "wfl_graph_add_nodes": (c_int, [_P, c_int, _P, _P]),
    "wfl_graph_add_arcs": (c_int, [_P, c_int64, _P, _P, _P, _P, _P]),
    "wfl_graph_num_nodes": (c_int, [_P]),
    "wfl_graph_num_arcs": (c_int64, [_P]),
    "wfl_graph_get": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "wfl_graph_set_weights": (c_int, [_P, _P]),
    "wfl_graph_arc_sort": (c_int, [_P, c_int]),
    "wfl_graph_compose": (_P, [_P, _P, POINTER(_P), POINTER(_P)]),
    "wfl_graph_token_alignments": (_P, [_P, _P]),
    "wfl_host_pool_wake": (None, []),
    "wfl_graph_remove": (_P, [_P, c_int, c_int, POINTER(_P)]),
    "wfl_graph_project": (_P, [_P, c_int]),
    "wfl_graph_viterbi_path": (_P, [_P]),
    "wfl_graph_equal": (c_int, [_P, _P]),
    "wfl_graph_isomorphic": (c_int, [_P, _P]),
    "wfl_graph_loadtxt": (_P, [c_char_p]),
    "wfl_graph_savetxt": (c_int, [_P, c_char_p]),
    "wfl_graph_load": (_P, [c_char_p]),
    "wfl_graph_save": (c_int, [_P, c_char_p]),
    # lattice packing
    "wfl_lattice_pack": (_P, [_P, _P, c_int, c_int, c_int, c_int]),
    "wfl_transducer_pack_batch": (_P, [_P, _P, _P, _P, _P, c_int, c_int, c_int]),
    "wfl_transducer_pack_batch_into": (_P, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_int64, c_int64]),
    "wfl_lattice_host_external": (c_int64, [_P]),
    "wfl_transducer_decode_batch": (c_int, [_P, _P, _P, c_int, _P, c_int64, _P, c_int]),
    "wfl_lattice_pack_ctc": (_P, [_P, _P, c_int, c_int, c_int]),
    "wfl_lattice_pack_asg_fal": (_P, [_P, _P, c_int, c_int]),
    "wfl_lattice_pack_stc": (_P, [_P, _P, c_int, c_int, c_float, c_int]),
    "wfl_lattice_host_free": (None, [_P]),
    "wfl_lattice_host_desc": (POINTER(LatticeDesc), [_P]),
    "wfl_lattice_host_ints": (_P, [_P]),
    "wfl_lattice_host_floats": (_P, [_P]),
    # device: generic lattice engine
    "wfl_lattice_workspace": (c_int, [POINTER(LatticeDesc), c_int, POINTER(c_int64), POINTER(c_int64)]),
    "wfl_lattice_formats_offset": (c_int, [POINTER(LatticeDesc), c_int, POINTER(c_int64)]),
    "wfl_lattice_gather": (c_int, [POINTER(LatticeDesc), _P, _P, c_int, c_int, _P, _P, _P]),
    "wfl_lattice_forward": (c_int, [POINTER(LatticeDesc), _P, _P, _P, c_int, _P, c_int, _P, _P, _P, _P, _P]),
    "wfl_lattice_grad": (
        c_int,
        [POINTER(LatticeDesc), _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P],
    ),
    "wfl_lattice_forward_grad": (
        c_int, [POINTER(LatticeDesc), _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, POINTER(c_int), _P]),
    "wfl_lattice_side_join": (c_int, [_P]),
    "wfl_lattice_diagnostics": (c_int, [_P, c_int]),
    "wfl_lattice_grad_rest": (
        c_int, [POINTER(LatticeDesc), _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "wfl_lattice_backtrace": (c_int, [POINTER(LatticeDesc), _P, _P, _P, _P, c_int, _P, _P, c_int, _P]),
    "wfl_debug_grad_occupancy": (c_int, [c_int]),
    # device: ConvTransduce1D
    "wfl_stc_augment": (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P, _P, _P]),
    "wfl_stc_augment_grad": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, _P, _P, _P, _P]),
    "wfl_conv_forward": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P]),
    "wfl_conv_grad": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P,
                              _P]),
    # device: dense transitions
    "wfl_dense_forward": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "wfl_dense_forward_parts": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P]),
    "wfl_dense_max_classes": (c_int, []),
    "wfl_dense_on_chip_classes": (c_int, []),
    "wfl_dense_workspace_field": (c_int, [c_int, c_int, c_int, POINTER(c_int64), POINTER(c_int64)]),
    "wfl_dense_workspace": (c_int, [c_int, c_int, c_int, POINTER(c_int64), POINTER(c_int64)]),
    "wfl_dense_grad": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "wfl_dense_grad_parts": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "wfl_dense_viterbi": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    # device: CTC fast path
    "wfl_ctc_workspace": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_int64)]),
    "wfl_ctc_workspace_field": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_int64), POINTER(c_int64)]),
    "wfl_ctc_forward": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "wfl_ctc_grad": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "wfl_ctc_forward_backward": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P,
                                         _P, _P]),
    "wfl_ctc_forward_backward_call": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P,
                                              _P, _P, _P]),
    "wfl_row_lse": (c_int, [_P, c_int64, c_int, _P, _P]),
    "wfl_row_argmax": (c_int, [_P, c_int64, c_int, _P, _P]),
    "wfl_upload": (c_int, [_P, _P, c_int64, _P]),
    "wfl_reduce_loss": (c_int, [_P, _P, _P, c_int, c_float, c_int, _P, _P]),
    "wfl_scale": (c_int, [_P, c_int64, _P, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here == the library does not export what wfl.h declares
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    msg = lib.wfl_last_error()
    return msg.decode() if msg else ""


def check(rc):
    """Raise for a non-zero status code of an int-returning entry point."""
    if rc != WFL_OK:
        cls = WflUnsupported if rc == ERR_UNSUPPORTED else WflError
        raise cls(rc, last_error())


def check_handle(h):
    if not h:
        raise WflError(ERR_INVALID, last_error())
    return h
