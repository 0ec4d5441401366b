"""ctypes binding of liblcf.so (include/lcf.h).  Fails loudly when the library is missing:
there is no CPU fallback in the product path."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblcf.so")
if os.environ.get("LCF_LIB"):  # an experimental build of the same library (A/B tools)
    LIB_PATH = os.path.abspath(os.environ["LCF_LIB"])

c_int = ctypes.c_int
c_size_t = ctypes.c_size_t
c_void_p = ctypes.c_void_p
_I3 = ctypes.c_int * 3
_F3 = ctypes.c_float * 3
_F6 = ctypes.c_float * 6

class RbLevel(ctypes.Structure):
    """lcf_rb_level (include/lcf.h)."""
    _fields_ = [("subm", c_int), ("ksize", _I3), ("stride", _I3), ("pad", _I3), ("dil", _I3),
                ("out_nbr", c_void_p), ("out_coors", c_void_p), ("cap_out", c_int),
                ("k_out", c_void_p), ("overflow", c_void_p), ("ws", c_void_p),
                ("ws_bytes", c_size_t), ("ready_event", c_void_p), ("plan", c_void_p),
                ("plan_tile_rows", c_int)]


# every symbol include/lcf.h declares: (restype, argtypes)
_PI3 = ctypes.POINTER(ctypes.c_int)
_PF = ctypes.POINTER(ctypes.c_float)
SYMBOLS = {
    "lcf_strerror": (ctypes.c_char_p, [c_int]),
    "lcf_version": (c_int, []),
    "lcf_set_option": (c_int, [c_int, c_int]),
    "lcf_ctx_create": (c_int, [ctypes.POINTER(c_void_p)]),
    "lcf_ctx_destroy": (None, [c_void_p]),
    "lcf_ctx_set_option": (c_int, [c_void_p, c_int, c_int]),
    "lcf_ctx_fill_batch_begin": (c_int, [c_void_p]),
    "lcf_ctx_fill_batch_flush": (c_int, [c_void_p, c_void_p]),
    "lcf_voxelize_workspace_bytes": (c_size_t, [c_int, c_int]),
    "lcf_voxelize_workspace_reset": (c_int, [c_void_p, c_size_t, c_int, c_int, c_void_p,
                                             c_void_p]),
    "lcf_voxelize": (c_int, [c_void_p, c_int, c_void_p, c_int, _PF, _PF, c_int, c_int, c_int,
                             c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_size_t, c_void_p, c_void_p]),
    "lcf_voxelize_append": (c_int, [c_void_p, c_int, c_int, c_int, c_int, _PF, _PF, c_int, c_int,
                                    c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_size_t, c_void_p, c_void_p]),
    "lcf_voxelize_scan_fallbacks": (c_int, [c_void_p, c_size_t, c_int, c_int,
                                            ctypes.POINTER(c_void_p)]),
    "lcf_voxelize_static_max_bytes": (c_size_t, []),
    "lcf_voxelize_static": (c_int, [c_void_p, c_int, c_void_p, c_int, _PF, _PF, c_int, c_int,
                                    c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "lcf_rulebook_subm_workspace_bytes": (c_size_t, [c_int]),
    "lcf_rulebook_subm_workspace_reset": (c_int, [c_void_p, c_size_t, c_int, c_void_p,
                                                  c_void_p]),
    "lcf_rulebook_subm": (c_int, [c_void_p, c_int, c_void_p, c_int, _PI3, _PI3, _PI3, c_void_p,
                                  c_void_p, c_size_t, c_void_p, c_void_p]),
    "lcf_rulebook_conv_max_out": (c_int, [c_int, _PI3, _PI3]),
    "lcf_conv_out_shape": (None, [_PI3, _PI3, _PI3, _PI3, _PI3, _PI3]),
    "lcf_rulebook_conv_workspace_bytes": (c_size_t, [c_int, _PI3, _PI3]),
    "lcf_rulebook_conv_workspace_reset": (c_int, [c_void_p, c_size_t, c_int, _PI3, _PI3,
                                                  c_void_p, c_void_p]),
    "lcf_rulebook_conv_discover": (c_int, [c_void_p, c_int, c_void_p, c_int, _PI3, _PI3, _PI3,
                                           _PI3, _PI3, c_void_p, c_int, c_void_p, c_void_p,
                                           c_void_p, c_size_t, c_void_p, c_void_p]),
    "lcf_rulebook_conv_fill": (c_int, [c_void_p, c_int, c_void_p, c_int, _PI3, _PI3, _PI3, _PI3,
                                       _PI3, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_size_t, c_void_p, c_void_p]),
    "lcf_rulebook_chain_workspace_bytes": (c_size_t, [c_int, _PI3, _PI3, c_int]),
    "lcf_rulebook_chain_reset": (c_int, [ctypes.POINTER(RbLevel), c_int, c_int, c_void_p,
                                         c_size_t, c_void_p, c_void_p]),
    "lcf_rulebook_chain": (c_int, [c_void_p, c_int, c_void_p, c_int, _PI3,
                                   ctypes.POINTER(RbLevel), c_int, c_void_p, c_size_t, c_void_p,
                                   c_void_p]),
    "lcf_rulebook_chain_num_parts": (c_int, [ctypes.POINTER(RbLevel), c_int]),
    "lcf_rulebook_chain_parts": (c_int, [c_void_p, c_int, c_void_p, c_int, _PI3,
                                         ctypes.POINTER(RbLevel), c_int, c_void_p, c_size_t,
                                         c_int, c_int, c_void_p, c_void_p]),
    "lcf_rulebook_mirror": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "lcf_rulebook_pairs_workspace_bytes": (c_size_t, [c_int, c_int]),
    "lcf_rulebook_pairs": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                   c_size_t, c_void_p]),
    "lcf_weight_image_elems": (c_size_t, [c_int, c_int, c_int, c_int]),
    "lcf_weight_image": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "lcf_indice_conv_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                    c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                    c_void_p]),
    "lcf_weight_image_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "lcf_indice_conv_fwd_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                         c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                         c_void_p, c_void_p]),
    "lcf_weight_image_bf16x3": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                        c_void_p]),
    "lcf_indice_conv_fwd_bf16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                           c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                           c_void_p, c_int, _PI3, c_void_p, c_void_p]),
    "lcf_weight_image_bf16x8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                        c_void_p]),
    "lcf_indice_conv_fwd_bf16x8": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                           c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                           c_void_p, c_int, _PI3, c_void_p, c_void_p]),
    "lcf_indice_conv_fwd_dense": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                          c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                          c_int, _PI3, c_void_p, c_void_p, c_void_p]),
    "lcf_conv_tile_plan_ints": (c_size_t, [c_int, c_int, c_int]),
    "lcf_conv_tile_rows": (c_int, [c_int]),
    "lcf_conv_tile_rows_for": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "lcf_conv_tile_plan": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "lcf_indice_conv_fwd_planned": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                            c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                            c_int, c_void_p, c_void_p, c_int, _PI3, c_void_p,
                                            c_void_p]),
    "lcf_indice_conv_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      c_void_p, c_void_p]),
    "lcf_linear_wgrad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "lcf_indice_conv_wgrad_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                           c_int, c_int, c_void_p, c_void_p]),
    "lcf_indice_conv_fwd_b16": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                        c_int, c_void_p, c_int, c_void_p]),
    "lcf_indice_conv_wgrad_b16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                          c_int, c_void_p, c_void_p]),
    "lcf_split_planes": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "lcf_indice_conv_fwd_x6": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                       c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                       c_int, c_void_p]),
    "lcf_dense_nhwc_bf16_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int,
                                         _PI3, c_int, c_void_p, c_void_p]),
    "lcf_dense_nhwc_bf16_bwd_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, _PI3,
                                             c_void_p, c_int, c_void_p]),
    "lcf_load_points": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "lcf_dense": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, _PI3, c_int,
                          c_void_p, c_void_p]),
    "lcf_dense_clear_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                     _PI3, c_void_p, c_void_p]),
    "lcf_dense_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, _PI3, c_void_p,
                              c_void_p]),
    "lcf_vfe_affine_relu_segmax_fwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p]),
    "lcf_vfe_affine_relu_segmax_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_int, c_int, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lcf_dense_nhwc_bf16": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, _PI3,
                                    c_int, c_void_p, c_void_p]),
    "lcf_dense_nhwc_bf16_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, _PI3, c_void_p,
                                        c_void_p]),
    "lcf_vfe_mean": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                             c_void_p]),
    "lcf_vfe_augment": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p,
                                c_void_p]),
    "lcf_vfe_pool": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p,
                             c_void_p, c_int, c_void_p, c_void_p]),
    "lcf_fill_u32": (c_int, [c_void_p, ctypes.c_uint32, c_size_t, c_void_p, c_void_p]),
    "lcf_vfe_fused": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                              c_int] + [c_void_p] * 11),
    "lcf_rotate_iou": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                               c_void_p]),
    "lcf_nms_workspace_bytes": (c_size_t, [c_int]),
    "lcf_nms": (c_int, [c_void_p, c_int, c_int, c_int, ctypes.c_float, ctypes.c_float, c_int,
                        c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lcf_vfe_layer_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "lcf_vfe_layer_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                  ctypes.c_longlong, c_void_p, c_void_p, c_void_p, ctypes.c_float,
                                  c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                  c_void_p]),
    "lcf_vfe_layer_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                  c_int, ctypes.c_longlong, c_void_p, c_void_p, c_void_p, c_int,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                  c_void_p]),
    "lcf_lc_entropy_lookup": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                      c_void_p]),
    "lcf_lc_dp_workspace_bytes": (c_size_t, [c_int, c_int]),
    "lcf_lc_dp_design": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, ctypes.c_double,
                                 c_void_p, c_void_p, c_size_t, c_void_p]),
    "lcf_bn_workspace_bytes": (c_size_t, [c_int, c_int]),
    "lcf_bn_relu_fwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, ctypes.c_float, c_int,
                                c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lcf_bn_relu_fwd_ex": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                   ctypes.c_float, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                   c_void_p, c_void_p, ctypes.c_float, c_void_p, c_size_t,
                                   c_void_p]),
    "lcf_bn_relu_bwd_ex": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int,
                                   c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
    "lcf_bn_relu_bwd_b16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_size_t, c_void_p]),
    "lcf_bn_relu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                c_void_p]),
}

# entry points whose LAST argument is the caller's lcf_ctx (NULL = defaults): the binding lets
# Python callers omit it
CTX_LAST = {"lcf_voxelize", "lcf_voxelize_append", "lcf_voxelize_workspace_reset", "lcf_rulebook_subm",
            "lcf_rulebook_subm_workspace_reset", "lcf_rulebook_conv_workspace_reset",
            "lcf_rulebook_conv_discover", "lcf_rulebook_conv_fill", "lcf_indice_conv_fwd",
            "lcf_indice_conv_fwd_bf16", "lcf_indice_conv_fwd_bf16x3",
            "lcf_indice_conv_fwd_bf16x8", "lcf_indice_conv_fwd_dense", "lcf_fill_u32",
            "lcf_rulebook_chain", "lcf_rulebook_chain_parts", "lcf_rulebook_chain_reset", "lcf_indice_conv_fwd_planned",
            "lcf_dense_clear_rows"}


class _Lib:
    """The loaded CDLL with every declared symbol bound; calls to a CTX_LAST entry point made
    without the trailing context get NULL for it."""

    def __init__(self, cdll):
        self._cdll = cdll
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(cdll, name)  # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
            if name in CTX_LAST:
                fn = self._optional_ctx(fn, len(args))
            setattr(self, name, fn)

    @staticmethod
    def _optional_ctx(fn, nargs):
        def call(*a):
            if len(a) == nargs - 1:
                a = a + (None,)
            return fn(*a)
        return call


# lcf_set_option keys (include/lcf.h)
OPT_CONV_SPLIT, OPT_CONV_IMPL, OPT_CONV_TILE, OPT_CONV_SETS, OPT_CONV_ROWS_HINT = 0, 1, 2, 3, 4
OPT_WS_CLEAN = 5
OPT_NBR_CLEAN = 6
OPT_RB_UNFUSED = 7

_lib = None


def build(force=False):
    """Compile liblcf.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc"), "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(_HERE, "csrc")])


def lib():
    """The loaded library; raises if liblcf.so has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C objectdetection_amd/csrc`.  There is no CPU fallback.")
        _lib = _Lib(ctypes.CDLL(LIB_PATH))
    return _lib


class Context:
    """A caller-owned lcf_ctx (include/lcf.h): per-call options and the open fill batch.  One per
    enqueueing thread / engine; pass ``ctx.handle`` as the last argument of the entry points that
    have options.  Destroyed with the Python object."""

    def __init__(self):
        h = c_void_p()
        check(lib().lcf_ctx_create(ctypes.byref(h)))
        self.handle = h

    def set(self, option, value):
        check(lib().lcf_ctx_set_option(self.handle, int(option), int(value)))

    def fill_batch_begin(self):
        check(lib().lcf_ctx_fill_batch_begin(self.handle))

    def fill_batch_flush(self, stream):
        check(lib().lcf_ctx_fill_batch_flush(self.handle, stream))

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h and _lib is not None:
            _lib.lcf_ctx_destroy(h)


def check(rc):
    if rc != 0:
        raise RuntimeError(lib().lcf_strerror(rc).decode())


def i3(v):
    """int | sequence of 3 -> ctypes int[3]."""
    if isinstance(v, int):
        v = (v, v, v)
    v = [int(x) for x in v]
    if len(v) != 3:
        raise ValueError(f"expected 3 values, got {v}")
    return _I3(*v)


def f3(v):
    return _F3(*[float(x) for x in v])


def f6(v):
    return _F6(*[float(x) for x in v])
