"""ctypes binding of libmmiss.so (include/mmiss.h, include/mmiss_debug.h).

The library is the product; this module only loads it and converts its status codes into
exceptions. There is deliberately NO fallback: if libmmiss.so is missing or does not load, importing
the compute path raises (build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C multimodal-image-similarity-search_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmmiss.so")

MMISS_OK = 0
MMISS_F32 = 0
MMISS_F16 = 1
MMISS_F8 = 2
MMISS_PREC_BF16 = 0
MMISS_PREC_FP8 = 1
MMISS_PREC_BF16_F32RESID = 2
MMISS_TOWER_VISION = 0
MMISS_TOWER_TEXT = 1

EPI_F32, EPI_BIAS_BF16, EPI_BIAS_QGELU_BF16, EPI_BIAS_RESID_F32, EPI_PATCH_F32 = range(5)


class MmissError(RuntimeError):
    """Non-zero status from libmmiss (the reference's `except Exception` paths catch it)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"libmmiss error {code}: {msg}")
        self.code = code


class ClipConfigStruct(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("v_hidden", C.c_int32), ("v_layers", C.c_int32), ("v_heads", C.c_int32),
        ("v_mlp", C.c_int32), ("v_patch", C.c_int32), ("v_image", C.c_int32),
        ("t_hidden", C.c_int32), ("t_layers", C.c_int32), ("t_heads", C.c_int32),
        ("t_mlp", C.c_int32), ("t_vocab", C.c_int32), ("t_ctx", C.c_int32),
        ("proj_dim", C.c_int32), ("eos_token_id", C.c_int32), ("ln_eps", C.c_float),
        ("max_batch_image", C.c_int32), ("max_batch_text", C.c_int32),
    ]


_P = C.c_void_p
_I = C.c_int
_I32 = C.c_int32
_I64 = C.c_int64

# name -> (restype, argtypes); every symbol declared in include/mmiss.h and include/mmiss_debug.h
SIGNATURES = {
    # mmiss.h
    "mmiss_abi_version": (_I, []),
    "mmiss_last_error": (C.c_char_p, []),
    "mmiss_device_count": (_I, [C.POINTER(_I)]),
    "mmiss_encoder_create": (_I, [C.POINTER(ClipConfigStruct), _I, C.POINTER(_P)]),
    "mmiss_encoder_destroy": (_I, [_P]),
    "mmiss_encoder_set_weight": (_I, [_P, C.c_char_p, _P, _I64, C.POINTER(_I)]),
    "mmiss_encoder_finalize": (_I, [_P]),
    "mmiss_encoder_set_precision": (_I, [_P, _I32]),
    "mmiss_encoder_set_tower_precision": (_I, [_P, _I32, _I32]),
    "mmiss_encoder_set_stream": (_I, [_P, _P, _I32]),
    "mmiss_encode_image": (_I, [_P, _P, _I32, _P]),
    "mmiss_encode_image_u8": (_I, [_P, _P, _I32, _P]),
    "mmiss_resize_crop_rgb": (_I, [_P, _P, _I64, _P, _P, _P, _I32, _P]),
    "mmiss_encode_image_rgb": (_I, [_P, _P, _I64, _P, _P, _P, _I32, _P]),
    "mmiss_encode_text": (_I, [_P, _P, _I32, _I32, _P]),
    "mmiss_encoder_tap": (_I, [_P, _I, _I, _P, _I64, C.POINTER(_I64)]),
    "mmiss_index_create": (_I, [_I32, _I32, _I, _I64, C.POINTER(_P)]),
    "mmiss_index_destroy": (_I, [_P]),
    "mmiss_index_set_stream": (_I, [_P, _P, _I32]),
    "mmiss_index_add": (_I, [_P, _P, _P, _I64]),
    "mmiss_index_update": (_I, [_P, _P, _P, _I64]),
    "mmiss_index_remove": (_I, [_P, _P, _I64, C.POINTER(_I64)]),
    "mmiss_index_clear": (_I, [_P]),
    "mmiss_index_count": (_I, [_P, C.POINTER(_I64)]),
    "mmiss_index_get": (_I, [_P, _P, _I64, _P]),
    "mmiss_index_labels": (_I, [_P, _P, _I64]),
    "mmiss_index_query": (_I, [_P, _P, _I32, _I32, _P, _P, _P]),
    "mmiss_index_query_begin": (_I, [_P, _P, _I32, _I32, _P, _P, _P]),
    "mmiss_index_query_end": (_I, [_P]),
    "mmiss_index_query_abort": (_I, [_P]),
    "mmiss_index_guard_stats": (_I, [_P, C.POINTER(_I64)]),
    "mmiss_index_guard_stats_ex": (_I, [_P, C.POINTER(_I64)]),
    "mmiss_index_save": (_I, [_P, C.c_char_p]),
    "mmiss_index_load": (_I, [_P, C.c_char_p]),
    "mmiss_blend": (_I, [_I, _P, _P, _P, C.c_double, _I32, _I32, _P]),
    "mmiss_merge_topk": (_I, [_I, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "mmiss_prof_enable": (_I, [_I]),
    "mmiss_prof_filter": (_I, [C.c_char_p, _I]),
    "mmiss_prof_reset": (_I, []),
    "mmiss_prof_read": (_I, [C.c_char_p, C.c_size_t]),
    # mmiss_debug.h
    "mmiss_dbg_gemm": (_I, [_I, _P, _I, _I, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32]),
    "mmiss_dbg_gemm_p256": (_I, [_I, _P, _I, _P, _P, _P, _P, _P, _P, C.c_float, _I32, _I32, _I32, _I32, _I32, _P]),
    "mmiss_dbg_gemm_resid16": (_I, [_I, _P, _I, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "mmiss_dbg_gemm_time": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32,
                                 C.POINTER(C.c_float)]),
    "mmiss_dbg_layernorm": (_I, [_I, _P, _P, _P, _P, _P, _I32, _I32, _I32, C.c_float]),
    "mmiss_dbg_attention": (_I, [_I, _P, _P, _P, _I32, _I32, _I32, _I32]),
    "mmiss_dbg_im2col": (_I, [_I, _P, _P, _P, _I32, _I32, _I32, _I32]),
    "mmiss_dbg_encoder_record_taps": (_I, [_P, _I]),
    "mmiss_dbg_encoder_set_fuse_ln": (_I, [_P, _I]),
    "mmiss_dbg_set_option": (_I, [C.c_char_p, _I]),
    "mmiss_dbg_build_flags": (_I, []),
    "mmiss_dbg_quantize_weights_fp8": (_I, [_I, _P, _P, _P, _P, _I32, _I32]),
    "mmiss_dbg_layernorm_mxfp8": (_I, [_I, _P, _P, _P, _P, _P, _P, _I32, _I32, C.c_float]),
    "mmiss_dbg_layernorm16_mxfp8": (_I, [_I, _P, _P, _P, _P, _P, _P, _I32, _I32, C.c_float]),
    "mmiss_dbg_attention_mx": (_I, [_I, _P, _P, _P, _P, _I32, _I32, _I32]),
    "mmiss_dbg_gemm8": (_I, [_I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32]),
    "mmiss_dbg_gemm8_xt": (_I, [_I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, C.c_float, _P, _P, _P]),
    "mmiss_dbg_quant16_mxfp8_stats": (_I, [_I, _P, _P, _P, _P, _P, _I32, _I32]),
    "mmiss_dbg_quantize_weights_fp8_csum": (_I, [_I, _P, _P, _P, _P, _P, _I32, _I32]),
    "mmiss_dbg_gemm8_time": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, C.POINTER(C.c_float)]),
    "mmiss_dbg_gemm_split_time": (_I, [_I, _I, _I, _P, _P, _P, _P, _I32, _I32, _I32, _I32, C.POINTER(C.c_float)]),
}

_lib = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """Load libmmiss.so once and attach prototypes. Raises if the library is absent."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is not built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` at the repo root. "
                "There is no CPU fallback for the mmiss hot path."
            )
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if lib.mmiss_abi_version() != 1:
            raise ImportError("libmmiss.so ABI version mismatch")
        if (lib.mmiss_dbg_build_flags() & 2) and os.environ.get("MMISS_ALLOW_AB_BUILD") != "1":
            raise ImportError(
                f"{LIB_PATH} was built with a timing-experiment macro (tools/*_ab.sh: P256_NO_LATE_WAIT, P256_SPLIT_STAGE, P256_STAGE_FIRST, "
                "MMISS_SCAN_NT) and is not a product build; rebuild it (`make -C .../csrc clean all`) or set "
                "MMISS_ALLOW_AB_BUILD=1 for the A/B run itself")
        _lib = lib
        # experiment knobs from the environment: MMISS_OPTIONS="gemm_wide=1,scan_rounds=2" (see mmiss_dbg_set_option)
        for kv in filter(None, os.environ.get("MMISS_OPTIONS", "").split(",")):
            k, _, v = kv.partition("=")
            lib.mmiss_dbg_set_option(k.strip().encode(), int(v))
        return lib


def check(status: int) -> None:
    if status != MMISS_OK:
        msg = load().mmiss_last_error()
        raise MmissError(status, msg.decode("utf-8", "replace") if msg else "")


def ptr(x) -> int:
    """Raw address of a torch tensor / numpy array (contiguous) or an int passthrough."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        if not x.is_contiguous():
            raise ValueError("tensor passed to libmmiss must be contiguous")
        return x.data_ptr()
    if hasattr(x, "ctypes"):
        if not x.flags["C_CONTIGUOUS"]:
            raise ValueError("array passed to libmmiss must be C-contiguous")
        return x.ctypes.data
    raise TypeError(f"cannot take the address of {type(x)!r}")


def device_count() -> int:
    n = _I(0)
    check(load().mmiss_device_count(C.byref(n)))
    return n.value


def current_stream_ptr(device=None):
    """hipStream_t of torch's current stream as an integer (0 = default stream)."""
    import torch

    return int(torch.cuda.current_stream(device).cuda_stream)


def has_experiments() -> bool:
    """True when libmmiss.so was built with `make EXPERIMENTS=1` (measured-slower GEMM / LayerNorm alternatives)."""
    return bool(load().mmiss_dbg_build_flags() & 1)


def set_option(key: str, value: int) -> None:
    check(load().mmiss_dbg_set_option(key.encode(), int(value)))


def prof_enable(on: bool) -> None:
    check(load().mmiss_prof_enable(1 if on else 0))


def prof_filter(kernel=None, stride: int = 1) -> None:
    check(load().mmiss_prof_filter(kernel.encode() if kernel else None, int(stride)))


def prof_reset() -> None:
    check(load().mmiss_prof_reset())


def prof_read() -> list:
    import json

    buf = C.create_string_buffer(1 << 16)
    check(load().mmiss_prof_read(buf, len(buf)))
    return json.loads(buf.value.decode())
