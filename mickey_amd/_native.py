"""ctypes binding of libmickey_hip.so (the C ABI declared in include/mickey_hip.h).

The product path calls the HIP kernels ONLY through this module; there is no CPU or ATen fallback:
if the shared library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

import torch

from . import build as _build

MK_BF16, MK_F16, MK_F32 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2

_P, _I, _L, _F, _U = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_ulonglong
_CODES = {"p": _P, "i": _I, "l": _L, "f": _F, "u": _U}

# name -> (restype code, argument codes); order and meaning exactly as in include/mickey_hip.h
SIGNATURES = {
    "mk_version": ("i", ""),
    "mk_last_error": ("s", ""),
    "mk_preprocess_u8": ("i", "pliiipiip"),
    "mk_gemm": ("i", "pipippiiiiiiip"),
    "mk_gemm_grouped": ("i", "pilpilplpiliiiiiiip"),
    "mk_gemm_ls_residual": ("i", "pipipppiiiiip"),
    "mk_gemm_qkv": ("i", "pipippppiiiifip"),
    "mk_gemm_patch_embed": ("i", "pipipppiiiiip"),
    "mk_gemm_ls_residual_ln": ("i", "pipippppipppiiiiip"),
    "mk_gemm_patch_embed_ln": ("i", "pipipppppiiiiip"),
    "mk_cls_token_ln": ("i", "pppppiiiip"),
    "mk_recentre_split": ("i", "ppplilp"),
    "mk_gemm_ln": ("i", "pipipppfppiiiiiip"),
    "mk_gemm_qkv_ln": ("i", "pipipppfppppiiiifip"),
    "mk_im2col_patch14": ("i", "plliiiipiip"),
    "mk_cls_token": ("i", "pppiiip"),
    "mk_layernorm": ("i", "pippfpiipiiiiiiiiiip"),
    "mk_gemm_ln128": ("i", "pilpilppfpipiiiiiiip"),
    "mk_layernorm_planes": ("i", "pippfppifpiiiiiiiiipp"),
    "mk_flash_attn_fwd": ("i", "ppppiiiiiip"),
    "mk_bordered_rows": ("l", "iii"),
    "mk_conv3x3": ("i", "pliplipilplplpiliiiiiiip"),
    "mk_conv3x3_split": ("i", "pplipplipilplppiliiiiiiffpp"),
    "mk_split_planes": ("i", "plilfpplpp"),
    "mk_gemm_grouped_split": ("i", "ppilpilplppiliiiiiffpp"),
    "mk_posenc_add": ("i", "ppppiiiiiip"),
    "mk_linattn_work_floats": ("l", "iiii"),
    "mk_linattn_kv": ("i", "pppiiiip"),
    "mk_linattn_apply": ("i", "pppiiiiiip"),
    "mk_head_tails": ("i", "pppppppppppiiiiiiiififp"),
    "mk_dual_softmax_work_floats": ("l", "iiii"),
    "mk_dual_softmax": ("i", "ppppfifppppiiiip"),
    "mk_dual_softmax_split_work_floats": ("l", "iii"),
    "mk_dual_softmax_split": ("i", "ppppfifppppiiiip"),
    "mk_sinkhorn_work_floats": ("l", "iii"),
    "mk_sinkhorn": ("i", "ppppfippppiiiip"),
    "mk_mutual_nn": ("i", "ppppiiip"),
    "mk_exprace_topk_work_bytes": ("l", "iiil"),
    "mk_exprace_topk_state_bytes": ("l", "ii"),
    "mk_exprace_topk": ("i", "ppuupppppiiliip"),
    "mk_counter_add": ("i", "pup"),
    "mk_gather_backproject": ("i", "ppppppppppppiiiiip"),
    "mk_gather_backproject_bwd": ("i", "ppppppppppiiiiip"),
    "mk_ransac_hypotheses": ("i", "pppppuupfppppiiilp"),
    "mk_refine_pose": ("i", "pppppfiipppppppiiiip"),
    "mk_pose_finalize": ("i", "ppppip"),
    "mk_train_ransac_masks": ("i", "pppppuupfiipppiiilp"),
    "mk_reinforce_scatter": ("i", "ppppiiilp"),
    "mk_train_tail_fwd": ("i", "pppppppiiiifiifpppp"),
    "mk_train_tail_bwd": ("i", "pppppppiiiifiifppppppp"),
    "mk_train_aggregate_fwd": ("i", "pppiiififfppppp"),
    "mk_train_aggregate_bwd": ("i", "ppiiipp"),
}

# development knobs (include/mickey_hip_dev.h): process-wide schedule selectors for benchmarks / tests, never called by
# the product path
DEV_SIGNATURES = {
    "mk_gemm_set_tile": ("i", "i"),
    "mk_attn_set_mode": ("i", "i"),
    "mk_sinkhorn_set_group": ("i", "i"),
    "mk_dual_softmax_set_chunks": ("i", "i"),
    "mk_exprace_set_mode": ("i", "i"),
    "mk_dev_mfma_sustained": ("i", "piiip"),
}

_lib = None
_missing = set()


class MickeyHipError(RuntimeError):
    pass


def lib_path():
    return os.environ.get("MICKEY_HIP_LIB", _build.lib_path())


def load():
    """Load libmickey_hip.so; raises MickeyHipError if it is absent (never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise MickeyHipError("libmickey_hip.so not found at %s -- build it with `python -m mickey_amd.build` "
                             "(mickey_amd has no CPU/ATen fallback)" % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in list(SIGNATURES.items()) + list(DEV_SIGNATURES.items()):
        try:
            fn = getattr(lib, name)
        except AttributeError:
            _missing.add(name)  # reported by missing_symbols(); calling it raises
            continue
        fn.restype = ctypes.c_char_p if res == "s" else _CODES[res]
        fn.argtypes = [_CODES[c] for c in args]
    _lib = lib
    return lib


def exported_symbols():
    return list(SIGNATURES)


def missing_symbols():
    load()
    return sorted(_missing)


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise MickeyHipError("expected a device tensor")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt):
    if dt == torch.bfloat16:
        return MK_BF16
    if dt == torch.float16:
        return MK_F16
    if dt == torch.float32:
        return MK_F32   # the exact parity mode: "lp" buffers hold fp32, contractions on the fp32-input MFMA
    raise MickeyHipError("unsupported low-precision dtype %s" % dt)


def call(name, *args):
    """Call an ABI function that returns a status code; raise on failure."""
    lib = load()
    if name in _missing:
        raise MickeyHipError("libmickey_hip.so does not export %s (stale build?)" % name)
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise MickeyHipError("%s failed (%d): %s" % (name, rc, lib.mk_last_error().decode()))


def query(name, *args):
    return getattr(load(), name)(*args)
