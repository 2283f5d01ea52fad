"""ctypes binding of libq3asr.so (C ABI in include/q3asr.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded
the import of this module raises.
"""
import ctypes
import os
from .config import Q3Config

HOST, DEVICE = 0, 1
F32, BF16, U32 = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libq3asr.so")


class Q3Request(ctypes.Structure):
    _fields_ = [
        ("audio", ctypes.c_void_p), ("offsets", ctypes.POINTER(ctypes.c_int64)), ("n_utt", ctypes.c_int),
        ("sample_rate", ctypes.c_int), ("audio_space", ctypes.c_int),
        ("lang_ids", ctypes.POINTER(ctypes.c_int32)), ("lang_offsets", ctypes.POINTER(ctypes.c_int32)),
        ("max_tokens", ctypes.c_int), ("stop_rules", ctypes.c_int),
        ("forced_tokens", ctypes.POINTER(ctypes.c_int32)),
        ("tap_logits", ctypes.POINTER(ctypes.c_float)), ("tap_steps", ctypes.c_int),
        ("tap_audio_embeds", ctypes.POINTER(ctypes.c_float)), ("tap_inputs_embeds", ctypes.POINTER(ctypes.c_float)),
        ("inject_inputs_embeds", ctypes.POINTER(ctypes.c_float)),
    ]


class Q3Result(ctypes.Structure):
    _fields_ = [
        ("tokens", ctypes.POINTER(ctypes.c_int32)), ("n_tokens", ctypes.POINTER(ctypes.c_int32)),
        ("n_audio_tokens", ctypes.POINTER(ctypes.c_int32)),
        ("audio_seconds", ctypes.c_double),
        ("ms_mel", ctypes.c_double), ("ms_encoder", ctypes.c_double), ("ms_prefill", ctypes.c_double),
        ("ms_decode", ctypes.c_double),
        ("decode_steps", ctypes.c_int64), ("tokens_total", ctypes.c_int64), ("decode_rows", ctypes.c_int64),
    ]


EXPORTS = [
    "q3asr_last_error", "q3asr_create", "q3asr_destroy", "q3asr_load_tensor", "q3asr_finalize", "q3asr_mel_filterbank",
    "q3asr_mel", "q3asr_encode", "q3asr_output_length", "q3asr_transcribe_tokens", "q3asr_op_gemm", "q3asr_op_gemm_q4",
    "q3asr_op_attention", "q3asr_profile_start", "q3asr_profile_stop", "q3asr_op_gemm_time", "q3asr_op_greedy_rules",
    "q3asr_op_read_bandwidth", "q3asr_op_mfma_f32_rate", "q3asr_workspace_bytes", "q3asr_mel_noalloc", "q3asr_decode_attention_cus",
]


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"libq3asr.so not found at {LIB_PATH}: build it with `make -C {os.path.dirname(LIB_PATH)}` "
                          "(the HIP library is the only compute path)")
    lib = ctypes.CDLL(LIB_PATH)
    c_int, c_void_p, P = ctypes.c_int, ctypes.c_void_p, ctypes.POINTER
    lib.q3asr_last_error.restype = ctypes.c_char_p
    lib.q3asr_create.argtypes = [P(Q3Config), c_int, P(c_void_p)]
    lib.q3asr_destroy.argtypes = [c_void_p]
    lib.q3asr_destroy.restype = None
    lib.q3asr_load_tensor.argtypes = [c_void_p, ctypes.c_char_p, c_int, P(ctypes.c_int64), c_int, c_void_p]
    lib.q3asr_finalize.argtypes = [c_void_p]
    lib.q3asr_mel_filterbank.argtypes = [c_void_p, c_void_p]
    lib.q3asr_mel.argtypes = [c_void_p, c_void_p, P(ctypes.c_int64), c_int, c_int, c_int, c_void_p, c_int, P(ctypes.c_int32), c_int]
    lib.q3asr_mel_noalloc.argtypes = [c_void_p, c_void_p, P(ctypes.c_int64), c_int, c_void_p, c_int, P(ctypes.c_int32), c_void_p]
    lib.q3asr_encode.argtypes = [c_void_p, c_void_p, c_int, P(ctypes.c_int32), c_int, c_int, c_void_p, ctypes.c_int64,
                                 P(ctypes.c_int32), c_int]
    lib.q3asr_output_length.argtypes = [c_void_p, c_int]
    lib.q3asr_workspace_bytes.argtypes = [c_void_p]
    lib.q3asr_workspace_bytes.restype = ctypes.c_int64
    lib.q3asr_decode_attention_cus.argtypes = [c_void_p, c_int]
    lib.q3asr_decode_attention_cus.restype = c_int
    lib.q3asr_transcribe_tokens.argtypes = [c_void_p, P(Q3Request), P(Q3Result)]
    lib.q3asr_op_gemm.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.q3asr_op_gemm_q4.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.q3asr_op_attention.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, P(ctypes.c_int32), c_int, c_int, c_int, c_int,
                                       c_int, c_void_p]
    lib.q3asr_op_gemm_time.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P(ctypes.c_float)]
    lib.q3asr_op_greedy_rules.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.q3asr_op_read_bandwidth.argtypes = [c_void_p, ctypes.c_int64, P(ctypes.c_float)]
    lib.q3asr_op_mfma_f32_rate.argtypes = [c_void_p, c_int, c_int, P(ctypes.c_float)]
    lib.q3asr_profile_start.argtypes = [c_void_p, c_int]
    lib.q3asr_profile_stop.argtypes = [c_void_p, ctypes.c_char_p, c_int]
    return lib


class Q3Error(RuntimeError):
    pass


def check(lib, rc):
    if rc != 0:
        raise Q3Error(lib.q3asr_last_error().decode("utf-8", "replace"))
