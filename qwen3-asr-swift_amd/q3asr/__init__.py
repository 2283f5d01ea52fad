"""q3asr: MI355X-native Qwen3-ASR offline transcription (host mirror of the reference API over libq3asr.so)."""
from .config import Q3Config, make_config, oracle_cfgs  # noqa: F401
from .model import Qwen3ASRModel, Qwen3AudioEncoder, WhisperFeatureExtractor, TranscribeOutput  # noqa: F401
from . import synth  # noqa: F401
