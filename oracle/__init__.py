"""CPU oracle for the Qwen3-ASR offline-transcription hot path.

TEST INFRASTRUCTURE ONLY.  This package restates, in plain numpy (fp32), the
algorithm of the reference Swift sources so that the HIP library can be checked
against it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; the product path (``q3asr`` + ``libq3asr.so``)
never does and fails loudly when the HIP library is missing.

PARITY UNPINNED by the reference itself: the reference (Swift + mlx-swift +
Accelerate) can be neither compiled nor imported in this environment and its own
tests hold no numerical golden vector for this path.  The restatement is instead
pinned against independent implementations of the shared architecture that ship
with the image (``transformers.WhisperFeatureExtractor`` for the log-mel frontend,
``transformers.models.qwen3_asr`` / ``qwen3`` for encoder and decoder) through the
fixtures under ``tests/golden/`` (generator: ``tests/golden/make_golden.py``).

Every function cites the reference file:line it follows (paths relative to the
reference checkout, e.g. ``Sources/Qwen3ASR/AudioPreprocessing.swift``).
"""
from . import mel, encoder, decoder, generate, q4  # noqa: F401
