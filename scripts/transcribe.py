#!/usr/bin/env python3
"""Transcribes WAV files with a local Qwen3-ASR checkpoint directory, like the reference CLI's `transcribe`
command (Sources/Qwen3ASRCLI/main.swift:297-315: load the model, read the file, call transcribe).

  python scripts/transcribe.py <model_dir> <file.wav> [more.wav ...] [--language English] [--long segment]

<model_dir> holds config.json, *.safetensors and (for text output) vocab.json / merges.txt.  All files are
transcribed as one batch on GPU 0; without a tokenizer the generated token ids are printed.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))

from q3asr import Qwen3ASRModel  # noqa: E402
from q3asr.audio_io import load_wav  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model_dir")
    ap.add_argument("wav", nargs="+")
    ap.add_argument("--language", default=None, help="force the output language (needs the tokenizer files)")
    ap.add_argument("--max-tokens", type=int, default=448)
    ap.add_argument("--long", default="truncate", choices=["truncate", "segment", "segment_fixed"],
                    help="recordings over 30 s: keep the first 30 s like the reference, or cut into pieces")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()

    model = Qwen3ASRModel.from_pretrained(args.model_dir, device=args.device, max_batch=max(1, min(len(args.wav) * 4, 2048)))
    clips, rate = [], None
    for path in args.wav:
        audio, sr = load_wav(path)
        if rate not in (None, sr):
            raise SystemExit(f"{path}: sample rate {sr} differs from {rate}; transcribe files of one rate per call")
        rate = sr
        clips.append(audio)
    t0 = time.perf_counter()
    if args.language is None:
        texts, _ = model.transcribe_long(clips, rate, mode=args.long, max_tokens=args.max_tokens)
    else:
        texts = [model.transcribe(c, rate, language=args.language, max_tokens=args.max_tokens) for c in clips]
    dt = time.perf_counter() - t0
    for path, text in zip(args.wav, texts):
        shown = text if isinstance(text, str) else " ".join(str(t) for t in text)
        print(f"{path}: {shown}")
    seconds = sum(len(c) for c in clips) / float(rate)
    print(f"# {seconds:.1f} s of audio in {dt:.2f} s ({seconds / dt:.0f}x real time)", file=sys.stderr)
    model.close()


if __name__ == "__main__":
    main()
