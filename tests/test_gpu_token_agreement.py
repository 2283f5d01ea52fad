"""Greedy-token agreement between the HIP path and the CPU restatement, counted over a population of synthetic clips.

The loop under test is Qwen3ASR.swift:567-632 (argmax per step, offset = cache length: QuantizedTextDecoder.swift:211-268).
The restatement decodes each clip freely; the GPU is then teacher-forced with the restatement's tokens, so that one
near-tie does not make every later step of that clip incomparable, and every step is one independent comparison:
GPU argmax (the fused LM-head epilogue's pick and the argmax of the tapped fp32 logits) against the restatement's token.
Reported per population: agree / total, how many steps had a margin above the gate (those are asserted one by one), the
first divergence and its margin.

The seeded test checkpoints are random matrices: their logits are near-Gaussian over 151 936 entries, whose top-2
margin is 3 % of the largest logit at the median and below 0.3 % for one step in twenty -- far flatter than a trained
model's distribution, i.e. the hard case for this criterion: the fp32 restatement and a pipeline that feeds bf16 to its
matrix products differ by 4-10e-3 of the largest logit, so one step in a hundred is decided by rounding.  Measured on
MI355X: dense bf16 weights 99.06 % (3043 / 3072, tiny model) and 99.41 % (509 / 512, 1.7B widths), required >= 99 %;
MLX 4-bit weights 98.14 % (1005 / 1024: the dequantised weights are rounded to bf16 for the matrix cores, a second
rounding per product that the restatement's fp32 `q * scale + bias` does not have), required >= 97.5 %.  In every
population no token differs where the restatement's margin exceeds 1e-2 of its largest logit, and the largest margin
at which one does differ is 3.4e-3.
"""
import numpy as np
import pytest

import oracle
import parity
from q3asr import Qwen3ASRModel, make_config, oracle_cfgs, synth

pytestmark = pytest.mark.gpu


def _population(name, cfg, ck, clips, steps, chunk=8, required=0.99):
    w = synth.checkpoint_to_f32(ck)
    enc_cfg, dec_cfg = oracle_cfgs(cfg)
    dec = oracle.decoder.Decoder(w, dec_cfg)
    m = Qwen3ASRModel.from_checkpoint(cfg, ck)
    agree = agree_tap = total = armed = armed_bad = 0
    worst_err, first, low = 0.0, None, []
    try:
        for c0 in range(0, len(clips), chunk):
            part = clips[c0:c0 + chunk]
            refs, forced = [], np.zeros((len(part), steps), dtype=np.int32)
            for i, audio in enumerate(part):
                trace = []
                toks = oracle.generate.transcribe_tokens(audio, 16000, w, enc_cfg, dec_cfg, max_tokens=steps, dec=dec, trace=trace)
                refs.append((toks, np.stack(trace)))
                forced[i, :len(toks)] = toks
            out = m.transcribe_batch(part, 16000, max_tokens=steps, forced_tokens=forced, tap_steps=steps)
            plain = m.transcribe_batch(part, 16000, max_tokens=steps, forced_tokens=forced)     # no tap: the fused argmax alone
            for i, (toks, ref) in enumerate(refs):
                n = len(toks)
                got = np.asarray(plain.tokens[i][:n])
                assert len(got) == n, "the GPU run stopped at another step than the restatement"
                lg = out.taps["logits"][i, :n]
                ref_ids, mar = parity.margins(ref)
                assert ref_ids.tolist() == toks
                worst_err = max(worst_err, parity.rel(lg, ref))
                same = got == ref_ids
                agree += int(same.sum())
                agree_tap += int((np.argmax(lg, axis=-1) == ref_ids).sum())
                total += n
                arm = mar > parity.TOKEN_GATE
                armed += int(arm.sum())
                armed_bad += int((arm & ~same).sum())
                for s in np.nonzero(~same)[0]:
                    low.append(float(mar[s]))
                    if first is None:
                        first = (c0 + i, int(s), float(mar[s]), float(parity.rel(lg[s], ref[s])))
    finally:
        m.close()
    text = (f"agree {agree}/{total} = {100.0 * agree / total:.2f} % (argmax of the tapped logits: {agree_tap}/{total}); "
            f"armed {armed}/{total}, armed and different {armed_bad}; largest logit distance {worst_err:.2e}")
    if first is not None:
        text += (f"; first divergence: clip {first[0]} step {first[1]}, margin {first[2]:.2e} of max |logit|, logit distance there "
                 f"{first[3]:.2e}; largest margin among the {len(low)} differing steps {max(low):.2e}")
    parity.note(name, text)
    print(name, text)
    assert worst_err <= parity.LOGIT_RTOL
    assert armed_bad == 0
    assert agree >= required * total, text
    return agree, total


def test_tiny_model_64_clips_of_48_steps():
    cfg = make_config("tiny")
    ck = synth.make_checkpoint(cfg, seed=1)
    clips = [synth.synth_audio(12000 + 1900 * i, 5000 + i, "noise" if i % 8 == 5 else "speechlike") for i in range(64)]   # 0.75 - 8.2 s
    _population("tiny, 64 clips x 48 steps", cfg, ck, clips, 48)


def test_tiny_q4_model_32_clips_of_32_steps():
    cfg = make_config("tiny", weight_format=1)
    ck = synth.make_checkpoint(cfg, seed=2)
    clips = [synth.synth_audio(14000 + 2100 * i, 5200 + i) for i in range(32)]
    _population("tiny MLX-q4, 32 clips x 32 steps", cfg, ck, clips, 32, required=0.975)


def test_true_width_17b_two_layers_16_clips_of_32_steps():
    """The 1.7B model's layer widths with two layers per stack (the restatement's LM head alone is 0.3 GFLOP per step)."""
    cfg = make_config("1.7b", enc_layers=2, layers=2)
    ck = synth.make_checkpoint(cfg, seed=5)
    clips = [synth.synth_audio(16000 + 2500 * i, 5400 + i) for i in range(16)]
    _population("1.7b widths x 2 layers, 16 clips x 32 steps", cfg, ck, clips, 32, chunk=4)


def test_448_steps_of_a_30s_clip_four_wave_attention(monkeypatch):
    """One 30 s clip to the loop's limit (405 prompt rows + 447 decoded positions: the RoPE positions and the key-tile
    counts, 26-54 tiles of 16 keys, that the throughput benchmark spends its time at), on the four-wave decode attention
    of the large groups, teacher-forced with the restatement's tokens: logits within the bound at every step -- steps
    100 / 200 / 300 / 447 are reported -- and the token agreement counted."""
    monkeypatch.setenv("Q3ASR_ATTN_WAVES", "4")
    cfg = make_config("tiny")
    ck = synth.make_checkpoint(cfg, seed=1)
    w = synth.checkpoint_to_f32(ck)
    enc_cfg, dec_cfg = oracle_cfgs(cfg)
    audio = synth.synth_audio(480000, 77)
    steps = 448
    # the restatement runs without the repetition guards (a random model may loop): the forced stream is its argmax chain
    dec = oracle.decoder.Decoder(w, dec_cfg)
    a = oracle.encoder.encode(oracle.mel.process(audio, 16000), w, enc_cfg)
    ids, a0, a1 = oracle.generate.prompt_ids(a.shape[0])
    assert len(ids) == 405
    h, cache = dec.forward(oracle.generate.splice(dec, ids, a0, a1, a, True), None)
    ref = [dec.logits(h[-1:])[0]]
    toks = [int(np.argmax(ref[0]))]
    for _ in range(steps - 1):
        h, cache = dec.forward(dec.embed([toks[-1]]), cache)
        ref.append(dec.logits(h)[0])
        toks.append(int(np.argmax(ref[-1])))
    ref = np.stack(ref)
    forced = np.asarray([toks], dtype=np.int32)
    m = Qwen3ASRModel.from_checkpoint(cfg, ck)
    try:
        # two copies of the clip: a second row keeps the group off the single-utterance GEMV kernels
        out = m.transcribe_batch([audio, audio], 16000, max_tokens=steps, forced_tokens=np.repeat(forced, 2, axis=0),
                                 stop_rules=False, tap_steps=steps)
    finally:
        m.close()
    lg = out.taps["logits"][0]
    per_step = np.abs(lg - ref).max(axis=1) / np.abs(ref).max(axis=1)
    parity.note("448 steps, 30 s clip, four-wave attention",
                "logit distance at steps 0/100/200/300/447: " + " ".join(f"{per_step[s]:.2e}" for s in (0, 100, 200, 300, 447))
                + f"; largest over all steps {per_step.max():.2e} at step {int(per_step.argmax())}")
    assert np.isfinite(lg).all()
    assert per_step.max() <= parity.LOGIT_RTOL, (int(per_step.argmax()), float(per_step.max()))
    assert np.array_equal(out.taps["logits"][1], lg)          # the same clip in the other row
    # one clip is not a population (a random model that decodes freely revisits the same near-ties): measured 436 / 448,
    # every difference at a margin below 1e-2; the count guards against a systematic error at long contexts
    agree, total, armed = parity.check_tokens("448 steps, 30 s clip, four-wave attention", out.tokens[0], ref)
    assert agree >= 0.95 * total
