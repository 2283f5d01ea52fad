"""Utterances that stop leave the decode steps (the reference computes nothing for a clip after its EOS,
Qwen3ASR.swift:571-574, 597-627): stop points are scripted through teacher-forced inputs, and the run that drops
finished rows must give, token for token, what the run that keeps them gives and what the CPU restatement gives."""
import numpy as np
import pytest

import oracle
from parity import check_tokens
from q3asr import Qwen3ASRModel, make_config, oracle_cfgs, synth

pytestmark = pytest.mark.gpu

EOS, ENDOFTEXT, IM_START = 151645, 151643, 151644


def _script(n_utt, max_tokens, stops, seed):
    """Forced decoder inputs [n_utt][max_tokens]: random text ids with a stop token at stops[u] (None: never)."""
    rng = np.random.default_rng(seed)
    forced = rng.integers(1000, 100000, size=(n_utt, max_tokens)).astype(np.int32)
    for u, s in enumerate(stops):
        if s is not None:
            forced[u, s] = (EOS, ENDOFTEXT, IM_START)[u % 3] if s > 0 else EOS    # a control token only stops inside the loop
    return forced


def _expected_counts(forced, max_tokens):
    """Tokens each utterance keeps when the loop's control flow follows the forced stream (oracle rules)."""
    out = []
    for row in forced:
        toks, seen = [int(row[0])], False
        for i in range(1, max_tokens):
            if toks[-1] in (EOS, ENDOFTEXT):
                break
            if toks[-1] == oracle.generate.ASR_TEXT:
                seen = True
            toks.append(int(row[i]))
            if oracle.generate.stop_after_append(toks, seen):
                break
        out.append(len(toks))
    return out


def _oracle_forced(cfg, ck, audio, forced_row, n):
    """Teacher-forced logits of the CPU restatement for the first n steps of one utterance."""
    w = synth.checkpoint_to_f32(ck)
    enc_cfg, dec_cfg = oracle_cfgs(cfg)
    a = oracle.encoder.encode(oracle.mel.process(audio, 16000), w, enc_cfg)
    dec = oracle.decoder.Decoder(w, dec_cfg)
    ids, a0, a1 = oracle.generate.prompt_ids(a.shape[0])
    h, cache = dec.forward(oracle.generate.splice(dec, ids, a0, a1, a, bool(cfg.embed_scale)), None)
    logits = [dec.logits(h[-1:])[0]]
    for i in range(n - 1):
        h, cache = dec.forward(dec.embed([int(forced_row[i])]), cache)
        logits.append(dec.logits(h)[0])
    return np.stack(logits)


def test_retired_rows_change_nothing(pinned_model, monkeypatch):
    """Ten utterances with scripted stops between the prefill's token and never, decoded with one kernel variant at
    every row count: dropping finished rows at the polls gives exactly the tokens of the run that keeps every row
    until the last utterance stops; the counts follow the reference's stop rules on the scripted stream."""
    m, cfg, ck = pinned_model
    max_tokens = 44
    stops = [0, 3, 9, 10, None, 17, 25, 33, 2, 40]
    audios = [synth.synth_audio(16000 + 3000 * i, 600 + i) for i in range(len(stops))]
    forced = _script(len(stops), max_tokens, stops, seed=5)
    want_counts = _expected_counts(forced, max_tokens)
    assert want_counts == [1, 4, 10, 11, 44, 18, 26, 34, 3, 41]
    monkeypatch.setenv("Q3ASR_RETIRE", "0")
    keep = m.transcribe_batch(audios, 16000, max_tokens=max_tokens, forced_tokens=forced, stop_rules="forced")
    monkeypatch.setenv("Q3ASR_RETIRE", "1")
    drop = m.transcribe_batch(audios, 16000, max_tokens=max_tokens, forced_tokens=forced, stop_rules="forced")
    assert [len(t) for t in keep.tokens] == want_counts
    assert drop.tokens == keep.tokens
    assert keep.decode_steps == drop.decode_steps == max_tokens - 1
    assert keep.decode_rows == len(stops) * (max_tokens - 1)
    # every utterance leaves within one poll interval (8 steps) of its stop
    upper = sum(min(max_tokens - 1, n - 1 + 8) for n in want_counts)
    assert sum(n - 1 for n in want_counts) <= drop.decode_rows <= upper < keep.decode_rows
    # against the CPU restatement: the tokens an utterance keeps are the argmax of its teacher-forced logits
    for u in (1, 5, 9):
        ref = _oracle_forced(cfg, ck, audios[u], forced[u], want_counts[u])
        check_tokens(f"retired run, utterance {u}", drop.tokens[u], ref[:want_counts[u]])


def test_retirement_with_the_default_plan_and_gemv_tail(tiny_model, monkeypatch):
    """The library's own choice of kernels per row count (MFMA tiles, then the few-row GEMV kernels once at most four
    utterances still run): same stop points, and the same tokens wherever the CPU restatement's top-2 margin exceeds
    the bf16 tolerance."""
    m, cfg, ck = tiny_model
    max_tokens = 30
    stops = [4, 4, 12, None, 20, 7, 1]
    audios = [synth.synth_audio(20000 + 5000 * i, 700 + i) for i in range(len(stops))]
    forced = _script(len(stops), max_tokens, stops, seed=9)
    want_counts = _expected_counts(forced, max_tokens)
    out = m.transcribe_batch(audios, 16000, max_tokens=max_tokens, forced_tokens=forced, stop_rules="forced")
    assert [len(t) for t in out.tokens] == want_counts
    assert out.decode_rows < len(stops) * (max_tokens - 1)
    for u in (3, 4):
        ref = _oracle_forced(cfg, ck, audios[u], forced[u], want_counts[u])
        check_tokens(f"default plan with GEMV tail, utterance {u}", out.tokens[u], ref[:want_counts[u]])


def test_free_running_stops_retire_too(pinned_model, monkeypatch):
    """Without teacher forcing the stop rules act on the model's own tokens; whatever they decide, dropping the
    finished rows must not change any utterance's transcript (here the repetition guards end most of the clips)."""
    m, cfg, _ = pinned_model
    audios = [synth.synth_audio(16000 + 7000 * i, 800 + i) for i in range(9)]
    prefix = [[4128, 6364]] * 9          # forced-language prompts arm the repetition rules from the first token
    monkeypatch.setenv("Q3ASR_RETIRE", "0")
    keep = m.transcribe_batch(audios, 16000, max_tokens=96, lang_ids=prefix)
    monkeypatch.setenv("Q3ASR_RETIRE", "1")
    drop = m.transcribe_batch(audios, 16000, max_tokens=96, lang_ids=prefix)
    assert drop.tokens == keep.tokens
    assert drop.decode_rows <= keep.decode_rows


@pytest.mark.parametrize("merge_rows,lm_split", [(0, "1"), (4, "2"), (100, "1"), (0, "2")])
def test_split_decode_retires_and_merges(pinned_model, monkeypatch, merge_rows, lm_split):
    """The two-half decode (attention on its own share of the compute units) drops finished rows per half and goes
    back to one list on one stream when few utterances are left (merge_rows) or a half has emptied (here: half 0
    holds only early stops): the tokens are those of the single-stream run that keeps every row."""
    m, cfg, ck = pinned_model
    max_tokens = 44
    stops = [0, 3, 9, 10, 2, 6, None, 17, 25, 33, 40]          # half 0 = utterances 0-5, half 1 = 6-10
    audios = [synth.synth_audio(16000 + 3000 * i, 900 + i) for i in range(len(stops))]
    forced = _script(len(stops), max_tokens, stops, seed=6)
    want_counts = _expected_counts(forced, max_tokens)
    monkeypatch.setenv("Q3ASR_DECODE_ATT_CUS", "0")
    monkeypatch.setenv("Q3ASR_RETIRE", "0")
    keep = m.transcribe_batch(audios, 16000, max_tokens=max_tokens, forced_tokens=forced, stop_rules="forced")
    assert [len(t) for t in keep.tokens] == want_counts
    monkeypatch.setenv("Q3ASR_RETIRE", "1")
    monkeypatch.setenv("Q3ASR_DECODE_ATT_CUS", "160")
    monkeypatch.setenv("Q3ASR_DECODE_MERGE_ROWS", str(merge_rows))
    monkeypatch.setenv("Q3ASR_LM_SPLIT", lm_split)   # 2: the LM head shared between the two streams even for this small group
    split = m.transcribe_batch(audios, 16000, max_tokens=max_tokens, forced_tokens=forced, stop_rules="forced")
    assert split.tokens == keep.tokens
    assert split.decode_steps == max_tokens - 1
    upper = sum(min(max_tokens - 1, n - 1 + 8) for n in want_counts)
    assert sum(n - 1 for n in want_counts) <= split.decode_rows <= upper < keep.decode_rows
    # and with every row kept, the two-half loop alone
    monkeypatch.setenv("Q3ASR_RETIRE", "0")
    split_keep = m.transcribe_batch(audios, 16000, max_tokens=max_tokens, forced_tokens=forced, stop_rules="forced")
    assert split_keep.tokens == keep.tokens
    assert split_keep.decode_rows == keep.decode_rows
