#!/usr/bin/env python3
"""Mints the golden vectors under tests/golden/.

The reference (Swift + MLX) cannot run here and its own tests hold no numerical vector for this
path, so the CPU restatement in oracle/ is pinned against INDEPENDENT implementations of the shared
architecture that ship with the image, run in this container:

  G1  log-mel of the reference's test clip (frames 450-600, 700-800 and 1900-2000) from
      transformers.WhisperFeatureExtractor (float64 numpy STFT);
  G2  the slaney filterbank from transformers.audio_utils.mel_filter_bank (float64);
  G3  audio-encoder + projector output of transformers.models.qwen3_asr (fp32 torch) for seeded tiny
      weights on 250 / 300 / 1000-frame inputs;
  G4  Qwen3 decoder hidden states / logits / greedy ids from transformers Qwen3Model (fp32 torch,
      rotate-half RoPE, no embedding rescale) for seeded tiny weights.

Where those implementations differ from the Swift by design (embedding rescale, interleaved RoPE,
single-chunk padding) the Swift wins and the restatement is the only statement; those paths are
covered by hand-computed cases in tests/test_oracle.py.  Run from the repository root:

  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

os.environ["HF_HUB_OFFLINE"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from conftest import read_wav_pcm16  # noqa: E402
import oracle  # noqa: E402
from q3asr import make_config, synth  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)


def g1_g2():
    from transformers import WhisperFeatureExtractor
    from transformers.audio_utils import mel_filter_bank
    audio, rate = read_wav_pcm16(os.path.join(HERE, "test_audio.wav"))
    a16 = oracle.mel.resample_linear(audio, rate, 16000)
    fe = WhisperFeatureExtractor(feature_size=128, sampling_rate=16000, hop_length=160, chunk_length=30, n_fft=400)
    feats = fe(a16, sampling_rate=16000, padding=False, return_tensors="np")["input_features"][0]
    assert feats.shape == (128, 2000), feats.shape
    # the clip is digital silence except 5.0 - 8.5 s: keep the onset, the loudest second and a silent stretch
    np.save(os.path.join(HERE, "g1_mel_hf_f450_600.npy"), feats[:, 450:600].astype(np.float32))
    np.save(os.path.join(HERE, "g1_mel_hf_f700_800.npy"), feats[:, 700:800].astype(np.float32))
    np.save(os.path.join(HERE, "g1_mel_hf_f1900_2000.npy"), feats[:, 1900:2000].astype(np.float32))
    fb = mel_filter_bank(num_frequency_bins=201, num_mel_filters=128, min_frequency=0.0, max_frequency=8000.0,
                         sampling_rate=16000, norm="slaney", mel_scale="slaney")
    np.save(os.path.join(HERE, "g2_filterbank_hf.npy"), fb.T.astype(np.float32))


def hf_encoder(cfg, w):
    from transformers.models.qwen3_asr.configuration_qwen3_asr import Qwen3ASRConfig, Qwen3ASREncoderConfig
    from transformers.models.qwen3_asr.modeling_qwen3_asr import Qwen3ASREncoder, Qwen3ASRMultiModalProjector
    ec = Qwen3ASREncoderConfig(num_mel_bins=128, encoder_layers=cfg.enc_layers, encoder_attention_heads=cfg.enc_heads,
                               encoder_ffn_dim=cfg.enc_ffn, d_model=cfg.enc_d_model, output_dim=cfg.enc_out_dim,
                               downsample_hidden_size=cfg.conv_channels, n_window=50, n_window_infer=800)
    ec._attn_implementation = "eager"
    enc = Qwen3ASREncoder(ec).eval().float()
    sd = {}
    for k, v in w.items():
        if not k.startswith("audio_tower.") or k.startswith("audio_tower.proj"):
            continue
        t = torch.from_numpy(np.ascontiguousarray(v))
        if k.endswith(("conv2d1.weight", "conv2d2.weight", "conv2d3.weight")):
            t = t.permute(0, 3, 1, 2).contiguous()          # MLX [out,kH,kW,in] -> torch [out,in,kH,kW]
        sd[k[len("audio_tower."):]] = t
    missing, unexpected = enc.load_state_dict(sd, strict=False)
    assert not unexpected and all("positional_embedding" in m for m in missing), (missing, unexpected)
    full = Qwen3ASRConfig(audio_config=ec.to_dict())
    proj = Qwen3ASRMultiModalProjector(full).eval().float()
    proj.load_state_dict({"linear_1.weight": torch.from_numpy(w["audio_tower.proj1.weight"]),
                          "linear_1.bias": torch.from_numpy(w["audio_tower.proj1.bias"]),
                          "linear_2.weight": torch.from_numpy(w["audio_tower.proj2.weight"]),
                          "linear_2.bias": torch.from_numpy(w["audio_tower.proj2.bias"])})
    return enc, proj


def g3():
    cfg = make_config("tiny")
    w = synth.checkpoint_to_f32(synth.make_checkpoint(cfg, seed=11))
    enc, proj = hf_encoder(cfg, w)
    for frames in (250, 300, 1000):
        mel = oracle.mel.process(synth.synth_audio(frames * 160, 1000 + frames), 16000)
        padded = -(-frames // 100) * 100
        x = np.zeros((1, 128, padded), dtype=np.float32)
        x[0, :, :frames] = mel
        mask = np.zeros((1, padded), dtype=np.int64)
        mask[0, :frames] = 1
        h = enc(torch.from_numpy(x), torch.from_numpy(mask)).last_hidden_state
        out = proj(h).numpy().astype(np.float32)
        np.savez(os.path.join(HERE, f"g3_encoder_hf_{frames}.npz"), mel=mel.astype(np.float32), out=out)


def g4():
    from transformers import Qwen3Config, Qwen3Model
    cfg = make_config("tiny", vocab=512, rope_interleaved=0)
    ck = synth.make_checkpoint(cfg, seed=13)
    w = synth.checkpoint_to_f32(ck)
    qc = Qwen3Config(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.layers,
                     num_attention_heads=cfg.heads, num_key_value_heads=cfg.kv_heads, head_dim=cfg.head_dim,
                     max_position_embeddings=4096, rms_norm_eps=cfg.rms_eps, rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta},
                     tie_word_embeddings=True, attention_bias=False)
    qc._attn_implementation = "eager"
    model = Qwen3Model(qc).eval().float()
    sd = {k[len("model."):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items() if k.startswith("model.")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not [m for m in missing if "rotary" not in m], (missing, unexpected)
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((37, cfg.hidden)) * 0.05).astype(np.float32)
    emb = torch.from_numpy(w["model.embed_tokens.weight"])
    out = model(inputs_embeds=torch.from_numpy(x)[None], use_cache=True)
    hidden = out.last_hidden_state[0].numpy()
    logits = [(out.last_hidden_state[0, -1] @ emb.T).numpy()]
    ids = [int(np.argmax(logits[-1]))]
    past = out.past_key_values
    for _ in range(8):
        o = model(input_ids=torch.tensor([[ids[-1]]]), past_key_values=past, use_cache=True)
        past = o.past_key_values
        logits.append((o.last_hidden_state[0, -1] @ emb.T).numpy())
        ids.append(int(np.argmax(logits[-1])))
    np.savez(os.path.join(HERE, "g4_decoder_hf.npz"), x=x, prefill_hidden=hidden.astype(np.float32),
             logits=np.stack(logits).astype(np.float32), ids=np.array(ids, dtype=np.int32))


if __name__ == "__main__":
    g1_g2()
    g3()
    g4()
    print("golden vectors written to", HERE)
