"""CPU tests of the host layer: C-ABI exports, config image, weight naming, safetensors / WAV ingest,
tokenizer, and the utterance sharding + gather over torch.distributed (gloo, world size 2)."""
import ctypes
import json
import os
import re
import struct
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from q3asr import make_config, synth
from q3asr import lib as q3lib
from q3asr.audio_io import AudioLoadError, load_wav, resample_linear
from q3asr.config import Q3Config
from q3asr.safetensors_io import load_dir, save_file
from q3asr.shard import shard_utterances, utterance_cost
from q3asr.tokenizer import Qwen3Tokenizer, TokenizerError


# ---------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "q3asr.h")).read()
    declared = sorted(set(re.findall(r"\b(q3asr_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 13
    so = ctypes.CDLL(q3lib.LIB_PATH)
    assert [s for s in declared if not hasattr(so, s)] == []
    assert sorted(q3lib.EXPORTS) == declared


def test_config_struct_matches_header():
    header = open(os.path.join(ROOT, "include", "q3asr.h")).read()
    body = header[header.index("typedef struct q3asr_config {"):header.index("} q3asr_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if decl:
            ctype, names = decl.split(None, 1)
            fields += [(n.strip(), ctype) for n in names.split(",")]
    want = [(n, "float" if t is ctypes.c_float else "int") for n, t in Q3Config._fields_]
    assert fields == want
    assert ctypes.sizeof(Q3Config) == 4 * len(want)


def test_create_without_gpu_fails_loudly():
    """No CPU fallback: on a box without a HIP device create() reports an error instead of a model."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = q3lib.load()
    h = ctypes.c_void_p()
    rc = lib.q3asr_create(ctypes.byref(make_config("tiny")), 0, ctypes.byref(h))
    assert rc != 0 and not h.value
    assert b"HIP" in lib.q3asr_last_error() or b"device" in lib.q3asr_last_error()


# ---------------------------------------------------------------- weights
def test_synthetic_checkpoint_uses_reference_key_names():
    cfg = make_config("tiny")
    ck = synth.make_checkpoint(cfg, seed=0)
    want = {"audio_tower.conv2d1.weight", "audio_tower.conv_out.weight", "audio_tower.ln_post.bias", "audio_tower.proj2.bias",
            "audio_tower.layers.1.self_attn.out_proj.bias", "audio_tower.layers.0.final_layer_norm.weight", "audio_tower.layers.0.fc1.weight",
            "model.embed_tokens.weight", "model.norm.weight", "model.layers.1.self_attn.q_norm.weight",
            "model.layers.0.post_attention_layernorm.weight", "model.layers.1.mlp.down_proj.weight"}
    assert want <= set(ck)
    assert ck["audio_tower.conv2d2.weight"][0].shape == (32, 3, 3, 32)         # [out, kH, kW, in]: WeightLoading.swift:277-284
    assert ck["audio_tower.conv_out.weight"][0].shape == (128, 32 * 16)
    q = synth.make_checkpoint(make_config("tiny", weight_format=1), seed=0)
    assert q["model.layers.0.mlp.gate_proj.weight"][1] == "u32" and q["model.layers.0.mlp.gate_proj.weight"][0].shape == (512, 256 // 8)
    assert q["model.embed_tokens.scales"][0].shape == (151936, 256 // 64)
    assert "audio_tower.layers.0.fc1.scales" not in q                          # the encoder stays dense


def test_safetensors_round_trip_and_bounds(tmp_path):
    ck = {k: v for k, v in synth.make_checkpoint(make_config("tiny", weight_format=1), seed=2).items() if "layers.0." in k or "conv2d1" in k}
    save_file(tmp_path / "model.safetensors", ck)
    back = load_dir(str(tmp_path))
    assert set(back) == set(ck)
    for k, (arr, tag) in ck.items():
        assert back[k][1] == tag and np.array_equal(back[k][0], arr)
    raw = (tmp_path / "model.safetensors").read_bytes()
    (tmp_path / "model.safetensors").write_bytes(struct.pack("<Q", 1 << 40) + raw[8:])
    with pytest.raises(ValueError):
        load_dir(str(tmp_path))
    with pytest.raises(FileNotFoundError):
        load_dir(str(tmp_path / "nothing") if (tmp_path / "nothing").mkdir() is None else "")


# ---------------------------------------------------------------- WAV
def _wav_stub(chunk_id, chunk_size, total):
    """The malformed files of SecurityRegressionTests.swift:16-57."""
    data = bytearray(total)
    data[0:4], data[8:12] = b"RIFF", b"WAVE"
    struct.pack_into("<HH", data, 20, 1, 1)
    struct.pack_into("<I", data, 24, 24000)
    struct.pack_into("<H", data, 34, 16)
    data[36:40] = chunk_id
    struct.pack_into("<I", data, 40, chunk_size)
    return bytes(data)


def test_wav_loader(tmp_path):
    audio, rate = load_wav(os.path.join(GOLDEN, "test_audio.wav"))             # has a LIST chunk before "data"
    assert rate == 24000 and audio.shape == (480000,) and audio.dtype == np.float32 and 0.5 < np.abs(audio).max() < 1.0
    assert len(resample_linear(audio, 24000, 16000)) == 320000
    for name, blob in (("bounds.wav", _wav_stub(b"JUNK", 0x7FFFFFF0, 64)), ("nodata.wav", _wav_stub(b"JUNK", 0, 64)),
                       ("short.wav", b"RIFF")):
        (tmp_path / name).write_bytes(blob)
        with pytest.raises(AudioLoadError):
            load_wav(str(tmp_path / name))


# ---------------------------------------------------------------- tokenizer
def test_tokenizer_toy_vocab_round_trip(tmp_path):
    """The fixture of RealtimeTranslationTests.swift:112-143."""
    vocab = {"H": 0, "e": 1, "l": 2, "o": 3, "Ġ": 4, "w": 5, "r": 6, "d": 7, "!": 8}
    (tmp_path / "vocab.json").write_text(json.dumps(vocab), encoding="utf-8")
    with pytest.raises(TokenizerError):
        Qwen3Tokenizer.load(str(tmp_path / "vocab.json"))                     # merges.txt is required
    (tmp_path / "merges.txt").write_text("#version: 0.2\n", encoding="utf-8")
    tok = Qwen3Tokenizer.load(str(tmp_path / "vocab.json"))
    ids = tok.encode("Hello world!")
    assert ids == [0, 1, 2, 2, 3, 4, 5, 3, 6, 2, 7, 8]
    assert tok.decode(ids) == "Hello world!"


def test_tokenizer_merges_specials_and_utf8(tmp_path):
    vocab = {"l": 0, "o": 1, "w": 2, "lo": 3, "low": 4, "Ġ": 5, "Ġlow": 6, "Ã": 7, "©": 8, "<asr_text>": 20}
    (tmp_path / "vocab.json").write_text(json.dumps(vocab), encoding="utf-8")
    (tmp_path / "merges.txt").write_text("#version: 0.2\nl o\nlo w\nĠ low\n", encoding="utf-8")
    cfg = {"added_tokens_decoder": {"30": {"content": "<|im_end|>"}, "31": {"content": "<|audio_pad|>"}}}
    (tmp_path / "tokenizer_config.json").write_text(json.dumps(cfg), encoding="utf-8")
    tok = Qwen3Tokenizer.load(str(tmp_path / "vocab.json"))
    assert tok.encode("low low<asr_text>low<|im_end|>") == [4, 6, 20, 4, 30]
    assert tok.decode([4, 6, 30, 20, 31, 4]) == "low low<asr_text>low"       # "<|...|>" dropped, "<asr_text>" kept
    assert tok.decode([7, 8]) == "é"                                          # bytes C3 A9 split over two tokens
    assert tok.post_process(tok.decode([4, 20, 6])) == "low"


# ---------------------------------------------------------------- long audio
def test_long_audio_split_points_and_join():
    from q3asr.segment import frame_rms, join_tokens, segment_batch, split_points
    sr = 16000
    rng = np.random.default_rng(7)
    audio = (rng.standard_normal(sr * 75) * 0.2).astype(np.float32)
    audio[sr * 27:sr * 27 + 3200] = 0.0                      # a 200 ms pause 3 s before the first limit
    spans = split_points(audio, sr)
    assert spans[0][0] == 0 and spans[-1][1] == len(audio)
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))            # exact cover, no overlap
    assert all(0 < e - s <= 30 * sr for s, e in spans)
    assert sr * 27 <= spans[0][1] < sr * 27 + 3200                        # the cut lands inside the pause
    assert frame_rms(audio[spans[0][1]:spans[0][1] + 320], 320)[0] < 0.01
    fixed = split_points(audio, sr, energy=False)
    assert [e - s for s, e in fixed] == [30 * sr, 30 * sr, 15 * sr]
    assert split_points(audio[:sr * 30], sr) == [(0, sr * 30)]            # at the limit: one piece
    pieces, owner = segment_batch([audio, audio[:sr * 5]], sr, mode="segment_fixed")
    assert owner == [0, 0, 0, 1] and [len(p) for p in pieces] == [30 * sr, 30 * sr, 15 * sr, 5 * sr]
    pieces, owner = segment_batch([audio], sr, mode="truncate")
    assert owner == [0] and len(pieces[0]) == len(audio)                  # the library truncates, as the reference does
    with pytest.raises(ValueError):
        segment_batch([audio], sr, mode="overlap")
    assert join_tokens([[5, 6, 151645], [7, 151643], [9]], [0, 0, 1], 2) == [[5, 6, 7], [9]]


# ---------------------------------------------------------------- sharding
def test_shard_balance_and_cover():
    rng = np.random.default_rng(4321)
    lengths = (rng.uniform(5, 300, size=203) * 16000).astype(int).tolist()
    for world in (1, 2, 4, 8):
        shards = shard_utterances(lengths, world)
        assert sorted(i for s in shards for i in s) == list(range(len(lengths)))
        loads = [sum(utterance_cost(lengths[i]) for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(utterance_cost(n) for n in lengths)
        for s in shards:
            costs = [utterance_cost(lengths[i]) for i in s]
            assert costs == sorted(costs, reverse=True)                       # each rank's list is length-bucketed
    assert utterance_cost(480000) == 390 + 15 + 448


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from q3asr.shard import transcribe_sharded
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    audios = [np.zeros(16000 * (1 + (i * 7) % 11), np.float32) + i for i in range(9)]
    seen = []

    def fake_transcribe(clips):                  # stands in for Qwen3ASRModel.transcribe_batch on this rank's GPU
        seen.extend(int(c[0]) for c in clips)
        return [[int(c[0])] * (int(c[0]) % 4) for c in clips]

    tokens, audio_s, n_tok = transcribe_sharded(fake_transcribe, audios, rank, world, dist)
    q.put((rank, seen, tokens, audio_s, n_tok))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_by_given_costs():
    """A recording that will be cut into pieces is placed by the summed cost of its pieces, not by its own length."""
    from q3asr.shard import shard_utterances
    lens = [16000 * 300, 16000 * 30, 16000 * 30, 16000 * 30]
    by_length = shard_utterances(lens, 2)                     # every clip capped at 30 s of work: 2 + 2
    assert sorted(len(s) for s in by_length) == [2, 2]
    by_cost = shard_utterances(lens, 2, costs=[10.0, 1.0, 1.0, 1.0])
    assert sorted(by_cost, key=len) == [[0], [1, 2, 3]]
    assert sorted(i for s in by_cost for i in s) == [0, 1, 2, 3]


def test_sharded_transcribe_gathers_over_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = sorted(i for _, s, _, _, _ in out for i in s)
    assert seen == list(range(9))                                             # every utterance on exactly one rank
    want = [[i] * (i % 4) for i in range(9)]
    for _, s, tokens, audio_s, n_tok in out:
        assert tokens == want                                                 # every rank ends with every transcript
        assert 0 < len(s) < 9
        assert n_tok == sum(len(t) for t in want)
        assert abs(audio_s - sum(1 + (i * 7) % 11 for i in range(9))) < 1e-9


def test_bench_launcher_starts_one_rank_per_gpu(tmp_path):
    """`python bench.py --gpus N` without a launcher starts N ranks itself: every child gets its own RANK / LOCAL_RANK,
    the shared WORLD_SIZE / MASTER_ADDR / MASTER_PORT and the original arguments; rank 0's output is passed through
    and a failing rank fails the run."""
    import importlib.util
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    envs = bench.rank_envs(4, 29517, base_env={"PATH": "/bin"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29517" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/bin" for e in envs)
    child = tmp_path / "child.py"
    child.write_text(
        "import json, os, sys\n"
        "keys = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')\n"
        "open(os.path.join(os.path.dirname(__file__), 'rank%s.json' % os.environ['RANK']), 'w').write(\n"
        "    json.dumps({'env': {k: os.environ[k] for k in keys}, 'argv': sys.argv[1:]}))\n"
        "print(json.dumps({'n_gpus': int(os.environ['WORLD_SIZE'])}))\n"
        "sys.exit(3 if '--fail' in sys.argv and os.environ['RANK'] == '1' else 0)\n")
    driver = ("import sys; sys.path.insert(0, %r); import bench; bench.spawn_ranks(2, %%r, child=[sys.executable, %r])" % (root, str(child)))
    run = subprocess.run([sys.executable, "-c", driver % (["--gpus", "2", "--steps", "1"],)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    assert json.loads(run.stdout.strip().splitlines()[-1]) == {"n_gpus": 2}            # exactly rank 0's line
    seen = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(2)]
    assert [s["env"]["RANK"] for s in seen] == ["0", "1"] and all(s["env"]["WORLD_SIZE"] == "2" for s in seen)
    assert seen[0]["env"]["MASTER_PORT"] == seen[1]["env"]["MASTER_PORT"] and seen[0]["env"]["MASTER_ADDR"] == "127.0.0.1"
    assert all(s["argv"] == ["--gpus", "2", "--steps", "1"] for s in seen)
    run = subprocess.run([sys.executable, "-c", driver % (["--fail"],)], capture_output=True, text=True, timeout=120)
    assert run.returncode != 0 and "rank 1 ended with exit code 3" in run.stderr


def test_bench_launcher_ends_the_job_when_one_rank_dies(tmp_path):
    """A rank that dies (out of memory, bad device ordinal) while its siblings sit in a collective: the launcher stops
    the siblings and returns non-zero within seconds, without rank 0's line -- instead of waiting for the backend's
    timeout or the driver's limit."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = tmp_path / "child.py"
    child.write_text(
        "import os, sys, time, signal\n"
        "if os.environ['RANK'] == '1':\n"
        "    time.sleep(0.5); sys.exit(3)\n"
        "if os.environ['RANK'] == '2':\n"
        "    signal.signal(signal.SIGTERM, signal.SIG_IGN)     # a rank stuck so deep that it ignores the request to stop\n"
        "print('{\"n_gpus\": 3}', flush=True)\n"
        "time.sleep(600)\n")
    driver = ("import sys; sys.path.insert(0, %r); import bench; bench.spawn_ranks(3, [], child=[sys.executable, %r], grace_s=2.0)" % (root, str(child)))
    t0 = time.monotonic()
    run = subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, timeout=60)
    took = time.monotonic() - t0
    assert run.returncode != 0 and "rank 1 ended with exit code 3" in run.stderr
    assert took < 10.0, took
    assert run.stdout.strip() == ""                      # a partial job prints no result line


def test_config_json_fields_the_reference_reads():
    """loadTextConfigIfPresent (Qwen3ASR.swift:774-818): rope_scaling.interleaved and mrope_section
    (rotaryDim = 2 * sum, clamped to the head, Configuration.swift:75-80) and quantization.group_size / bits."""
    from q3asr.model import config_from_json
    preset, kw = config_from_json({"thinker_config": {"text_config": {
        "hidden_size": 2048, "head_dim": 128, "rope_theta": 1000000, "rope_scaling": {"interleaved": False, "mrope_section": [8, 12, 12]}}},
        "quantization": {"group_size": 64, "bits": 4}})
    assert preset == "1.7b" and kw["rope_interleaved"] == 0 and kw["rotary_dim"] == 64
    assert kw["weight_format"] == 1 and kw["group_size"] == 64 and kw["bits"] == 4
    preset, kw = config_from_json({"thinker_config": {"text_config": {"hidden_size": 1024, "rope_scaling": {"interleaved": True, "mrope_section": [24, 20, 20]}}}})
    assert preset == "0.6b" and kw["rope_interleaved"] == 1 and kw["rotary_dim"] == 128 and kw["weight_format"] == 0
    preset, kw = config_from_json({"thinker_config": {"text_config": {}}})
    assert "rope_interleaved" not in kw and "rotary_dim" not in kw        # defaults: interleaved, full rotary
    with pytest.raises(ValueError):
        config_from_json({"thinker_config": {"text_config": {"rope_scaling": {"mrope_section": []}}}})
    # a tail piece shorter than 100 ms is folded into the previous cut instead of becoming an utterance of its own
    from q3asr.segment import split_points
    n = 16000 * 60 + 37
    spans = split_points(np.zeros(n, np.float32), 16000, energy=False)
    assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert all(1600 <= e - s <= 480000 for s, e in spans)
    assert len(split_points(np.zeros(480000, np.float32), 16000)) == 1


def test_bench_traffic_figure_follows_the_decode_mode(monkeypatch):
    """bench.py's roofline.traffic comes from committed counter passes: the default run (two halves per launch pair)
    and the one-stream run (Q3ASR_DECODE_ATT_CUS=0) each have their own summary, any other share or workload has none."""
    import argparse
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    args = argparse.Namespace(preset="1.7b", max_tokens=448, clip_seconds=30.0, weights="bf16", workload="transcribe",
                              audio="speechlike", length_dist="none", batch=2048)
    monkeypatch.delenv("Q3ASR_DECODE_ATT_CUS", raising=False)
    split_bytes, split_src = bench.pmc_traffic("decode_attn_kernel", args)
    monkeypatch.setenv("Q3ASR_DECODE_ATT_CUS", "0")
    one_bytes, one_src = bench.pmc_traffic("decode_attn_kernel", args)
    assert split_src.endswith("_pmc_decode_attn_b2048.json") and one_src.endswith("_pmc_decode_attn_b2048_one_stream.json")
    assert 1.9 < one_bytes / split_bytes < 2.1           # one launch covers the whole group instead of a half
    monkeypatch.setenv("Q3ASR_DECODE_ATT_CUS", "176")
    assert bench.pmc_traffic("decode_attn_kernel", args) == (None, None)
    monkeypatch.delenv("Q3ASR_DECODE_ATT_CUS")
    args.max_tokens = 96
    assert bench.pmc_traffic("decode_attn_kernel", args) == (None, None)


def test_kernel_stats_summary_from_a_trace(tmp_path):
    """scripts/kernel_stats.py: per-kernel calls / total / average from a rocprofv3 kernel trace."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = tmp_path / "prof" / "host" / "1"
    d.mkdir(parents=True)
    rows = ["Kind,Agent_Id,Queue_Id,Kernel_Name,Start_Timestamp,End_Timestamp",
            "KERNEL_DISPATCH,1,1,kern_a,1000,3000", "KERNEL_DISPATCH,1,2,kern_b,1500,2000", "KERNEL_DISPATCH,1,1,kern_a,4000,8000"]
    (d / "t_kernel_trace.csv").write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "kernel_stats.py"), str(tmp_path / "prof")],
                         check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert out[0] == "Name,Calls,TotalDurationUs,AverageUs,Percentage"
    assert out[1].startswith("kern_a,2,6.0,3.0,") and out[2].startswith("kern_b,1,0.5,0.5,")
