#!/usr/bin/env python3
"""RTFx benchmark of the Qwen3-ASR offline-transcription hot path on MI355X.

One "step" = one pass of the whole hot path (log-mel -> audio encoder -> prompt splice -> decoder
prefill -> greedy KV-cache decode with the reference's stop rules) over a batch of synthetic 30 s
clips that already sit in HBM.  Metric: RTFx = audio seconds transcribed / wall seconds, summed
over all ranks (weak scaling: every GPU gets its own --batch clips, utterance-sharded, no data-path
collective).  Weights are seeded random tensors of the true Qwen3-ASR-1.7B shapes (no checkpoint
exists offline); with such weights the decoder never emits EOS, so every clip runs the reference's
full max_tokens = 448 greedy steps -- the most expensive case the reference's loop allows.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python bench.py --gpus N --steps K --warmup W          (starts its own N ranks, one per GPU)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# Multi-process GPU work on this platform needs dmabuf IPC (RCCL's peer mappings fail with "hipIpcGetMemHandle: invalid
# argument" under the legacy mode); set before anything initialises HIP, and only if the caller has not decided.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))

T_START = time.perf_counter()
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16


def host_cpu_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(cfg, ck, audio, decode_steps, sample_rate=16000, what="1 clip x 30 s, 1.7B dims"):
    """CPU restatement of the reference algorithm (numpy fp32; NOT the reference's Swift/MLX build,
    which cannot be produced here), timed on this box's host cores on one clip of the same workload.
    Decode is timed for `decode_steps` greedy steps; the remaining steps of the 448 tokens the GPU run executes are
    charged by a linear fit of the measured per-step time over the growing context.  Beside it, the reference
    frontend's own arithmetic -- a direct DFT per frame and bin on one thread (AudioPreprocessing.swift:184-240) --
    is timed through its plain-C restatement (oracle/mel_direct.c)."""
    import oracle
    from q3asr import oracle_cfgs, synth
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(threads, int(os.environ.get("OMP_NUM_THREADS", threads)))
    w = synth.checkpoint_to_f32(ck)
    enc_cfg, dec_cfg = oracle_cfgs(cfg)
    try:
        a16 = oracle.mel.resample_linear(audio, sample_rate, 16000) if sample_rate != 16000 else audio
        _, direct_s = oracle.mel.extract_features_c(a16[:480000])
        direct = f"; the reference's direct-DFT log-mel alone, plain C on one thread: {direct_s:.2f} s"
    except (RuntimeError, OSError) as e:
        direct = f"; (direct-DFT C restatement not timed: {e})"
    t0 = time.perf_counter()
    feats = oracle.mel.process(audio, sample_rate)
    t1 = time.perf_counter()
    a = oracle.encoder.encode(feats, w, enc_cfg)
    t2 = time.perf_counter()
    dec = oracle.decoder.Decoder(w, dec_cfg)
    ids, a0, a1 = oracle.generate.prompt_ids(a.shape[0])
    x = oracle.generate.splice(dec, ids, a0, a1, a, bool(cfg.embed_scale))
    t3 = time.perf_counter()
    h, cache = dec.forward(x, None)
    nxt = int(np.argmax(dec.logits(h[-1:])[0]))
    t4 = time.perf_counter()
    per = []
    for _ in range(decode_steps):
        ts = time.perf_counter()
        h, cache = dec.forward(dec.embed([nxt]), cache)
        nxt = int(np.argmax(dec.logits(h)[0]))
        per.append(time.perf_counter() - ts)
    # a step's cost grows with the cached context: fit t(step) = a + b * step on the measured steps and sum the fit
    # over all 447 steps of the GPU run's 448 tokens (the measured steps count as measured)
    idx = np.arange(1, decode_steps + 1, dtype=np.float64)
    c1, c0 = np.polyfit(idx, np.asarray(per), 1) if decode_steps >= 8 else (0.0, float(np.mean(per)))
    rest = np.arange(decode_steps + 1, 448, dtype=np.float64)
    decode_total = float(np.sum(per)) + float(np.sum(c0 + max(c1, 0.0) * rest))
    total = (t1 - t0) + (t2 - t1) + (t4 - t3) + decode_total
    seconds = min(len(audio) / float(sample_rate), 30.0)        # the reference drops what follows 30 s
    return {
        "value": round(seconds / total, 4), "unit": "RTFx (audio s per wall s)", "cores": threads, "kind": "port",
        "sample": f"{what}, numpy fp32 restatement of the reference algorithm (not its Swift/MLX build) on {host_cpu_name()}, {threads} threads: "
                  f"mel {t1 - t0:.2f} s + encoder {t2 - t1:.2f} s + prefill "
                  f"{t4 - t3:.2f} s + {decode_steps} decode steps {sum(per):.2f} s measured; steps {decode_steps + 1}-447 from a linear fit "
                  f"of the measured per-step times over the context length ({c0 * 1e3:.1f} ms + {max(c1, 0.0) * 1e6:.1f} us per cached token): "
                  f"{decode_total:.2f} s of decode in all" + direct,
        "stage_seconds": {"mel": round(t1 - t0, 3), "encoder": round(t2 - t1, 3), "prefill": round(t4 - t3, 3), "decode": round(decode_total, 3)},
    }


def run_c1(args, emit):
    """The reference's own way of running (CLI transcribe, main.swift:297-315): the 20 s test clip, loaded at 24 kHz,
    one utterance, 0.6B dimensions -- the GPU's batch-1 call beside the CPU restatement, stage by stage."""
    import torch
    from q3asr import Qwen3ASRModel, make_config, synth
    from q3asr.audio_io import load_wav
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libq3asr has no CPU path")
    audio, rate = load_wav(os.path.join(ROOT, "tests", "golden", "test_audio.wav"))
    cfg = make_config("0.6b", max_batch=1, weight_format=1 if args.weights == "q4" else 0)
    ck = synth.make_checkpoint(cfg, seed=0, workers=8)
    model = Qwen3ASRModel.from_checkpoint(cfg, ck, device=0)
    run = lambda: model.transcribe_batch([audio], rate, max_tokens=args.max_tokens, stop_rules=not args.no_stop_rules)
    for _ in range(max(1, args.warmup)):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = [run() for _ in range(args.steps)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ms = {k: round(sum(o.ms[k] for o in outs) / args.steps, 2) for k in outs[0].ms}
    model.close()
    seconds = len(audio) / float(rate)
    line = {"metric": f"RTFx (audio s per wall s), 0.6B {args.weights}, the reference's test clip, batch 1", "value": round(seconds / dt, 2),
            "unit": "x real-time", "n_gpus": 1, "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": round(dt * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "tests/golden/test_audio.wav (the "
            "reference's Tests/Qwen3ASRTests/Resources/test_audio.wav: 20.0 s, 24 kHz mono), seeded random weights",
            "config": {"workload": f"c1: one utterance, the whole call from host samples at {rate} Hz to host token ids (resample, log-mel, encoder, "
                                   f"prefill, {outs[0].tokens_total} greedy tokens), Qwen3-ASR-0.6b {args.weights}",
                       "tokens_per_clip": outs[0].tokens_total, "stage_ms_per_step_rank0": ms},
            "roofline": None}
    line["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(cfg, ck, audio, args.cpu_decode_steps, sample_rate=rate,
                                                                          what="the same clip (20 s at 24 kHz), 0.6B dims, batch 1")
    emit(line)
    return 0


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 counter passes (FETCH_SIZE doubled as gfx950
    needs, plus WRITE_SIZE; the passes cannot run inside this process).  Only a summary taken on this exact
    workload counts; anything else reports None."""
    import glob
    if (args.preset != "1.7b" or args.max_tokens != 448 or args.clip_seconds != 30.0 or args.weights != "bf16"
            or args.workload == "ragged" or args.audio != "speechlike" or args.length_dist != "none"):
        return None, None
    short = kernel.split("<")[0].replace("_kernel", "")
    # the passes exist for the library's own decode (two halves per launch pair) and for one stream (Q3ASR_DECODE_ATT_CUS=0)
    share = os.environ.get("Q3ASR_DECODE_ATT_CUS")
    if share not in (None, "0"):
        return None, None
    suffix = "_one_stream" if share == "0" and short == "decode_attn" else ""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{short}_b{args.batch}{suffix}.json")), reverse=True):
        with open(path) as f:
            d = json.load(f)
        return d["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def measured_copy_ceiling(torch, nbytes=1 << 30, reps=10):
    """What a plain device-to-device copy reaches on this GPU (read + write bytes per second, GB/s): the practical
    HBM ceiling beside the vendor peak the roofline fraction is quoted against."""
    src = torch.empty(nbytes, dtype=torch.uint8, device="cuda").random_(0, 255)
    dst = torch.empty_like(src)
    dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return round(2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)


def rank_envs(n, port, base_env=None):
    """Environment of each of the n ranks the launcher starts (what torch.distributed's env:// rendezvous reads)."""
    envs = []
    for r in range(n):
        e = dict(os.environ if base_env is None else base_env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=e.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        envs.append(e)
    return envs


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv, child=None, poll_s=0.2, grace_s=5.0):
    """`python bench.py --gpus N` without a launcher: start N copies of this script, one rank per GPU, before this
    process has touched the GPU, and watch all of them.  Rank 0's result line is passed through once every rank has
    ended well.  The first rank that ends badly ends the job: its siblings (which would otherwise sit in a collective
    until the backend's timeout) are terminated, killed if they ignore that for `grace_s` seconds, and the launcher
    exits non-zero naming the rank and its exit code."""
    import subprocess
    import tempfile
    child = child or [sys.executable, os.path.abspath(__file__)]
    port = int(os.environ.get("MASTER_PORT") or free_port())
    procs = []
    with tempfile.TemporaryFile(mode="w+") as out0:
        for r, env in enumerate(rank_envs(n, port)):
            procs.append(subprocess.Popen(child + list(argv), env=env, stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
        bad = None
        while bad is None and any(p.poll() is None for p in procs):
            time.sleep(poll_s)
            for r, p in enumerate(procs):
                if p.poll() not in (None, 0):
                    bad = (r, p.returncode)
                    break
        if bad is None:
            bad = next(((r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0), None)
        if bad is not None:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            deadline = time.monotonic() + grace_s
            for p in procs:
                try:
                    p.wait(timeout=max(0.0, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            raise SystemExit(f"bench.py: rank {bad[0]} ended with exit code {bad[1]}; the other ranks were stopped, no result")
        out0.seek(0)
        sys.stdout.write(out0.read())
        sys.stdout.flush()
    return 0


def scripted_lengths(kind, n, max_tokens, seed):
    """Tokens each clip emits before its EOS under a labelled synthetic length distribution (random weights never
    emit EOS on their own).  "u60-150+straggler": U[60, 150] tokens, and one clip of every batch runs to max_tokens."""
    lo, hi = (int(x) for x in kind.split("+")[0][1:].split("-"))
    lens = np.random.default_rng(seed).integers(lo, hi + 1, size=n)
    lens = np.minimum(lens, max_tokens)
    if kind.endswith("+straggler") and n > 0:
        lens[n // 2] = max_tokens
    return lens


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--preset", default="1.7b")
    ap.add_argument("--batch", type=int, default=2048, help="30 s clips per GPU per step (2048: 200 GB of KV cache, the largest "
                    "power of two that fits 288 GB; utterances decoded together)")
    ap.add_argument("--clip-seconds", type=float, default=30.0)
    ap.add_argument("--max-tokens", type=int, default=448)
    ap.add_argument("--no-stop-rules", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-decode-steps", type=int, default=128)
    ap.add_argument("--profile-stride", type=int, default=17, help="time every n-th launch with HIP events (odd, so that the "
                    "samples walk through the per-layer launch pattern)")
    ap.add_argument("--no-profile", action="store_true", help="no per-kernel HIP-event timing (and no roofline object): what a "
                    "latency measurement of small groups needs, since timed launches are not replayed as a graph")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path on a box with one GPU, together with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--collective-timeout", type=float, default=240.0, help="seconds after which a collective that a rank never "
                    "joins is given up (the process group's timeout): ranks differ by their set-up time at most")
    ap.add_argument("--rehearse-collectives", action="store_true", help="one rank only: open a process group of one rank on "
                    "--backend anyway and run the job's exchange (all_gather of token lists, all_reduce of counters, barriers) "
                    "through it -- the RCCL calls of a multi-GPU run, on a box with one GPU")
    ap.add_argument("--leak-model", action="store_true", help="diagnostic: end the process without q3asr_destroy, like a host whose "
                    "finaliser never runs (the library's own exit handler then has to release the model)")
    ap.add_argument("--workload", default="transcribe", choices=["transcribe", "mel", "ragged", "c1"],
                    help="c1: the reference's test clip through the 0.6B model, one utterance, GPU beside the CPU restatement.  transcribe: the headline run, --batch clips of --clip-seconds per GPU.  mel: the log-mel frontend alone on the "
                         "same clips (device buffers in and out).  ragged: --utterances recordings per GPU with durations U[5, 300] s "
                         "(seed 4321), cut into pieces of at most 30 s, cost-sharded over the ranks, length-bucketed by the library")
    ap.add_argument("--weights", default="bf16", choices=["bf16", "q4"], help="q4: MLX affine 4-bit group-quantised decoder weights")
    ap.add_argument("--audio", default="speechlike", choices=["speechlike", "noise", "gapped"],
                    help="gapped: speech-like with a quarter of every clip digitally silent (the log-mel clamp then has work to do)")
    ap.add_argument("--utterances", type=int, default=256, help="ragged workload: recordings per GPU")
    ap.add_argument("--length-dist", default="none", help="none: every clip decodes until the stop rules or max_tokens end it (with "
                    "random weights: max_tokens).  uLO-HI[+straggler]: scripted lengths -- clip u stops on an EOS after U[LO, HI] tokens "
                    "(teacher-forced inputs, the reference's stop rules applied to them), optionally one clip per batch running to max_tokens")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    # stdout carries the one JSON line and nothing else: whatever libraries print there (RCCL announces its version on
    # stdout when a process group starts) goes to stderr instead
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    if args.workload == "c1":
        raise SystemExit(run_c1(args, emit))

    import datetime
    import torch
    import torch.distributed as dist
    from q3asr import Qwen3ASRModel, make_config, synth

    t_setup = time.perf_counter()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libq3asr has no CPU path")
    torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"   # where the collectives' tensors live
    group = world > 1 or args.rehearse_collectives       # a process group exists and the exchange runs through it
    if group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        timeout = datetime.timedelta(seconds=args.collective_timeout)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=timeout)
        else:
            dist.init_process_group(args.backend, timeout=timeout)
        world = dist.get_world_size()      # what the group really holds, not what the environment promised
        rank = dist.get_rank()

    # Set-up is host work that every rank of a node does at once on its share of the cores: the weights are drawn on
    # a few threads per rank, and a rank's clips repeat 64 distinct seeded ones (kernel times do not depend on the
    # sample values; the tests' full-size batches are built the same way).
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    workers = max(2, min(16, cores // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))))
    cfg = make_config(args.preset, max_batch=args.batch, weight_format=1 if args.weights == "q4" else 0)
    ck = synth.make_checkpoint(cfg, seed=0, workers=workers)
    model = Qwen3ASRModel.from_checkpoint(cfg, ck, device=local_rank)
    if args.workload == "ragged":
        # Every rank derives the same recording list, keeps its cost-balanced share and numbers the pieces globally.
        from q3asr.shard import shard_utterances, utterance_cost
        piece = 480000
        lens = (np.random.default_rng(4321).uniform(5.0, 300.0, args.utterances * world) * 16000).astype(np.int64)
        spans = [[(s, min(s + piece, int(L))) for s in range(0, int(L), piece)] for L in lens]
        costs = [sum(utterance_cost(e - s, args.max_tokens) for s, e in sp) for sp in spans]
        first_piece = np.concatenate([[0], np.cumsum([len(sp) for sp in spans])])
        mine = shard_utterances(lens, world, args.max_tokens, costs=costs)[rank]
        base = [synth.synth_audio(piece, 1234 + i, args.audio) for i in range(8)]
        clips = [np.tile(base[i % 8], len(spans[i]))[:int(lens[i])] for i in mine]
        piece_ids = [int(first_piece[i]) + k for i in mine for k in range(len(spans[i]))]
        bounds = [0]
        for i in mine:
            for s, e in spans[i]:
                bounds.append(bounds[-1] + e - s)
        offsets = np.asarray(bounds, dtype=np.int64)
        n_units_total = int(first_piece[-1])
        what = (f"{args.utterances} recordings per GPU, durations U[5, 300] s (seed 4321), cut every 30 s into "
                f"{n_units_total} pieces over {world} GPU(s), cost-sharded, groups of {args.batch} length-sorted pieces")
    else:
        n = int(args.clip_seconds * 16000)
        base = [synth.synth_audio(n, 1234 + rank * 64 + i, args.audio) for i in range(min(64, args.batch))]
        clips = [base[i % len(base)] for i in range(args.batch)]
        piece_ids = [rank * args.batch + i for i in range(args.batch)]
        offsets = np.arange(args.batch + 1, dtype=np.int64) * n
        n_units_total = world * args.batch
        what = f"{args.batch} x {args.clip_seconds:g} s synthetic 16 kHz clips per GPU per step ({len(base)} distinct seeded clips, repeated)"
    forced = None
    if args.length_dist != "none":
        lens_tok = scripted_lengths(args.length_dist, len(piece_ids), args.max_tokens, 99 + rank)
        forced = np.random.default_rng(7 + rank).integers(1000, 100000, size=(len(piece_ids), args.max_tokens)).astype(np.int32)
        for u, L in enumerate(lens_tok):
            if L < args.max_tokens:
                forced[u, L - 1] = 151645           # <|im_end|>: kept as the clip's last token (Qwen3ASR.swift:571-574)
    dev_audio = torch.from_numpy(np.concatenate(clips)).cuda()
    clips = clips[:1]          # only the CPU baseline still needs a host copy (of one clip)
    if args.workload == "mel":
        mel_out = torch.empty((len(piece_ids), 128, 3000), dtype=torch.float32, device="cuda")

    class MelOnly:
        tokens_total, ms = 0, {}

    def step():
        if args.workload == "mel":
            frames = model.feature_extractor.process_device(dev_audio.data_ptr(), offsets, mel_out.data_ptr(), 3000)
            out = MelOnly()
            out.audio_seconds = float(np.minimum(np.diff(offsets), 480000).sum()) / 16000.0
            assert int(frames.min()) > 0
            return out
        out = model.transcribe_batch(None, 16000, max_tokens=args.max_tokens, device_audio=(dev_audio.data_ptr(), offsets),
                                     stop_rules="forced" if forced is not None else not args.no_stop_rules, forced_tokens=forced)
        if group:
            # the job's one exchange: every rank ends the batch holding every transcript (token ids)
            from q3asr.shard import gather_token_lists
            local = dict(zip(piece_ids, out.tokens))
            everything = gather_token_lists(local, n_units_total, dist, device=coll_dev, always=True)
            assert len(everything) == n_units_total and all(everything[i] == t for i, t in local.items())
        return out

    def fence():
        torch.cuda.synchronize()
        if group:
            dist.barrier()
        torch.cuda.synchronize()

    last_note = [T_START]

    def note(what):
        # a line on stderr at most every 20 s: a long run shows that it is alive (stdout stays reserved for the result)
        now = time.perf_counter()
        if rank == 0 and now - last_note[0] >= 20.0:
            last_note[0] = now
            print(f"[bench] {what}", file=sys.stderr, flush=True)

    setup_s = time.perf_counter() - t_setup
    # who is really there: every rank's device and set-up time, gathered through the group
    me = {"rank": rank, "device": local_rank, "name": torch.cuda.get_device_name(local_rank), "setup_s": round(setup_s, 1)}
    ranks_seen = [me]
    if group:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, me)
    note("weights and audio resident, starting the warmup steps")
    for i in range(args.warmup):
        step()
        note(f"warmup step {i + 1} of {args.warmup} done")
    if not args.no_profile:
        model.profile_start(args.profile_stride)
    fence()
    t0 = time.perf_counter()
    outs = []
    for i in range(args.steps):
        outs.append(step())
        note(f"timed step {i + 1} of {args.steps} issued and returned")
    fence()
    dt = time.perf_counter() - t0
    prof = [] if args.no_profile else model.profile_stop()

    audio_s = sum(o.audio_seconds for o in outs)
    tokens = sum(o.tokens_total for o in outs)
    row_steps = sum(getattr(o, "decode_rows", 0) for o in outs)
    stats = torch.tensor([dt, audio_s, float(tokens), float(row_steps)], dtype=torch.float64, device=coll_dev)
    if group:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
    audio_s, tokens, row_steps = float(stats[1]), float(stats[2]), float(stats[3])
    lengths = ("every clip runs to max_tokens (random weights never emit EOS)" if args.length_dist == "none" else
               f"scripted lengths {args.length_dist}: clips stop on a teacher-forced EOS, finished clips leave the decode steps"
               + (" (retirement off)" if os.environ.get("Q3ASR_RETIRE") == "0" else ""))

    n_local = len(piece_ids)
    if args.workload == "mel":
        metric = "RTFx (audio s per wall s) of the log-mel frontend alone, 30 s clips"
        label = f"log-mel frontend only (resident fp32 samples -> normalised [128][3000] fp32 log-mel in HBM), {args.audio} audio, {what}"
    elif args.workload == "ragged":
        metric = f"RTFx (audio s per wall s) node-wide, {args.preset} {args.weights}, 5-300 s recordings in <= 30 s pieces"
        label = (f"Qwen3-ASR-{args.preset} {args.weights} full transcribe (reference stop rules, max_tokens {args.max_tokens}), {what}, "
                 f"audio resident in HBM, seeded random weights of the true shapes; {lengths}")
    else:
        metric = ("RTFx (audio s per wall s) node-wide, 1.7B bf16, 30 s clips" if (args.preset, args.weights) == ("1.7b", "bf16")
                  else f"RTFx (audio s per wall s) node-wide, {args.preset} {args.weights}, 30 s clips")
        label = (f"Qwen3-ASR-{args.preset} {args.weights} full transcribe (mel+encoder+prefill+greedy decode, reference stop "
                 f"rules, max_tokens {args.max_tokens}), {what}, audio resident in HBM, seeded random weights of the true shapes; {lengths}")
    if args.share_gpu or args.rehearse_collectives:
        # a run that shares one GPU between ranks, or pushes one rank through the collectives, checks plumbing: its
        # number must not read like a throughput result
        metric = "REHEARSAL of the multi-rank path (not a throughput result): " + metric
    if rank == 0 and not prof:
        ms = {k: sum(o.ms[k] for o in outs) / args.steps for k in outs[0].ms}
        emit({"metric": metric, "value": round(audio_s / dt, 2),
                          "unit": "x real-time", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32" if args.workload == "mel" else "bf16", "data": "synthetic",
                          "config": {"workload": "run without kernel timing: " + label, "clips_per_gpu": n_local,
                                     "stage_ms_per_step_rank0": {k: round(v, 2) for k, v in ms.items()},
                                     "ranks_seen": len(ranks_seen), "ranks": ranks_seen},
                          "roofline": None, "cpu_baseline": None})
    elif rank == 0:
        # dominant kernel by device time (sampled HIP-event timings scaled to all launches)
        for p in prof:
            p["ms_est"] = p["ms_timed"] * p["launches"] / max(p["timed"], 1)
        prof.sort(key=lambda p: -p["ms_est"])
        top = prof[0]
        per_launch_work = top["work_timed"] / max(top["timed"], 1)
        avg_ms = top["ms_timed"] / max(top["timed"], 1)
        if top["name"].startswith(("gemm", "flash")):
            ach, peak, unit, bound = per_launch_work / (avg_ms * 1e-3) / 1e12, MFMA_BF16_PEAK_TFLOPS, "TFLOP/s", "mfma"
        else:
            ach, peak, unit, bound = per_launch_work / (avg_ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
        ms = {k: sum(o.ms[k] for o in outs) / args.steps for k in outs[0].ms}
        traffic, traffic_src = pmc_traffic(top["name"], args)
        copy_gbs = measured_copy_ceiling(torch)
        read_gbs = round(model.op_read_bandwidth(1 << 32), 1)   # streaming read of 4 GiB: what an HBM-read-bound kernel can reach
        line = {
            "metric": metric, "value": round(audio_s / dt, 2),
            "unit": "x real-time", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.workload == "mel" else "bf16", "data": "synthetic",
            "config": {"workload": label,
                       "clips_per_gpu": n_local, "tokens_per_clip": round(tokens / (n_units_total * args.steps), 1),
                       "decode_row_steps_per_clip": round(row_steps / (n_units_total * args.steps), 1),
                       "stage_ms_per_step_rank0": {k: round(v, 2) for k, v in ms.items()}, "sharding": f"utterance-dp{world}",
                       "ranks_seen": len(ranks_seen), "ranks": ranks_seen,
                       # > 0: the group decodes as two halves, their attention kernels on this many CUs beside the
                       # other kernels on the rest (kernel times then overlap: they sum to more than the stage time)
                       "decode_attention_cus": 0 if args.workload == "mel" else model.decode_attention_cus(min(n_local, args.batch))},
            "roofline": {"bound": bound, "kernel": top["name"], "achieved": round(ach, 2), "peak": peak, "unit": unit,
                         "frac": round(ach / peak, 4), "peak_measured_copy": copy_gbs, "peak_measured_read": read_gbs, "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(avg_ms, 5),
                         "work_per_launch": per_launch_work, "share_of_device_time": round(top["ms_est"] / sum(p["ms_est"] for p in prof), 3),
                         # the two halves' attention launches run on two streams and overlap by a fifth of their time; a launch
                         # whose duration is sampled is made to run alone, so avg_launch_ms is the kernel's own time (what
                         # rocprofv3 reports per launch when both halves share one stream, Q3ASR_ATT_STREAMS=1)
                         "sampling": "every %d-th launch, timed with HIP events on its launch stream; a sampled decode-attention launch runs "
                                     "without the other half's launch beside it" % args.profile_stride},
            "kernels": [{"name": p["name"], "launches_per_step": p["launches"] / args.steps, "ms_per_step": round(p["ms_est"] / args.steps, 2),
                         "rate": round(p["work_timed"] / max(p["ms_timed"], 1e-9) / 1e9, 2)} for p in prof[:16]],
        }
        if not args.leak_model:
            model.close()
        del dev_audio
        if world == 1 and not args.no_cpu_baseline and args.workload == "transcribe" and args.weights == "bf16":
            line["cpu_baseline"] = cpu_baseline(cfg, ck, clips[0], args.cpu_decode_steps)
        else:
            line["cpu_baseline"] = None
        emit(line)
    if args.leak_model:
        model._h = None    # the handle is dropped without q3asr_destroy
    model.close()          # streams and events are released before the process (and any profiler around it) winds down
    if group:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
