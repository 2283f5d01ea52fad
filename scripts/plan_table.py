#!/usr/bin/env python3
"""Measures every (tile, split-K) candidate of the four decode projections per row count on the GPU
(Q3ASR_DECODE_AUTOTUNE=1 Q3ASR_VERBOSE=2) and prints, next to the fastest candidate, what the library's fixed
rule picks and how far its time is from the fastest.  Used to derive and re-check `decode_plan_rule`.

  python scripts/plan_table.py --preset 1.7b --weights bf16 > gpurun_out/plan_1.7b_bf16.json
"""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "qwen3-asr-swift_amd"))
import numpy as np
from q3asr import Qwen3ASRModel, make_config, synth
cfg = make_config(%(preset)r, max_batch=4096, weight_format=%(wf)d)
ck = synth.make_checkpoint(cfg, seed=0)
m = Qwen3ASRModel.from_checkpoint(cfg, ck)
clip = synth.synth_audio(16000, 1)
for rows in %(rows)r:
    m.transcribe_batch([clip] * rows, 16000, max_tokens=3, stop_rules=False)
m.close()
"""


def run(preset, wf, rows, env_extra):
    env = dict(os.environ, Q3ASR_VERBOSE="2", Q3ASR_DECODE_GEMV="0", **env_extra)
    code = CHILD % dict(root=ROOT, preset=preset, wf=wf, rows=rows)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stderr[-4000:])
    return r.stderr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="1.7b")
    ap.add_argument("--weights", default="bf16", choices=["bf16", "q4"])
    ap.add_argument("--rows", default="8,16,32,64,128,256,512,1024,2048")
    ap.add_argument("--att-cus", type=int, default=0, help="two-half decode with the attention on this many compute units: the "
                    "projections are measured on the stream confined to the other ones, at half of each row count")
    args = ap.parse_args()
    rows = [int(x) for x in args.rows.split(",")]
    wf = 1 if args.weights == "q4" else 0
    split_env = {"Q3ASR_DECODE_ATT_CUS": str(args.att_cus)}
    measured = run(args.preset, wf, rows, {"Q3ASR_DECODE_AUTOTUNE": "1", **split_env})
    mm_cus = 0
    if args.att_cus:
        mo = re.search(r"cus (\d+) rows", measured)
        mm_cus = int(mo[1]) if mo else 0
        rows = [(r + 1) // 2 for r in rows]
    cand = {}
    for mo in re.finditer(r"rows (\d+) proj (\d) tile (\d) split (\d): ([\d.]+) us \(\+([\d.]+) us", measured):
        r, g, t, ks, us, extra = int(mo[1]), int(mo[2]), int(mo[3]), int(mo[4]), float(mo[5]), float(mo[6])
        cand.setdefault((r, g), {})[(t, ks)] = round(us + extra, 1)
    ruled = run(args.preset, wf, [2 * r for r in rows] if args.att_cus else rows, split_env)
    rule = {}
    for mo in re.finditer(r"decode plan for up to (\d+) rows on \d+ CUs(?: \(measured\))? \(tile/split.*?: qkv (\d)/(\d), o (\d)/(\d), gate_up (\d)/(\d), down (\d)/(\d)", ruled):
        rule[int(mo[1])] = [(int(mo[2 + 2 * g]), int(mo[3 + 2 * g])) for g in range(4)]
    names = ["qkv", "o", "gate_up", "down"]
    out = []
    for r in rows:
        for g in range(4):
            c = cand.get((r, g), {})
            if not c:
                continue
            best = min(c, key=c.get)
            pick = rule.get(r, [None] * 4)[g]
            out.append({"rows": r, "proj": names[g], "best": list(best), "best_us": c[best], "rule": list(pick) if pick else None,
                        "rule_us": c.get(pick), "all": {f"{t}/{k}": v for (t, k), v in sorted(c.items())}})
    print(json.dumps({"preset": args.preset, "weights": args.weights, "cus": mm_cus, "table": out}, indent=1))
    tot_best = sum(e["best_us"] for e in out)
    tot_rule = sum(e["rule_us"] or 0 for e in out)
    print(f"sum of best {tot_best:.0f} us, sum of rule picks {tot_rule:.0f} us", file=sys.stderr)


if __name__ == "__main__":
    main()
