"""Byte-level BPE tokenizer of the harness (token ids <-> text).

Behaviour follows the reference's Qwen3Tokenizer (Tokenizer.swift): vocab.json is a {token: id}
map (:42-53), tokenizer_config.json's added_tokens_decoder overrides it (:77-100), merges.txt gives
the BPE ranks (:103-123); decode() drops chat specials spelled "<|...|>" (:138-141), keeps markers
such as "<asr_text>" (:144-147), and maps the byte alphabet back to UTF-8 over the whole stitched
string (:126-160, 202-221); encode() matches special tokens greedily (longest first, :248-266) and
runs GPT-2 pre-tokenisation + BPE on the text in between (:268-340).
"""
import json
import os

import regex

_PRETOKENIZE = regex.compile(r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+")   # Tokenizer.swift:31


def _byte_alphabet():
    """GPT-2 byte <-> printable unicode table (Tokenizer.swift:164-190)."""
    direct = list(range(33, 127)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    table, n = {}, 0
    for b in direct:
        table[b] = chr(b)
    for b in range(256):
        if b not in table:
            table[b] = chr(0x100 + n)
            n += 1
    return table


_B2U = _byte_alphabet()
_U2B = {u: b for b, u in _B2U.items()}


class TokenizerError(ValueError):
    pass


class Qwen3Tokenizer:
    def __init__(self):
        self.id_to_token, self.token_to_id, self.ranks = {}, {}, {}
        self._specials, self._cache = [], {}

    @classmethod
    def load(cls, vocab_path):
        t = cls()
        with open(vocab_path, encoding="utf-8") as f:
            vocab = json.load(f)
        if not isinstance(vocab, dict) or not all(isinstance(v, int) for v in vocab.values()):
            raise TokenizerError("Invalid tokenizer format: Expected {token: id} dictionary")
        for tok, i in vocab.items():
            t.id_to_token[i] = tok
            t.token_to_id[tok] = i
        d = os.path.dirname(os.path.abspath(vocab_path))
        cfg = os.path.join(d, "tokenizer_config.json")
        if os.path.exists(cfg):
            with open(cfg, encoding="utf-8") as f:
                added = (json.load(f) or {}).get("added_tokens_decoder", {})
            for key, info in added.items():
                if isinstance(info, dict) and isinstance(info.get("content"), str) and key.lstrip("-").isdigit():
                    t.id_to_token[int(key)] = info["content"]
                    t.token_to_id[info["content"]] = int(key)
        merges = os.path.join(d, "merges.txt")
        if not os.path.exists(merges):
            raise TokenizerError("Invalid tokenizer format: Missing merges.txt next to vocab.json")
        with open(merges, encoding="utf-8") as f:
            rank = 0
            for line in f.read().splitlines():
                line = line.strip()
                if not line or line.startswith("#"):
                    continue
                parts = line.split(" ", 1)
                if len(parts) == 2 and parts[1]:
                    t.ranks[(parts[0], parts[1].strip())] = rank
                    rank += 1
        t._specials = sorted((k for k in t.token_to_id if k.startswith("<") and k.endswith(">")), key=len, reverse=True)
        return t

    # ---- ids -> text ----
    def decode(self, tokens):
        parts = []
        for i in tokens:
            tok = self.id_to_token.get(int(i))
            if tok is None:
                continue
            if tok.startswith("<|") and tok.endswith("|>"):
                continue
            parts.append(tok)
        stitched = "".join(parts)
        raw = bytearray()
        for ch in stitched:
            raw += bytes([_U2B[ch]]) if ch in _U2B else ch.encode("utf-8")
        try:
            return raw.decode("utf-8").strip()
        except UnicodeDecodeError:
            return stitched.strip()                      # Tokenizer.swift:215-220 falls back to the mapped string

    @staticmethod
    def post_process(text):
        """Keep what follows the last "<asr_text>" (Qwen3ASR.swift:655-662)."""
        i = text.rfind("<asr_text>")
        return text if i < 0 else text[i + len("<asr_text>"):].strip()

    # ---- text -> ids ----
    def _bpe(self, word):
        if word in self._cache:
            return self._cache[word]
        parts = list(word)
        while len(parts) > 1:
            best, best_rank = None, None
            for a, b in zip(parts, parts[1:]):
                r = self.ranks.get((a, b))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (a, b), r
            if best is None:
                break
            merged, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and (parts[i], parts[i + 1]) == best:
                    merged.append(parts[i] + parts[i + 1])
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        self._cache[word] = parts
        return parts

    def _encode_plain(self, text):
        ids = []
        for piece in _PRETOKENIZE.findall(text):
            mapped = "".join(_B2U[b] for b in piece.encode("utf-8"))
            ids += [self.token_to_id[t] for t in self._bpe(mapped) if t in self.token_to_id]
        return ids

    def encode(self, text):
        ids, pos = [], 0
        while pos < len(text):
            hit = next((s for s in self._specials if text.startswith(s, pos)), None)
            if hit is not None:
                ids.append(self.token_to_id[hit])
                pos += len(hit)
                continue
            nxt = min((p for p in (text.find(s, pos) for s in self._specials) if p >= 0), default=len(text))
            ids += self._encode_plain(text[pos:nxt])
            pos = nxt
        return ids
