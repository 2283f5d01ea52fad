"""Minimal safetensors reader for the harness (the reference uses MLX.loadArrays,
WeightLoading.swift:62-64, and merges every *.safetensors file of the model directory,
:71-88).  Only the dtypes the checkpoints of this path hold are accepted: BF16, U32, F32.
Nothing from the file is executed: an 8-byte length, a JSON header, raw little-endian data."""
import json
import mmap
import os
import struct

import numpy as np

_DT = {"BF16": ("bf16", np.uint16), "U32": ("u32", np.uint32), "F32": ("f32", np.float32)}


def load_file(path):
    out = {}
    with open(path, "rb") as f:
        size = os.fstat(f.fileno()).st_size
        if size < 8:
            raise ValueError(f"{path}: not a safetensors file")
        (hlen,) = struct.unpack("<Q", f.read(8))
        if hlen > size - 8 or hlen > (1 << 27):
            raise ValueError(f"{path}: header length {hlen} out of bounds")
        header = json.loads(f.read(hlen).decode("utf-8"))
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    base = 8 + hlen
    for name, info in header.items():
        if name == "__metadata__":
            continue
        if info["dtype"] not in _DT:
            raise ValueError(f"{path}: tensor {name} has unsupported dtype {info['dtype']}")
        tag, npdt = _DT[info["dtype"]]
        b0, b1 = info["data_offsets"]
        n = int(np.prod(info["shape"])) if info["shape"] else 1
        if b0 < 0 or b1 < b0 or base + b1 > size or b1 - b0 != n * np.dtype(npdt).itemsize:
            raise ValueError(f"{path}: tensor {name} has inconsistent offsets")
        out[name] = (np.frombuffer(mm, dtype=npdt, count=n, offset=base + b0).reshape(info["shape"]), tag)
    return out


def load_dir(directory):
    files = sorted(p for p in os.listdir(directory) if p.endswith(".safetensors"))
    if not files:
        raise FileNotFoundError(f"No safetensors files found in: {directory}")   # WeightLoadingError.noWeightsFound
    out = {}
    for p in files:
        out.update(load_file(os.path.join(directory, p)))
    return out


def save_file(path, tensors):
    """Writer used by tests to mint small checkpoints.  tensors: name -> (array, tag)."""
    rev = {v[0]: k for k, v in _DT.items()}
    header, blobs, off = {}, [], 0
    for name, (arr, tag) in tensors.items():
        raw = np.ascontiguousarray(arr).tobytes()
        header[name] = {"dtype": rev[tag], "shape": list(arr.shape), "data_offsets": [off, off + len(raw)]}
        blobs.append(raw)
        off += len(raw)
    hj = json.dumps(header).encode("utf-8")
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for b in blobs:
            f.write(b)
