"""On-disk formats either side of the synthesis path - host code, no GPU involved.

* WAV in:  what `Megatts.forward` does with librosa (reference models/megatts2.py:333-336:
  `librosa.load(wav, sr=16000)` -> mono float32, then `librosa.util.normalize`), without librosa:
  RIFF PCM 8/16/24/32-bit and IEEE float32, channels averaged, peak-normalised.  A file at another
  sample rate is rejected unless `resample=True` (polyphase; librosa's soxr resampler is not
  reproduced - parity unpinned for resampled input).
* WAV out: `torchaudio.save('test.wav', audio, 16000)` (models/megatts2.py:375) writes 32-bit float PCM
  for a float32 tensor; `write_wav` does the same by default, or 16-bit PCM with clipping.
* Packed weights: the three Lightning checkpoints are pickles read through `torch.load`
  (models/megatts2.py:111,192,287).  `save_packed` / `load_packed` keep the same state-dict names in
  one flat file (JSON index + 64-byte aligned raw f32), memory-mapped on load so that
  `mt2_model_load_tensor` reads straight from the page cache.
"""
from __future__ import annotations

import json
import struct
from typing import Dict, Tuple

import numpy as np

MAGIC = b"MT2PACK1"


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """-> (samples float32 [L] in [-1, 1), mono = mean of channels; sample_rate)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE and size >= 26:          # WAVE_FORMAT_EXTENSIBLE: real tag in the sub-format GUID
                fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    tag, ch, sr, _, _, bits = fmt
    if tag == 3 and bits == 32:
        x = np.frombuffer(pcm, "<f4").astype(np.float32)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(pcm, "<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = np.frombuffer(pcm, "<i4").astype(np.float32) / 2147483648.0
    elif tag == 1 and bits == 24:
        b = np.frombuffer(pcm, np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif tag == 1 and bits == 8:
        x = (np.frombuffer(pcm, np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"{path}: unsupported WAV encoding (format tag {tag}, {bits} bits)")
    x = x[:x.size // ch * ch].reshape(-1, ch).mean(axis=1).astype(np.float32) if ch > 1 else x
    return x, int(sr)


def normalize(y: np.ndarray) -> np.ndarray:
    """librosa.util.normalize(y) with its defaults: divide by max |y| (left unchanged when that is tiny)."""
    peak = float(np.max(np.abs(y))) if y.size else 0.0
    return y if peak < np.finfo(np.float32).tiny else (y / peak).astype(np.float32)


def load_audio(path: str, sr: int = 16000, resample: bool = False) -> np.ndarray:
    """librosa.load(path, sr=sr)[0] followed by librosa.util.normalize (models/megatts2.py:335-336)."""
    y, file_sr = read_wav(path)
    if file_sr != sr:
        if not resample:
            raise ValueError(f"{path}: {file_sr} Hz, expected {sr} Hz (pass resample=True for a polyphase resample)")
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(sr, file_sr)
        y = resample_poly(y, sr // g, file_sr // g).astype(np.float32)
    return normalize(y)


def write_wav(path: str, samples, sample_rate: int = 16000, encoding: str = "PCM_F") -> None:
    """samples: [L] or [channels, L] float array in [-1, 1].  encoding "PCM_F" (32-bit float, what
    torchaudio.save writes for float32 input) or "PCM_S16" (clipped, round-to-nearest)."""
    x = np.asarray(samples.detach().cpu().numpy() if hasattr(samples, "detach") else samples, np.float32)
    if x.ndim == 1:
        x = x[None]
    ch, n = x.shape
    inter = np.ascontiguousarray(x.T)
    if encoding == "PCM_F":
        tag, bits, body = 3, 32, inter.astype("<f4").tobytes()
    elif encoding == "PCM_S16":
        tag, bits = 1, 16
        body = np.clip(np.rint(inter * 32768.0), -32768, 32767).astype("<i2").tobytes()
    else:
        raise ValueError("encoding must be PCM_F or PCM_S16")
    block = ch * bits // 8
    fmt = struct.pack("<HHIIHH", tag, ch, sample_rate, sample_rate * block, block, bits)
    extra = struct.pack("<4sII", b"fact", 4, n) if tag == 3 else b""
    riff = b"WAVE" + struct.pack("<4sI", b"fmt ", len(fmt)) + fmt + extra + struct.pack("<4sI", b"data", len(body)) + body
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(riff)) + riff + (b"\0" if len(body) & 1 else b""))


def save_packed(path: str, state_dict: Dict[str, np.ndarray]) -> None:
    """One flat file: MAGIC, u64 header length, JSON {name: [shape, offset]}, then 64-byte aligned raw f32."""
    index, off, arrs = {}, 0, []
    for k, v in state_dict.items():
        a = np.ascontiguousarray(np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, np.float32))
        index[k] = [list(a.shape), off]
        arrs.append(a)
        off += (a.nbytes + 63) & ~63
    head = json.dumps(index).encode()
    head += b" " * (-(len(MAGIC) + 8 + len(head)) % 64)
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<Q", len(head)) + head)
        for a in arrs:
            f.write(a.tobytes())
            f.write(b"\0" * (-a.nbytes % 64))


def load_packed(path: str) -> Dict[str, np.ndarray]:
    """-> {name: float32 array} backed by one read-only memory map of the file."""
    with open(path, "rb") as f:
        if f.read(len(MAGIC)) != MAGIC:
            raise ValueError(f"{path}: not a packed weight file")
        hlen = struct.unpack("<Q", f.read(8))[0]
        index = json.loads(f.read(hlen).decode())
    base = len(MAGIC) + 8 + hlen
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    out = {}
    for k, (shape, off) in index.items():
        n = int(np.prod(shape)) if shape else 1
        out[k] = np.frombuffer(mm, np.float32, n, base + off).reshape(shape)
    return out
