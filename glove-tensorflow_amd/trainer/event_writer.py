"""Scalar summaries as TensorBoard event files, written without TensorFlow.

The reference's Estimator leaves `events.out.tfevents.*` files in the job directory (training summaries every
100 steps: `loss`, `global_step/sec`, the `mf/global_bias` scalar of add_summary, src/models/model_utils.py:113-118)
and in `eval/` (the eval metrics).  Both are plain TFRecord streams of `tensorflow.Event` protocol buffers, so the
few fields needed here are encoded by hand:

    record  = uint64 length | masked crc32c(length) | payload | masked crc32c(payload)        (little endian)
    Event   = 1: wall_time (double)  2: step (int64)  3: file_version (string)  5: summary (message)
    Summary = 1: value (repeated message);  Value = 1: tag (string)  2: simple_value (float)  5: histo (message)
    HistogramProto = 1: min  2: max  3: num  4: sum  5: sum_squares (double)  6: bucket_limit  7: bucket (packed double)

Histograms (the reference's `summary.histogram` of the row and col biases, src/models/model_utils.py:116-117) use
TensorFlow's default bucket limits: +-1e-12 * 1.1^k up to 1e20, 0, and DBL_MAX at the ends; as TF's encoder does,
runs of empty buckets are merged into one entry.
"""
from __future__ import annotations

import os
import socket
import struct
import time

_CRC_TABLE = []
for _n in range(256):
    _c = _n
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1       # reflected Castagnoli polynomial
    _CRC_TABLE.append(_c)


def crc32c(data: bytes) -> int:
    crc = 0xFFFFFFFF
    for byte in data:
        crc = _CRC_TABLE[(crc ^ byte) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _masked(data: bytes) -> bytes:
    crc = crc32c(data)
    return struct.pack("<I", (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF)


def _varint(n: int) -> bytes:
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        out.append((n & 0x7F) | (0x80 if n > 0x7F else 0))
        n >>= 7
        if not n:
            return bytes(out)


def _field(number: int, wire: int, payload: bytes) -> bytes:
    head = _varint((number << 3) | wire)
    return head + (_varint(len(payload)) if wire == 2 else b"") + payload


_LIMITS = None


def default_bucket_limits() -> list:
    """tensorflow/core/lib/histogram/histogram.cc InitDefaultBucketsInner: 1e-12 * 1.1^k below 1e20 and DBL_MAX,
    mirrored to the negative side around 0."""
    global _LIMITS
    if _LIMITS is None:
        pos, v = [], 1e-12
        while v < 1e20:
            pos.append(v)
            v *= 1.1
        pos.append(1.7976931348623157e308)
        _LIMITS = [-x for x in reversed(pos)] + [0.0] + pos
    return _LIMITS


_LIMITS_NP = None


def _limits_np():
    global _LIMITS_NP
    if _LIMITS_NP is None:
        import numpy as np
        _LIMITS_NP = np.asarray(default_bucket_limits(), dtype=np.float64)
    return _LIMITS_NP


def histogram_of(values) -> dict:
    """HistogramProto fields of a 1-D tensor or array (any device): bucket i counts limit[i-1] <= x < limit[i]
    (TF: upper_bound over the limits).  Host data goes through NumPy, device tensors stay on their device."""
    import numpy as np
    import torch
    limits = default_bucket_limits()
    if isinstance(values, torch.Tensor) and values.is_cuda:
        x = values.detach().double().flatten()
        lim = torch.tensor(limits, dtype=torch.float64, device=x.device)
        counts = torch.bincount(torch.bucketize(x, lim, right=True).clamp_(max=len(limits) - 1), minlength=len(limits)).tolist()
        n = x.numel()
        stats = [float(v) for v in torch.stack([x.min(), x.max(), x.sum(), (x * x).sum()]).tolist()] if n else [0.0] * 4
    else:
        x = np.asarray(values.detach().numpy() if isinstance(values, torch.Tensor) else values, dtype=np.float64).ravel()
        counts = np.bincount(np.minimum(np.searchsorted(_limits_np(), x, side="right"), len(limits) - 1), minlength=len(limits))
        n = x.size
        stats = [float(x.min()), float(x.max()), float(x.sum()), float((x * x).sum())] if n else [0.0] * 4
    # Histogram::EncodeToProto: a run of empty buckets becomes one entry (its last limit, count 0)
    c = np.asarray(counts, dtype=np.int64)
    empty = c <= 0
    keep = ~empty | np.append(~empty[1:], True)            # every non-empty bucket, and the LAST bucket of every empty run
    lim_np = _limits_np()
    out_limits, out_counts = lim_np[keep].tolist(), c[keep].astype(np.float64).tolist()
    return {"min": stats[0], "max": stats[1], "num": float(n), "sum": stats[2], "sum_squares": stats[3],
            "bucket_limit": out_limits, "bucket": out_counts}


def _histo(h: dict) -> bytes:
    d = lambda v: struct.pack("<d", float(v))
    packed = lambda vs: b"".join(d(v) for v in vs)
    return (_field(1, 1, d(h["min"])) + _field(2, 1, d(h["max"])) + _field(3, 1, d(h["num"])) + _field(4, 1, d(h["sum"])) +
            _field(5, 1, d(h["sum_squares"])) + _field(6, 2, packed(h["bucket_limit"])) + _field(7, 2, packed(h["bucket"])))


def encode_event(wall_time: float, step: int = 0, scalars: dict | None = None, file_version: str | None = None,
                 histograms: dict | None = None) -> bytes:
    event = _field(1, 1, struct.pack("<d", wall_time)) + _field(2, 0, _varint(step))
    if file_version is not None:
        event += _field(3, 2, file_version.encode())
    values = b""
    for tag, v in (scalars or {}).items():
        values += _field(1, 2, _field(1, 2, tag.encode()) + _field(2, 5, struct.pack("<f", float(v))))
    for tag, h in (histograms or {}).items():
        values += _field(1, 2, _field(1, 2, tag.encode()) + _field(5, 2, _histo(h)))
    if values:
        event += _field(5, 2, values)
    return event


class EventWriter:
    """Appends scalar summaries to one `events.out.tfevents.<time>.<host>` file in `logdir`."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "events.out.tfevents.%010d.%s" % (int(time.time()), socket.gethostname()))
        self._write(encode_event(time.time(), file_version="brain.Event:2"))

    def _write(self, payload: bytes):
        header = struct.pack("<Q", len(payload))
        with open(self.path, "ab") as f:
            f.write(header + _masked(header) + payload + _masked(payload))

    def scalars(self, step: int, values: dict, histograms: dict | None = None):
        self._write(encode_event(time.time(), step, {k: v for k, v in values.items() if isinstance(v, (int, float))},
                                 histograms=histograms))


def read_events(path):
    """Decodes a file written above (checks every checksum): yields (wall_time, step, {tag: value}); a histogram's
    value is the dict of its HistogramProto fields."""
    def parse(buf):
        pos, out = 0, []
        while pos < len(buf):
            key, shift = 0, 0
            while True:
                b = buf[pos]; pos += 1
                key |= (b & 0x7F) << shift; shift += 7
                if not b & 0x80:
                    break
            number, wire = key >> 3, key & 7
            if wire == 0:
                val, shift = 0, 0
                while True:
                    b = buf[pos]; pos += 1
                    val |= (b & 0x7F) << shift; shift += 7
                    if not b & 0x80:
                        break
            elif wire == 1:
                val = buf[pos:pos + 8]; pos += 8
            elif wire == 5:
                val = buf[pos:pos + 4]; pos += 4
            else:
                n, shift = 0, 0
                while True:
                    b = buf[pos]; pos += 1
                    n |= (b & 0x7F) << shift; shift += 7
                    if not b & 0x80:
                        break
                val = buf[pos:pos + n]; pos += n
            out.append((number, val))
        return out

    data = open(path, "rb").read()
    pos = 0
    while pos < len(data):
        header = data[pos:pos + 8]
        (n,) = struct.unpack("<Q", header)
        if data[pos + 8:pos + 12] != _masked(header):
            raise ValueError("%s: corrupt record length at byte %d" % (path, pos))
        payload = data[pos + 12:pos + 12 + n]
        if data[pos + 12 + n:pos + 16 + n] != _masked(payload):
            raise ValueError("%s: corrupt record payload at byte %d" % (path, pos))
        pos += 16 + n
        wall, step, scalars = 0.0, 0, {}
        for number, val in parse(payload):
            if number == 1:
                (wall,) = struct.unpack("<d", val)
            elif number == 2:
                step = val
            elif number == 5:
                for _, value in parse(val):
                    fields = dict(parse(value))
                    if 5 in fields:
                        h = dict(parse(fields[5]))
                        un = lambda b: list(struct.unpack("<%dd" % (len(b) // 8), b))
                        names = {1: "min", 2: "max", 3: "num", 4: "sum", 5: "sum_squares"}
                        scalars[fields[1].decode()] = {**{names[k]: struct.unpack("<d", h[k])[0] for k in names},
                                                       "bucket_limit": un(h.get(6, b"")), "bucket": un(h.get(7, b""))}
                    else:
                        scalars[fields[1].decode()] = struct.unpack("<f", fields[2])[0]
        yield wall, step, scalars
