"""Scalar summaries as TensorBoard event files, written without TensorFlow.

The reference's Estimator leaves `events.out.tfevents.*` files in the job directory (training summaries every
100 steps: `loss`, `global_step/sec`, the `mf/global_bias` scalar of add_summary, src/models/model_utils.py:113-118)
and in `eval/` (the eval metrics).  Both are plain TFRecord streams of `tensorflow.Event` protocol buffers, so the
few fields needed here are encoded by hand:

    record  = uint64 length | masked crc32c(length) | payload | masked crc32c(payload)        (little endian)
    Event   = 1: wall_time (double)  2: step (int64)  3: file_version (string)  5: summary (message)
    Summary = 1: value (repeated message);  Value = 1: tag (string)  2: simple_value (float)
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


def encode_event(wall_time: float, step: int = 0, scalars: dict | None = None, file_version: str | None = None) -> bytes:
    event = _field(1, 1, struct.pack("<d", wall_time)) + _field(2, 0, _varint(step))
    if file_version is not None:
        event += _field(3, 2, file_version.encode())
    if scalars:
        values = b"".join(_field(1, 2, _field(1, 2, tag.encode()) + _field(2, 5, struct.pack("<f", float(v))))
                          for tag, v in scalars.items())
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

    def scalars(self, step: int, values: dict):
        self._write(encode_event(time.time(), step, {k: v for k, v in values.items() if isinstance(v, (int, float))}))


def read_events(path):
    """Decodes a file written above (checks every checksum): yields (wall_time, step, {tag: value})."""
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
                    scalars[fields[1].decode()] = struct.unpack("<f", fields[2])[0]
        yield wall, step, scalars
