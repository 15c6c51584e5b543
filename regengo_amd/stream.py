"""Mirror of the reference's runtime `stream` package types (stream/stream.go:21-134), unchanged in meaning."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any


@dataclass
class Config:  # stream.Config, stream.go:21-39
    BufferSize: int = 0   # 0 = 64 KiB default
    MaxLeftover: int = 0  # 0 = pattern default, -1 = unlimited


def DefaultConfig() -> Config:  # stream.go:41-49
    return Config(64 * 1024, 0)


@dataclass
class Match:  # stream.Match[T], stream.go:66-79
    Result: Any
    StreamOffset: int
    ChunkIndex: int


class ErrBufferTooSmall(Exception):  # stream.go:85-94
    def __init__(self, requested: int, minimum: int):
        super().__init__("stream: buffer size too small")
        self.Requested = requested
        self.Minimum = minimum
