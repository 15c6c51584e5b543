"""regengo_amd: MI355X-native (HIP/CDNA4) backend for regengo's MatchBytes / FindAllBytes / FindReader hot path.

Layout: csrc/ (front-end, table compiler, HIP kernels, C ABI -> lib/librgx_hip.so), api.py (host mirror of the
generated Compiled<Name> API), stream.py (stream.Config/Match), dist.py (multi-GPU sharding), synth.py (inputs).
"""
from .api import BytesResult, Compiled, Package, field_names  # noqa: F401
from .stream import Config, DefaultConfig, ErrBufferTooSmall, Match  # noqa: F401
