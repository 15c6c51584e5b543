"""C ABI on a box without a GPU: the library loads, exports every symbol include/rgx.h declares, compiles patterns,
round-trips blobs, and REFUSES to compute (no CPU fallback exists)."""
import ctypes as C
import os
import re

import pytest

from regengo_amd import _capi, field_names

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_binding_agree(built):
    hdr = open(os.path.join(ROOT, "include", "rgx.h")).read()
    declared = set(re.findall(r"\b(rgx_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"rgx_status"}
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    lib = _capi.lib()
    for s in declared:
        assert hasattr(lib, s), s


def test_compile_info_blob(built):
    lib = _capi.lib()
    h = C.c_void_p()
    assert lib.rgx_compile(rb"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})", 0, C.byref(h)) == 0
    info = _capi.Info()
    assert lib.rgx_program_info(h, C.byref(info)) == 0
    assert (info.ncap, info.min_match_len, info.max_match_len, info.default_max_leftover, info.min_buffer_size) == (8, 10, 10, 1024, 65536)
    assert info.n_inst == 18 and info.fixed_captures == 1 and info.anchored == 0
    n = lib.rgx_program_blob_size(h)
    assert n > 0
    buf = C.create_string_buffer(n)
    assert lib.rgx_program_blob_write(h, buf, n) == n
    h2 = C.c_void_p()
    assert lib.rgx_program_from_blob(buf, n, C.byref(h2)) == 0
    i2 = _capi.Info()
    lib.rgx_program_info(h2, C.byref(i2))
    assert bytes(info) == bytes(i2)
    assert lib.rgx_program_from_blob(buf, n // 2, C.byref(C.c_void_p())) == -9
    nn = lib.rgx_program_capture_names(h, None, 0)
    nb = C.create_string_buffer(nn)
    lib.rgx_program_capture_names(h, nb, nn)
    assert nb.raw.split(b"\0")[:4] == [b"", b"year", b"month", b"day"]
    lib.rgx_program_destroy(h)
    lib.rgx_program_destroy(h2)


def test_errors(built):
    lib = _capi.lib()
    h = C.c_void_p()
    assert lib.rgx_compile(b"(unclosed", 0, C.byref(h)) == _capi.RGX_E_SYNTAX
    assert lib.rgx_compile(b"a**", 0, C.byref(h)) == _capi.RGX_E_SYNTAX
    assert lib.rgx_compile(rb"\p{Hani}+", 0, C.byref(h)) == _capi.RGX_E_SYNTAX      # ISO code, not a key of unicode.Scripts
    assert lib.rgx_compile(rb"\p{Garay}+", 0, C.byref(h)) == _capi.RGX_E_SYNTAX      # a Unicode 16.0 script: the tables are 15.0, like Go 1.24's (rgx_info.unicode_version)
    assert lib.rgx_compile(rb"\p{Han}+", 0, C.byref(h)) == 0
    lib.rgx_program_destroy(h)
    assert lib.rgx_compile(rb"\p{Greek}+", 0, C.byref(h)) == 0
    assert b"syntax" in lib.rgx_status_str(-2)


def test_field_names_follow_reference_rules():
    # captures.go:63-76: UpperFirst(name) | Group<i>; collisions get the group number appended
    assert field_names(["", "year", "", "match", "year"]) == ["Match", "Year", "Group2", "Match3", "Year4"]


def test_no_cpu_fallback(built):
    """Without a usable GPU every compute path must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _capi.lib()
    assert lib.rgx_device_count() == 0
    h = C.c_void_p()
    assert lib.rgx_compile(rb"(\d+)", 0, C.byref(h)) == 0
    assert lib.rgx_program_to_device(h, 0) == _capi.RGX_E_NO_DEVICE
    ctx = C.c_void_p()
    assert lib.rgx_stream_ctx_create(h, C.byref(ctx)) == _capi.RGX_E_NO_DEVICE
    from regengo_amd import Compiled
    with pytest.raises(_capi.RgxError):
        Compiled(r"(\d+)").FindAllBytes(b"123")


def test_stream_config_resolve(built, kats):
    lib = _capi.lib()
    h = C.c_void_p()
    assert lib.rgx_compile(rb"(\d{4}-\d{2}-\d{2})", 0, C.byref(h)) == 0
    out = _capi.StreamConfig()
    assert lib.rgx_stream_config_resolve(h, C.byref(_capi.StreamConfig(0, 0)), C.byref(out)) == 0
    assert (out.buffer_size, out.max_leftover) == (65536, 1024)
    assert lib.rgx_stream_config_resolve(h, C.byref(_capi.StreamConfig(100, 0)), C.byref(out)) == _capi.RGX_E_BUFFER_TOO_SMALL
    assert lib.rgx_stream_config_resolve(h, C.byref(_capi.StreamConfig(1 << 20, -1)), C.byref(out)) == 0
    assert (out.buffer_size, out.max_leftover) == (1 << 20, -1)


def test_corrupted_blob_is_refused(built):
    """The blob carries a checksum and every index in it is range-checked on load (a truncated, padded or hand-assembled blob
    must not reach the table walkers): RGX_E_BAD_BLOB, never a program."""
    lib = _capi.lib()
    h = C.c_void_p()
    assert lib.rgx_compile(rb"(?P<user>\w+)@(?P<domain>\w+)", 0, C.byref(h)) == 0
    n = lib.rgx_program_blob_size(h)
    buf = C.create_string_buffer(n)
    assert lib.rgx_program_blob_write(h, buf, n) == n
    good = bytes(buf.raw)
    lib.rgx_program_destroy(h)
    h2 = C.c_void_p()
    assert lib.rgx_program_from_blob(good, len(good), C.byref(h2)) == 0
    lib.rgx_program_destroy(h2)
    import random
    rng = random.Random(1)
    for trial in range(200):
        bad = bytearray(good)
        k = rng.randrange(len(bad))
        bad[k] ^= 1 << rng.randrange(8)
        assert lib.rgx_program_from_blob(bytes(bad), len(bad), C.byref(h2)) == -9, (trial, k)
    for cut in (0, 7, 100, len(good) - 1):
        assert lib.rgx_program_from_blob(good[:cut], cut, C.byref(h2)) == -9
    assert lib.rgx_program_from_blob(good + b"\0" * 8, len(good) + 8, C.byref(h2)) == -9
