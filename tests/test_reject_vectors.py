"""CPU: the oracle (the C restatement of the reference) gives every hand-built vector of tests/vectors_reject.py
the status the reference's rules call for (README.org:161-354 and the `die` sites cited in
include/semz_status.h), the expected message argument where the vector names one, and the expected bytes for
the accepted oddities.  Together with the list of statuses no input can reach this accounts for every code of
the enum.  Accepted vectors are cross-checked with the system decoders where those accept them too."""
import bz2
import lzma
import os
import re
import zlib

import pytest

import reject_runner
import vectors_reject as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("i", range(len(V.VECTORS)), ids=lambda i: V.VECTORS[i]["name"][:60])
def test_oracle_gives_the_expected_status(i):
    v = V.VECTORS[i]
    status, detail, out = reject_runner.oracle_run(v)
    assert status == v["status"], (v["name"], status)
    if v["detail"] is not None and status != 0:
        assert detail == v["detail"], (v["name"], detail)
    if v["out"] is not None:
        assert out == v["out"], v["name"]


def test_every_status_is_accounted_for():
    text = open(os.path.join(ROOT, "include", "semz_status.h")).read()
    enum = {int(n) for n in re.findall(r"SEMZ_\w+ = (\d+)", text)}
    seen = {v["status"] for v in V.VECTORS}
    assert enum - seen == set(V.UNREACHABLE) | set(V.ENGINE_LEVEL), sorted(enum - seen - set(V.UNREACHABLE) - set(V.ENGINE_LEVEL))
    assert len(V.VECTORS) >= 190


def test_accepted_vectors_against_the_system_decoders():
    """What the reference accepts and the system decoders accept too must decode to the same bytes."""
    checked = 0
    for v in V.VECTORS:
        if v["status"] != 0 or v["route"] != "stream" or v["out"] is None:
            continue
        try:
            if v["fmt"] == V.FMT_DEFLATE:
                got = zlib.decompressobj(-15).decompress(v["data"])
            elif v["fmt"] == V.FMT_ZLIB:
                got = zlib.decompress(v["data"])
            elif v["fmt"] == V.FMT_GZIP:
                d, got, rest = zlib.decompressobj(31), b"", v["data"]
                while rest:
                    got += d.decompress(rest)
                    rest, d = d.unused_data, zlib.decompressobj(31)
            elif v["fmt"] == V.FMT_BZIP2:
                got = bz2.decompress(v["data"])            # includes the randomised blocks: libbz2 still reads them
            elif v["fmt"] == V.FMT_XZ:
                got = lzma.decompress(v["data"])
            elif v["fmt"] == V.FMT_LZMA:
                got = lzma.decompress(v["data"], lzma.FORMAT_ALONE)
            elif v["fmt"] == V.FMT_RAW_LZMA2:
                got = lzma.decompress(v["data"], lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "dict_size": max(v["opt"], 4096)}])
            else:
                continue
        except (zlib.error, OSError, lzma.LZMAError, ValueError):
            continue                                       # e.g. zlib rejects trees that define codes 286 / 287
        if v["all"]:
            if v["fmt"] == V.FMT_XZ and b"\0\0\0\0\xfd7zXZ" in v["data"]:
                assert v["out"].startswith(got) and got      # CPython's lzma.decompress stops at stream padding
            else:
                assert got == v["out"], v["name"]
            checked += 1
    assert checked >= 25
    names = {v["name"] for v in V.VECTORS if v["status"] == 0}
    assert any("randomised" in n for n in names)
