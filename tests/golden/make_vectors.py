#!/usr/bin/env python3
"""Writes tests/golden/reference_vectors.json.

The reference ships no test suite; its known-answer material is the worked
examples of example.lisp and one base64 .xz file in README.org.  This script
holds those inputs and expected outputs as data (typed in from the cited
lines) plus catalogue check values, and re-verifies the decode vectors with
CPython's zlib/gzip/lzma before writing the fixture, so a typo cannot pin the
oracle to a wrong answer.  No Common Lisp implementation exists in the build
container, so the reference itself cannot be run.
"""
import base64
import gzip
import hashlib
import json
import lzma
import os
import zlib

Z = [120, 156, 99, 84, 100, 7, 0, 0, 79, 0, 42]
GZ = [31, 139, 8, 8, 0, 0, 0, 0, 0, 0, 102, 111, 111, 0, 99, 84, 100, 7, 0, 101, 51, 120, 236, 3, 0, 0, 0,
      31, 139, 8, 8, 0, 0, 0, 0, 0, 0, 98, 97, 114, 0, 99, 84, 100, 7, 0, 101, 51, 120, 236, 3, 0, 0, 0]
MULTI = [99, 84, 100, 7, 0, 99, 84, 123, 7, 0, 1, 2, 3]
ZDICT = [120, 249, 0, 79, 0, 42, 3, 34, 0, 0, 79, 0, 42]
XZ_LOL = "/Td6WFoAAAD/EtlBAgAhAQoAAABTxyq54AAAAAUJACX//AAAAgAAT4AAAAAFACfRR0AAAAABKAM7StLkBnKeegEAAAAAAFla"

vectors = [
    {"id": "example.lisp:6-7", "format": "zlib", "all_members": True, "input": Z, "output": [1, 33, 7],
     "header": {"window_size": 32768, "level": 2, "dictionary": None}},
    {"id": "example.lisp:12-19", "format": "gzip", "all_members": True, "input": GZ, "output": [1, 33, 7, 1, 33, 7],
     "header": {"textp": False, "extra_fields": None, "filename": "foo", "comment": None, "modification_time": 0,
                "extra_flags": 0, "operating_system": 0}},
    {"id": "example.lisp:24-26", "format": "deflate", "all_members": True, "input": Z, "start": 2, "end": 7,
     "output": [1, 33, 7], "header": None},
    {"id": "example.lisp:29-30", "format": "zlib", "all_members": True, "input": Z + [1, 2, 3],
     "error": "decompression-error"},
    {"id": "example.lisp:45-52/first", "format": "deflate", "all_members": False, "input": MULTI,
     "output": [1, 33, 7], "consumed": 5},
    {"id": "example.lisp:45-52/second", "format": "deflate", "all_members": False, "input": MULTI, "start": 5,
     "output": [1, 38, 238], "consumed": 5},
    {"id": "example.lisp:58-61", "format": "zlib", "all_members": False, "input": Z, "output": [1, 33, 7],
     "header": {"window_size": 32768, "level": 2, "dictionary": None}},
    {"id": "example.lisp:67-69", "format": "deflate", "all_members": True, "input": MULTI, "start": 5,
     "error": "decompression-error"},
    {"id": "example.lisp:72-74", "format": "zlib", "all_members": True, "input": ZDICT, "dictionary": [1, 33, 7],
     "output": [1, 33, 7], "header": {"window_size": 32768, "level": 3, "dictionary": 5177386}},
    {"id": "README.org:306-308", "format": "xz", "all_members": True,
     "input": list(base64.b64decode(XZ_LOL)), "output": list(b"LOL")},
]

# catalogue check values ("123456789") for the checksum parameter sets the reference names
checks = {
    "message": list(b"123456789"),
    "crc32": 0xCBF43926,            # CRC-32/ISO-HDLC, crc.lisp:19-110
    "crc64": 0x995DC9BBDF1939FA,    # CRC-64/XZ, crc.lisp:122-193
    "bzcrc": 0xFC891918,            # CRC-32/BZIP2, bzip2.lisp:13-39
    "adler32": 0x091E01DE,          # adler32.lisp
    "sha256": hashlib.sha256(b"123456789").hexdigest(),
}


def verify():
    assert zlib.decompress(bytes(Z)) == bytes([1, 33, 7])
    assert gzip.decompress(bytes(GZ)) == bytes([1, 33, 7] * 2)
    assert zlib.decompress(bytes(Z[2:7]), -15) == bytes([1, 33, 7])
    d = zlib.decompressobj(-15)
    assert d.decompress(bytes(MULTI)) == bytes([1, 33, 7]) and d.unused_data == bytes(MULTI[5:])
    d = zlib.decompressobj(-15)
    assert d.decompress(bytes(MULTI[5:])) == bytes([1, 38, 238]) and d.unused_data == bytes([1, 2, 3])
    d = zlib.decompressobj(15, bytes([1, 33, 7]))
    assert d.decompress(bytes(ZDICT)) == bytes([1, 33, 7])
    assert zlib.adler32(bytes([1, 33, 7])) == 5177386
    assert lzma.decompress(base64.b64decode(XZ_LOL)) == b"LOL"
    assert zlib.crc32(b"123456789") == checks["crc32"] and zlib.adler32(b"123456789") == checks["adler32"]


if __name__ == "__main__":
    verify()
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "reference_vectors.json"), "w") as f:
        json.dump({"vectors": vectors, "checks": checks}, f, indent=1)
    print("wrote reference_vectors.json:", len(vectors), "vectors")
