"""CPU: lisp/gpu.lisp, the CFFI binding of the reference-side plugin API (common.lisp:262-289), cannot be
executed here (the image holds no Common Lisp implementation), so it is checked against its two sources of
truth as text:

* every symbol of the reference it names (`semz.decompress::x`, `semz.decompress:x`) is defined, and where it
  is named with one colon also exported, by the files under /root/reference (skipped when that tree is absent,
  e.g. on the GPU box);
* every foreign function, struct and constant it binds agrees with include/semz_gpu.h and
  include/semz_status.h: names, argument counts, field order and types, format codes, status codes;
* the file is readable: balanced parentheses outside strings and comments, all forms inside one package.
"""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = os.path.join(ROOT, "lisp", "gpu.lisp")
ASD = os.path.join(ROOT, "lisp", "semz.decompress.gpu.asd")
REF = "/root/reference"


def strip_lisp(text, keep_strings=False):
    """Source text without comments and with string contents blanked (so that parentheses can be counted)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == ";":
            while i < n and text[i] != "\n":
                i += 1
        elif c == "#" and text[i:i + 2] == "#|":
            j = text.index("|#", i)
            i = j + 2
        elif c == "#" and text[i:i + 2] == "#\\":
            out.append("#\\x")
            i += 3
            while i < n and (text[i].isalnum() or text[i] == "-"):
                i += 1
        elif c == '"':
            j = i + 1
            while text[j] != '"':
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1] if keep_strings else '""')
            i = j + 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def test_files_are_readable():
    for path in (GLUE, ASD):
        code = strip_lisp(open(path).read())
        depth = 0
        for ch in code:
            depth += ch == "("
            depth -= ch == ")"
            assert depth >= 0, path
        assert depth == 0, path
    asd = open(ASD).read()
    assert '"semz.decompress"' in asd and '"cffi"' in asd and '(:file "gpu")' in asd


def reference_definitions():
    defined, exported = set(), set()
    for path in glob.glob(os.path.join(REF, "*.lisp")):
        text = strip_lisp(open(path).read())
        for m in re.finditer(r"\((?:defun|defmacro|defgeneric|defclass|define-condition|deftype|defparameter|defvar|"
                             r"define-constant|defmethod)\s+\(?(?:setf\s+)?([^\s()]+)", text):
            defined.add(m.group(1).lower())
        for m in re.finditer(r"\(define-fast(?:-inline)?-function\s+\(?([^\s()]+)", text):
            defined.add(m.group(1).lower())
        # structures: constructor, predicate and slot accessors (slots of an :include'd structure are inherited)
        for m in re.finditer(r"\(defstruct\s+(\(?)([^\s()]+)", text):
            name = m.group(2).lower()
            body_start = m.end()
            depth, j = 1 + (1 if m.group(1) else 0), body_start
            if m.group(1):                                   # skip the options list
                d = 1
                while d:
                    d += text[j] == "("
                    d -= text[j] == ")"
                    j += 1
                depth = 1
            body_from = j
            while depth:
                depth += text[j] == "("
                depth -= text[j] == ")"
                j += 1
            body = text[body_from:j - 1]
            slots = re.findall(r"^\s*\(([a-z0-9-]+)\s", body, flags=re.M) + re.findall(r"^\s*([a-z0-9-]+)\s*$", body, flags=re.M)
            defined.update({name, f"make-{name}", f"{name}-p", f"copy-{name}"})
            defined.update(f"{name}-{s}" for s in slots)
            inc = re.search(r":include\s+([^\s()]+)", text[m.start():body_from])
            if inc:
                defined.add(("include", name, inc.group(1).lower()))
        for m in re.finditer(r"\(:export((?:\s+#:[^\s()]+)+)", text):
            exported.update(s.lower() for s in re.findall(r"#:([^\s()]+)", m.group(1)))
    # inherited accessors of :include'd structures
    for item in [d for d in defined if isinstance(d, tuple)]:
        _, child, parent = item
        for d in list(defined):
            if isinstance(d, str) and d.startswith(parent + "-"):
                defined.add(child + d[len(parent):])
    return {d for d in defined if isinstance(d, str)}, exported


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_reference_symbols_exist():
    defined, exported = reference_definitions()
    code = strip_lisp(open(GLUE).read())
    internal = set(re.findall(r"semz\.decompress::([^\s()']+)", code))
    external = set(re.findall(r"semz\.decompress:([^\s():']+)", code)) - {""}
    assert {"byte-source->decompression-state", "next-decompressed-chunk", "make-reset-state"} <= internal
    for sym in sorted(internal | external):
        assert sym.lower() in defined, f"semz.decompress has no definition of {sym}"
    for sym in sorted(external):
        assert sym.lower() in exported, f"{sym} is not exported by semz.decompress (common.lisp:3-19)"
    # the protocol functions are generic, the public entry points are not: hooks must hang on the former
    common = strip_lisp(open(os.path.join(REF, "common.lisp")).read())
    for g in ("byte-source->decompression-state", "next-decompressed-chunk", "make-reset-state"):
        assert re.search(r"\(defgeneric\s+" + re.escape(g) + r"\s", common)
    for m in re.finditer(r"\(defmethod\s+([^\s()]+)", code):
        assert m.group(1).startswith("semz.decompress::") and m.group(1).split("::")[1] in (
            "byte-source->decompression-state", "next-decompressed-chunk", "make-reset-state"), m.group(1)
    # every format keyword the dispatch covers is one the reference registers (pushnew ... *known-formats*)
    known = set()
    for path in glob.glob(os.path.join(REF, "*.lisp")):
        known.update(re.findall(r"\(pushnew\s+(:[a-z0-9-]+)\s+\*known-formats\*\)", open(path).read()))
    hooked = set(re.findall(r":[a-z0-9-]+", re.search(r"\(define-dispatch ((?::[a-z0-9-]+\s*)+)\)", code).group(1)))
    assert hooked == known and len(known) == 9


def header_text():
    text = open(os.path.join(ROOT, "include", "semz_gpu.h")).read()
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


C_TYPES = {":int32": "int32_t", ":uint32": "uint32_t", ":uint64": "uint64_t", ":uint8": "uint8_t"}


def test_foreign_functions_match_header():
    code = strip_lisp(open(GLUE).read(), keep_strings=True)
    hdr = header_text()
    funs = re.findall(r'\(cffi:defcfun \("(semz_[a-z_]+)"\s+[a-z-]+\)\s+(:[a-z0-9]+)((?:\s*\([a-z-]+ :[a-z0-9]+\))*)\)', code)
    assert len(funs) >= 9
    for name, ret, args in funs:
        m = re.search(r"\b(\w[\w\s\*]*?)\b" + name + r"\s*\(([^)]*)\)\s*;", hdr)
        assert m, f"{name} is not declared in include/semz_gpu.h"
        nargs = len([a for a in m.group(2).split(",") if a.strip() and a.strip() != "void"])
        assert nargs == len(re.findall(r"\([a-z-]+ :[a-z0-9]+\)", args)), name
        c_ret = m.group(1).strip()
        assert {":int": "int", ":void": "void", ":string": "const char *"}[ret] == c_ret, (name, ret, c_ret)


def test_foreign_structs_match_header():
    code = strip_lisp(open(GLUE).read())
    hdr = header_text()
    for lisp_name, c_name in (("semz-result", "semz_result"), ("semz-stream-info", "semz_stream_info"),
                              ("semz-member", "semz_member")):
        body = re.search(r"\(cffi:defcstruct " + lisp_name + r"((?:\s*\([a-z-]+ :[a-z0-9]+(?: :count \d+)?\))+)\)", code).group(1)
        lisp_fields = [(n.replace("-", "_"), C_TYPES[t], int(c) if c else 1)
                       for n, t, c in re.findall(r"\(([a-z-]+) (:[a-z0-9]+)(?: :count (\d+))?\)", body)]
        cbody = re.search(r"typedef struct " + c_name + r"\s*\{(.*?)\}\s*" + c_name + r"\s*;", hdr, flags=re.S).group(1)
        c_fields = [(n, t, int(c) if c else 1)
                    for t, n, c in re.findall(r"(\w+)\s+(\w+)(?:\[(\d+)\])?\s*;", cbody)]
        assert lisp_fields == c_fields, (lisp_name, lisp_fields, c_fields)


def test_constants_match_headers():
    code = strip_lisp(open(GLUE).read())
    hdr = header_text()
    codes = dict(re.findall(r"(:[a-z0-9-]+) (\d+)", re.search(r"\(defparameter \*format-codes\*\s+'\(([^)]*)\)", code).group(1)))
    enum = dict(re.findall(r"SEMZ_FMT_(\w+) = (\d+)", hdr))
    assert {k[1:].upper().replace("-", "_"): v for k, v in codes.items()} == enum
    status = open(os.path.join(ROOT, "include", "semz_status.h")).read()
    values = {int(v) for v in re.findall(r"SEMZ_\w+ = (\d+)", status)}
    listed = re.search(r"\(defparameter \*header-statuses\*\s+'\(([^)]*)\)", code).group(1)
    assert {int(v) for v in listed.split()} <= values
    for v, name in ((1, "SEMZ_EOF"), (44, "SEMZ_E_ZLIB_UNKNOWN_DICT"), (200, "SEMZ_E_OUT_CAP")):
        assert re.search(name + r" = " + str(v) + r"\b", status)
    # the options the glue sets are options the library knows
    code = strip_lisp(open(GLUE).read(), keep_strings=True)
    api = open(os.path.join(ROOT, "decompress_amd", "csrc", "semz_api.hip")).read()
    for opt in set(re.findall(r'\(cons "([a-z-]+)"', code)):
        assert f'"{opt}"' in api, opt


def glue_messages():
    """*messages* of lisp/gpu.lisp: {status: (control string, source keywords...)}"""
    text = open(GLUE).read()
    body = text[text.index("(defparameter *messages*"):text.index("(defstruct (failure")]
    out = {}
    for m in re.finditer(r'\((\d+) "((?:[^"\\]|\\.)*)"((?:\s+:[a-z-]+)*)\)', body, flags=re.S):
        out[int(m.group(1))] = (m.group(2),) + tuple(k[1:] for k in m.group(3).split())
    return out


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_message_control_strings_are_the_references():
    """Every format-control string the glue signals with is byte-identical to one of the reference's `die` calls."""
    ref = set()
    for path in glob.glob(os.path.join(REF, "*.lisp")):
        ref.update(re.findall(r'\(die\s+"((?:[^"\\]|\\.)*)"', open(path).read(), flags=re.S))
    assert len(ref) > 90
    msgs = glue_messages()
    assert len(msgs) > 90
    for status, entry in msgs.items():
        assert entry[0] in ref, (status, entry[0])
    # and the other way round: every `die` of the hot path has a status that signals it (sha256.lisp:142 is unreachable
    # per unit)
    used = {e[0] for e in msgs.values()}
    assert ref - used == {"Input data is too long for SHA-256."}, ref - used
    # the number of arguments fits the directives that take one
    for status, entry in msgs.items():
        directives = re.findall(r"~(?:[0-9v]*(?:,'?.)?)?([dxXa])", entry[0])
        vs = len(re.findall(r"~v", entry[0]))
        assert len(entry) - 1 == len(directives) + vs, (status, entry)


def test_python_mirror_has_the_same_messages():
    """decompress_amd/api.py (the executed mirror) and lisp/gpu.lisp signal with the same table; every status of
    include/semz_status.h that stands for a `die` site is in it."""
    import sys
    sys.path.insert(0, ROOT)
    from decompress_amd import api
    msgs = glue_messages()
    assert api.MESSAGES == msgs
    status = open(os.path.join(ROOT, "include", "semz_status.h")).read()
    values = {int(v) for v in re.findall(r"SEMZ_\w+ = (\d+)", status)}
    no_message = {0, 1, 44, 75, 107}          # ok, eof, the dictionary condition, two divergences without a reference message
    assert {v for v in values if v < 200} - no_message == set(msgs)
    listed = re.search(r"\(defparameter \*header-statuses\*\s+'\(([^)]*)\)", strip_lisp(open(GLUE).read())).group(1)
    assert {int(v) for v in listed.split()} == api._HEADER_STATUSES
    assert re.search(r"\(defvar \*chunk-size\* \(\* 4 1024 1024\)", open(GLUE).read()) and api.CHUNK_SIZE == 4 << 20


def test_lisp_format_subset():
    import sys
    sys.path.insert(0, ROOT)
    from decompress_amd.api import lisp_format, Failure
    assert lisp_format("Invalid selector: ~d", [7]) == "Invalid selector: 7"
    assert lisp_format("(wanted ~8,'0x, got ~8,'0x).", [0x1234, 0xABCDEF01]) == "(wanted 00001234, got ABCDEF01)."
    assert lisp_format("method: ~x", [255]) == "method: FF" and lisp_format("~2,'0x ~2,'0x", [0x42, 5]) == "42 05"
    assert lisp_format("(expected ~v,'0x, got ~v,'0x).", [8, 1, 8, 2]) == "(expected 00000001, got 00000002)."
    assert lisp_format("is not ~\n        encoded", []) == "is not encoded"
    assert lisp_format("~a", ["text"]) == "text"
    f = Failure(60, 0x4259)
    assert f.message() == "Incorrect bzip2 magic bytes: 42 59"
    assert Failure(155, 1, 0, 0xAB, 4, 0xCD).message() == "Incorrect block checksum (expected 000000CD, got 000000AB)."
