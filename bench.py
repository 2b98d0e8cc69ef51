#!/usr/bin/env python3
"""bench.py -- batched unit decode on MI355X, device-resident.

One "step" = one pass of the hot path over one batch of synthetic units: the
C-ABI call semz_decode_batch_device (decode kernels + checksum/verify kernel)
over inputs already resident in HBM, outputs left in HBM.

Default workload (N=1), `value` of the line: BASELINE.json configs[1], 4096
independent 256 KiB gzip members (gzip -6 of seeded Zipf text), CRC-32 verified
on the GPU.  The default run also measures, in the same process and with the
same protocol, the other formats the metric names and puts them under
`formats` (each with its own roofline, cpu_baseline and end_to_end):
  bzip2_c3_share  600 bzip2 -9 blocks (configs[2], one GPU's share of 8)
  bzip2_c3_full   4800 blocks (configs[2] whole, on one GPU)
  zlib_c5         2^20 zlib records of 4 KiB (configs[4])
  xz_c4_crc64     2048 xz blocks of 1 MiB, x86 BCJ + LZMA2, CRC64 (configs[3])
  xz_c4_sha256    the same with SHA-256
Multi-GPU (--gpus N, launched by torch.distributed.run): units shard by rank, every rank
decodes its own 4096 members (weak scaling); no data-path collective.  After the timed loop
rank 0 alone decodes ONE corpus on all N GPUs through semz_open_multi (`end_to_end.multi`).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# what the SQ counters (profiles/*_pmc_sq.txt) say bounds the dominant kernel in practice
ISSUE_NOTES = {
    "gzip": "instruction-issue bound, the scalar unit first (four waves per SIMD share it and take turns at its priorities); "
            "traffic above the algorithmic bytes is one line per far match, re-read through L2 (DESIGN.md section 3.1)",
    "zlib": "instruction-issue bound like gzip at 2.3 bytes per token instead of 4.9, plus a table build per 4 KiB record "
            "(DESIGN.md section 3.1)",
    "bzip2": "bz_decode_kernel: the latency of two serial waves per block; bz_walk_kernel: one dependent 4-byte read per output "
             "byte out of the XCD's L2 (DESIGN.md section 3.2)",
    "xz": "serial range decoder, instruction-issue bound with two waves per SIMD (DESIGN.md section 3.3)",
}
METRIC = "decompressed GB/s (node) batched gzip/bzip2 blocks @1/2/4/8 GPUs; % HBM peak"
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
_T0 = time.perf_counter()


def log(msg):
    """Progress on stderr (stdout carries the one JSON line): a run that is silent for minutes is taken to be hung."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)

UNIT_DT = np.dtype([("in_off", "<u8"), ("in_limit", "<u8"), ("out_off", "<u8"), ("out_cap", "<u8"),
                    ("kind", "<u4"), ("flags", "<u4"), ("in_bit_off", "<u4"), ("param0", "<u4"),
                    ("param1", "<u4"), ("param2", "<u4"), ("param3", "<u4"), ("filters", "<u4"),
                    ("prefix_off", "<u8"), ("prefix_len", "<u8"), ("size_in", "<u8"), ("size_out", "<u8")])
RESULT_DT = np.dtype([("status", "<i4"), ("detail", "<u4"), ("in_end_bit", "<u8"), ("out_len", "<u8"), ("out_off", "<u8"),
                      ("check", "u1", (32,)), ("check_len", "<u4"), ("aux", "<u4")])

# the workloads of the `formats` object (and of --format X): format, units, unit bytes, replication, xz check
SPECS = {
    "gzip_c2": dict(format="gzip", units=4096, unit_bytes=262144),
    "bzip2_c3_share": dict(format="bzip2", units=8, unit_bytes=64 << 20, replicate=1),
    "bzip2_c3_full": dict(format="bzip2", units=8, unit_bytes=64 << 20, replicate=8),
    "zlib_c5": dict(format="zlib", units=1 << 20, unit_bytes=4096),
    "xz_c4_crc64": dict(format="xz", units=64, unit_bytes=16 << 20, replicate=2, check="crc64"),
    "xz_c4_sha256": dict(format="xz", units=64, unit_bytes=16 << 20, replicate=2, check="sha256"),
}
DEFAULT_FORMATS = ["bzip2_c3_share", "bzip2_c3_full", "zlib_c5", "xz_c4_crc64", "xz_c4_sha256"]


def describe(fmt, nunits, unit_bytes, rep, check):
    if fmt == "gzip":
        return f"{nunits} independent {unit_bytes // 1024} KiB gzip -6 members per GPU (BASELINE configs[1]), CRC-32 + ISIZE verified"
    if fmt == "zlib":
        return f"{nunits} zlib -6 records of {unit_bytes} B per GPU (BASELINE configs[4]), Adler-32 verified"
    dup = f" ({nunits} distinct, each {rep} times)" if rep > 1 else ""
    if fmt == "bzip2":
        return (f"{nunits * rep} bzip2 -9 streams of {unit_bytes >> 20} MiB per GPU{dup}, block-parallel (BASELINE configs[2]: "
                f"{nunits * rep * unit_bytes >> 20} MiB of the 4 GiB on this GPU), block and stream CRCs verified")
    return (f"{nunits * rep} xz streams of {unit_bytes >> 20} MiB per GPU{dup}, 1 MiB LZMA2 blocks + x86 BCJ, "
            f"check {check} verified (BASELINE configs[3] shape)")


# ---------------------------------------------------------------- corpus (host only, before the GPU is touched)
# The value distribution of the plaintext: Zipf word text is what the metric is quoted on;
# incompressible bytes (stored blocks) and highly repetitive bytes (long matches, run-length heavy) are the two
# robustness distributions (--data).  Forked pool workers inherit the setting.
DATA_KINDS = ("text", "random", "repetitive")
_DATA = "text"


def _plain(n, seed, x86=False):
    import corpus
    if _DATA == "random":
        return corpus.random_bytes(n, seed=seed)
    if _DATA == "repetitive":
        return corpus.repetitive(n, seed=seed)
    return corpus.x86ish(n, seed=seed) if x86 else corpus.text(n, seed=seed)


def _make_gzip_members(args):
    """Worker: members [lo, hi) of the corpus -> list of gzip -6 blobs."""
    lo, hi, unit_bytes, seed = args
    out = []
    for i in range(lo, hi):
        d = _plain(unit_bytes, seed + i)
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        out.append(co.compress(d) + co.flush())
    return out


def _make_zlib_records(args):
    lo, hi, unit_bytes, seed = args
    big = _plain((hi - lo) * unit_bytes, seed + lo)
    return [zlib.compress(big[k * unit_bytes:(k + 1) * unit_bytes], 6) for k in range(hi - lo)]


def _make_bzip2_streams(args):
    """bzip2 -9 streams of `unit_bytes` each (several 900k blocks per stream)."""
    import bz2
    lo, hi, unit_bytes, seed = args
    level = int(os.environ.get("SEMZ_BENCH_BZ_LEVEL", "9"))          # (experiments: smaller blocks)
    return [bz2.compress(_plain(unit_bytes, seed + i), level) for i in range(lo, hi)]


def _make_xz_streams(args):
    """xz streams with 1 MiB blocks, x86 BCJ + LZMA2 preset 6 and the given block check, through the xz tool
    (CPython's lzma writes one block per stream)."""
    import subprocess
    lo, hi, unit_bytes, seed, check = args
    out = []
    for i in range(lo, hi):
        d = _plain(unit_bytes, seed + i, x86=True)
        out.append(subprocess.run(["xz", "-c", "-T1", "--block-size=1MiB", "--x86", "--lzma2=preset=6", f"--check={check}"],
                                  input=d, stdout=subprocess.PIPE, check=True).stdout)
    return out


def _pool_map(fn, jobs, workers):
    if workers > 1:
        with mp.get_context("fork").Pool(workers) as pool:
            return pool.map(fn, jobs)
    return [fn(j) for j in jobs]


def _split(fmt_id, data):
    """Units of a container through the product's host splitter (semz_split)."""
    from decompress_amd import _lib
    lib = _lib.load()
    units = C.POINTER(_lib.Unit)()
    n = C.c_size_t()
    info = _lib.StreamInfo()
    rc = lib.semz_split(fmt_id, data, len(data), None, None, 0, C.byref(units), C.byref(n), C.byref(info))
    assert rc == 0
    arr = (_lib.Unit * n.value)()
    C.memmove(arr, units, C.sizeof(_lib.Unit) * n.value)
    lib.semz_free(units)
    return arr, info


def make_blobs(fmt, nunits, unit_bytes, seed, workers, check="crc64", cache=None, first=0):
    """The distinct compressed inputs [first, first + nunits) of a workload (unit i is made from seed + i, so the
    ranks of a multi-process run hold contiguous ranges of one corpus).  `cache` (a directory) keeps them between runs on one box: the
    profiling scripts fill it in a run of their own, so that no run under rocprofv3 starts a child process
    (the xz tool, pool workers) after the profiler has initialised the GPU."""
    path = None
    if cache:
        path = os.path.join(cache, f"{fmt}_{_DATA}_{first}+{nunits}x{unit_bytes}_{seed:x}_{check if fmt == 'xz' else 'x'}.npz")
        if os.path.exists(path):
            z = np.load(path)
            ends = np.cumsum(z["lengths"])
            raw = z["data"].tobytes()
            return [raw[e - l:e] for e, l in zip(ends.tolist(), z["lengths"].tolist())]
    if fmt in ("gzip", "zlib"):
        step = max(1, nunits // (workers * 4))
        jobs = [(lo, min(first + nunits, lo + step), unit_bytes, seed) for lo in range(first, first + nunits, step)]
        parts = _pool_map(_make_gzip_members if fmt == "gzip" else _make_zlib_records, jobs, workers)
    else:
        jobs = [(i, i + 1, unit_bytes, seed) + ((check,) if fmt == "xz" else ()) for i in range(first, first + nunits)]
        parts = _pool_map(_make_bzip2_streams if fmt == "bzip2" else _make_xz_streams, jobs, workers)
    blobs = [b for p in parts for b in p]
    if path:
        os.makedirs(cache, exist_ok=True)
        np.savez(path, lengths=np.array([len(b) for b in blobs], dtype=np.int64), data=np.frombuffer(b"".join(blobs), dtype=np.uint8))
    return blobs


def build_workload(fmt, nunits, unit_bytes, seed, workers, check="crc64", replicate=1, cache=None, blobs=None, first=0):
    """Returns dict(data, uarr, nunits, total_out, total_in, kind, blobs, out_cap)."""
    from decompress_amd import _lib
    if blobs is None:
        blobs = make_blobs(fmt, nunits, unit_bytes, seed, workers, check, cache, first)
    if fmt in ("gzip", "zlib"):
        kind = _lib.UNIT_GZIP if fmt == "gzip" else _lib.UNIT_ZLIB
        data = b"".join(blobs)
        lens = np.array([len(b) for b in blobs], dtype=np.uint64)
        offs = np.concatenate([np.zeros(1, np.uint64), np.cumsum(lens)[:-1]])
        u = np.zeros(nunits, dtype=UNIT_DT)
        u["in_off"] = offs
        # all-members semantics: a gzip member may look ahead to the end of the input
        u["in_limit"] = len(data) if fmt == "gzip" else offs + lens
        u["out_off"] = np.arange(nunits, dtype=np.uint64) * np.uint64(unit_bytes)
        u["out_cap"] = unit_bytes
        u["kind"] = kind
        uarr = (_lib.Unit * nunits).from_buffer_copy(u.tobytes())
        return dict(data=data, uarr=uarr, nunits=nunits, total_out=nunits * unit_bytes, total_in=len(data), kind=kind,
                    blobs=blobs, out_cap=nunits * unit_bytes, sample_out=unit_bytes)
    # bzip2 / xz: `nunits` streams of `unit_bytes`; the blocks inside are found by the host splitter
    blobs = blobs * replicate
    nunits *= replicate
    data = b"".join(blobs)
    uarr, info = _split(_lib.FMT_BZIP2 if fmt == "bzip2" else _lib.FMT_XZ, data)
    total_out = nunits * unit_bytes
    kind = _lib.UNIT_BZIP2_BLOCK if fmt == "bzip2" else _lib.UNIT_XZ_BLOCK
    return dict(data=data, uarr=uarr, nunits=len(uarr), total_out=total_out, total_in=len(data), kind=kind, blobs=blobs,
                out_cap=total_out + 4096, sample_out=unit_bytes)


# ---------------------------------------------------------------- CPU baseline (oracle = checker, timed beside the GPU)
_CPU_FMT = {"gzip": 2, "zlib": 1, "bzip2": 3, "xz": 8}


def _oracle_worker(args):
    """One core: decode blobs[start], blobs[start + 1], ... round and round through oracle/ until the budget is spent."""
    fmt, blobs, sample_out, budget_s, start = args
    from oracle import oracle
    lib = oracle.lib()
    out = C.create_string_buffer(sample_out + 16)
    r = oracle.OResult()
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 <= budget_s:
        b = blobs[(start + done) % len(blobs)]
        lib.orc_decompress(_CPU_FMT[fmt], b, len(b), 1, 0, out, sample_out + 16, C.byref(r))
        assert r.status == 0 and r.out_len == sample_out
        done += 1
    return done, time.perf_counter() - t0


def _tool_rate(tool, blob_concat, out_bytes):
    """`tool -dc` over concatenated members / streams, one process (what a user without either library would run)."""
    import shutil
    import subprocess
    if not shutil.which(tool):
        return None
    t0 = time.perf_counter()
    p = subprocess.run([tool, "-dc"], input=blob_concat, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    return round(out_bytes / dt / 1e9, 4) if p.returncode == 0 else None


def cpu_baseline(fmt, blobs, sample_out, ncores, budget_s=8.0, tool_bytes=256 << 20):
    """CPU rates on the GPU box's host cores, all on bounded samples of the same workload and all before the GPU is
    initialised (the workers are forked).  `value` is the oracle (C restatement of the reference's algorithm) on
    every core of this process's CPU share; `single` is one core of it; `proxies` are the system decoders, one process
    each.  The reference itself is Common Lisp and no Lisp implementation exists on this box."""
    sample = blobs[:max(1, min(len(blobs), 4 * ncores if sample_out > 65536 else 4096))]
    one, dt1 = _oracle_worker((fmt, sample, sample_out, budget_s, 0))
    jobs = [(fmt, sample, sample_out, budget_s, i * max(1, len(sample) // ncores)) for i in range(ncores)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(ncores) as pool:
        parts = pool.map(_oracle_worker, jobs)
    wall = time.perf_counter() - t0
    alln = sum(d for d, _ in parts)
    tool = {"gzip": "gzip", "zlib": None, "bzip2": "bzip2", "xz": "xz"}[fmt]
    proxies = {}
    k = max(1, min(len(sample), tool_bytes // max(1, sample_out)))       # a bounded amount of output through the tool
    if tool:
        proxies[f"{tool} -dc (1 process)"] = _tool_rate(tool, b"".join(sample[:k]), k * sample_out)
    else:
        t1 = time.perf_counter()
        for b in sample[:k]:
            zlib.decompress(b)
        proxies["CPython zlib.decompress (1 thread)"] = round(k * sample_out / (time.perf_counter() - t1) / 1e9, 4)
    return {"value": round(alln * sample_out / wall / 1e9, 4), "unit": "GB/s", "cores": ncores, "kind": "port",
            "single": round(one * sample_out / dt1 / 1e9, 4), "all_cores": round(alln * sample_out / wall / 1e9, 4),
            "sbcl": "unavailable on this box (no Common Lisp implementation in the image): the reference itself cannot be timed",
            "proxies_GBps": proxies,
            "sample": f"oracle/ (C restatement of the reference) over {len(sample)} of the bench's {fmt} inputs, cycled: "
                      f"1 thread did {one} decodes ({one * sample_out / 2**20:.0f} MiB out) in {dt1:.1f} s; {ncores} forked workers "
                      f"did {alln} decodes ({alln * sample_out / 2**20:.0f} MiB out) in {wall:.1f} s wall; system decoder on "
                      f"{k} concatenated inputs"}


def c1_single_stream(seed):
    """BASELINE configs[0]: one 64 MiB gzip -6 member through the CPU path (the oracle stands in for SBCL), one thread."""
    import corpus
    from oracle import oracle
    n = 64 << 20
    d = corpus.text(n, seed=seed)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    blob = co.compress(d) + co.flush()
    lib = oracle.lib()
    out = C.create_string_buffer(n + 16)
    r = oracle.OResult()
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        lib.orc_decompress(2, blob, len(blob), 1, 0, out, n + 16, C.byref(r))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert r.status == 0 and r.out_len == n and zlib.crc32(out.raw[:n]) == zlib.crc32(d)
    return {"value": round(n / best / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port", "seconds": round(best, 3),
            "sample": "BASELINE configs[0]: one gzip -6 member of 64 MiB of the seeded text, whole, through oracle/ on one thread, "
                      "best of two; a single member does not shard, so this is the CPU path the GPU engine has no batch for"}


# ---------------------------------------------------------------- GPU legs
class DeviceWorkload:
    """A workload resident in HBM: input, unit table, output span and result table."""

    def __init__(self, eng, W):
        from decompress_amd import _lib
        self.eng, self.W = eng, W
        self.nun = W["nunits"]
        self.d_in = eng.dev_alloc(W["total_in"] + 64)
        self.d_units = eng.dev_alloc(C.sizeof(W["uarr"]))
        self.d_out = eng.dev_alloc(W["out_cap"] + 64)
        self.d_res = eng.dev_alloc(self.nun * C.sizeof(_lib.Result))
        eng.upload(self.d_in, W["data"])
        eng.upload(self.d_units, W["uarr"], C.sizeof(W["uarr"]))

    def step(self):
        W = self.W
        self.eng.decode_batch_device(W["kind"], self.d_in, W["total_in"], self.d_units, self.nun, self.d_out, W["out_cap"], self.d_res)

    def results(self):
        raw = self.eng.download(self.d_res, self.nun * RESULT_DT.itemsize)
        return np.frombuffer(raw, dtype=RESULT_DT)

    def verify(self, fmt, unit_bytes, who=""):
        """Every unit decoded with status 0, the total size is right, sampled outputs equal the system decoder's."""
        W, eng = self.W, self.eng
        res = self.results()
        bad = np.nonzero(res["status"] != 0)[0]
        if bad.size:
            raise RuntimeError(f"{who}{bad.size} units failed, first {int(bad[0])} status {int(res['status'][bad[0]])}")
        if int(res["out_len"].sum()) != W["total_out"]:
            raise RuntimeError(f"{who}decoded {int(res['out_len'].sum())} bytes, expected {W['total_out']}")
        blobs = W["blobs"]
        if fmt in ("gzip", "zlib"):
            picks = (0, self.nun // 2, self.nun - 1)
            for i in picks:
                if eng.download(self.d_out, unit_bytes, offset=i * unit_bytes) != zlib.decompress(blobs[i], 47):
                    raise RuntimeError(f"{who}unit {i} output differs from zlib")
            return f"all {self.nun} units status 0; units {picks} byte-compared with zlib"
        import bz2
        import lzma
        dec = bz2.decompress if fmt == "bzip2" else lzma.decompress
        last = int(res["out_off"][-1] + res["out_len"][-1])
        if eng.download(self.d_out, unit_bytes, offset=last - unit_bytes) != dec(blobs[-1]):
            raise RuntimeError(f"{who}output of the last stream differs from the system decoder")
        if eng.download(self.d_out, unit_bytes, offset=int(res["out_off"][0])) != dec(blobs[0]):
            raise RuntimeError(f"{who}output of the first stream differs from the system decoder")
        return f"all {self.nun} blocks status 0; first and last stream byte-compared with {'bz2' if fmt == 'bzip2' else 'liblzma'}"

    def free(self):
        for p in (self.d_in, self.d_units, self.d_out, self.d_res):
            self.eng.dev_free(p)


def roofline(fmt, W, unit_bytes, spans, steps, ms_step):
    nun = W["nunits"]
    names = sorted({k for k, _ in spans})
    avg = {n: sum(m for k, m in spans if k == n) / max(1, steps) for n in names}   # ms per step
    if not avg:                                 # no spans: the whole step stands in for the dominant kernel
        names, avg = ["whole step (kernels overlap)"], {"whole step (kernels overlap)": ms_step}
    kname = max(avg, key=avg.get)
    kavg = avg[kname]
    alg = W["total_in"] + W["total_out"]
    achieved = alg / (kavg * 1e-3) / 1e9
    # HBM bytes per launch of that kernel from the committed PMC passes of this same workload
    # (scripts/pmc_traffic.py; rocprofv3 cannot run inside the timed process)
    traffic, traffic_src = None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    pmc_key = f"{fmt}:{nun}x{unit_bytes}:{kname}"
    if os.path.exists(pmc_path) and _DATA == "text":        # (the passes were made on the text corpus)
        with open(pmc_path) as f:
            ent = json.load(f).get(pmc_key)
        if ent:
            traffic, traffic_src = ent["traffic_bytes_per_launch"], f"profiles/pmc_traffic.json[{pmc_key}]"
    return {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": alg, "kernel_ms_avg": round(kavg, 4),
            "other_kernels_ms_avg": {n: round(avg[n], 4) for n in names if n != kname},
            "whole_step_frac": round(alg / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "note": ISSUE_NOTES.get(fmt)}


def end_to_end(eng, fmt, W, windowed=True, batch_budget=0):
    """The drop-in call on host buffers (pageable): split on the host, upload, kernels, download.  Never `value`."""
    from decompress_amd import _lib
    data, total_out, nun = W["data"], W["total_out"], W["nunits"]
    out_buf = (C.c_uint8 * (total_out + 64))()
    C.memset(out_buf, 1, total_out + 64)            # the caller's buffer exists and has been touched
    if fmt == "zlib":
        # records have no stream container: the batch call on host buffers is the boundary a record store would use
        res = (_lib.Result * nun)()
        def call():
            t1 = time.perf_counter()
            rc = eng.lib.semz_decode_batch(eng.h, data, len(data), W["uarr"], nun, C.cast(out_buf, C.c_void_p), total_out, res)
            dt1 = time.perf_counter() - t1
            if rc != 0:
                raise RuntimeError(f"semz_decode_batch: rc {rc}")
            return dt1
        call()
        e2e = min(call(), call())
        st = np.frombuffer(res, dtype=RESULT_DT)["status"]
        if st.any():
            raise RuntimeError("semz_decode_batch: a unit failed")
        return {"value": round(total_out / e2e / 1e9, 3), "unit": "GB/s", "ms": round(e2e * 1e3, 2),
                "path": "one semz_decode_batch call on pageable host buffers (input, unit table, output, results); never `value`"}
    fmt_id = {"gzip": _lib.FMT_GZIP, "bzip2": _lib.FMT_BZIP2, "xz": _lib.FMT_XZ}[fmt]

    def c_call():                                   # the C ABI itself, as a Lisp host would call it through CFFI
        n, r, si = C.c_size_t(), _lib.Result(), _lib.StreamInfo()
        t1 = time.perf_counter()
        rc = eng.lib.semz_decode_stream_into(eng.h, fmt_id, data, len(data), 1, C.cast(out_buf, C.c_void_p), total_out + 64,
                                             C.byref(n), C.byref(r), C.byref(si))
        dt1 = time.perf_counter() - t1
        if rc != 0 or r.status != 0 or n.value != total_out:
            raise RuntimeError(f"semz_decode_stream_into: rc {rc}, status {r.status}, {n.value} bytes")
        return dt1
    c_call()                                        # grows the context's scratch and its buffer pool
    e2e = min(c_call(), c_call())
    line = {"value": round(total_out / e2e / 1e9, 3), "unit": "GB/s", "ms": round(e2e * 1e3, 2),
            "path": "one semz_decode_stream_into call on pageable host buffers (input and a caller-owned, "
                    "already touched output buffer): host split, H2D, kernels, D2H in slices; never `value`"}
    if windowed:
        # the same call with half the device memory the batch wants (decoder scratch included): window by window,
        # windows side by side on the context's lanes
        scratch = nun * 10_400_000 if fmt == "bzip2" else 0
        budget = batch_budget or (W["total_in"] + total_out + scratch) // 2
        eng.set_option("batch-budget", budget)
        try:
            c_call()
            e2e_w = min(c_call(), c_call())
            line["windowed"] = {"value": round(total_out / e2e_w / 1e9, 3), "unit": "GB/s", "ms": round(e2e_w * 1e3, 2),
                                "batch_budget_bytes": budget, "windows": eng.get_option("windows"),
                                "of_one_piece": round(e2e / e2e_w, 3)}
        finally:
            eng.set_option("batch-budget", 0)
    return line


def timed_steps(dw, steps, barrier, per_step=None):
    """K steps between two barriers, wall clock.  `per_step` (a list) also receives every step's device time: an event
    on the engine's stream behind each step (SURVEY.md section 8d asks for best and median next to the mean)."""
    eng = dw.eng
    evs = [eng.event() for _ in range(steps + 1)] if per_step is not None else None
    t0 = time.perf_counter()
    if evs:
        eng.record(evs[0])
    for i in range(steps):
        dw.step()
        if evs:
            eng.record(evs[i + 1])
    barrier()
    dt = time.perf_counter() - t0
    if evs:
        per_step.extend(eng.elapsed_ms(evs[i], evs[i + 1]) for i in range(steps))
        for e in evs:
            eng.lib.semz_event_destroy(eng.h, e)
    return dt


def step_stats(per_step):
    v = sorted(per_step)
    return {"best": round(v[0], 4), "median": round(v[len(v) // 2] if len(v) % 2 else (v[len(v) // 2 - 1] + v[len(v) // 2]) / 2, 4),
            "worst": round(v[-1], 4), "unit": "ms", "how": "device time between events behind consecutive steps"}


def run_format(eng, name, spec, W, cpu_line, steps, warmup, t_gen, e2e=True):
    """One entry of `formats`: the same protocol as the main line, one GPU."""
    fmt, unit_bytes = spec["format"], spec["unit_bytes"]
    dw = DeviceWorkload(eng, W)
    try:
        log(f"gpu: {name}: {W['nunits']} units resident, {warmup} warmup + {steps} timed steps")
        for _ in range(warmup):
            dw.step()
        eng.sync()
        eng.set_option("profile", 0)
        per_step = []
        dt = timed_steps(dw, steps, eng.sync, per_step)   # the rate: no event pairs inside a step, kernels of large bzip2 batches overlap
        eng.set_option("profile", 1)
        eng.profile()
        pk = max(2, min(steps, 5))
        timed_steps(dw, pk, eng.sync)               # per-kernel device times (HIP events on the engine's stream)
        spans = eng.profile()
        eng.set_option("profile", 0)
        verified = dw.verify(fmt, unit_bytes, who=f"{name}: ")
    finally:
        dw.free()
    ms_step = dt / steps * 1e3
    ent = {"value": round(W["total_out"] / (dt / steps) / 1e9, 3), "unit": "GB/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(ms_step, 4), "step_ms": step_stats(per_step),
           "config": {"workload": describe(fmt, spec["units"], unit_bytes, spec.get("replicate", 1), spec.get("check", "crc64")),
                      "units_per_gpu": W["nunits"], "compressed_bytes_per_gpu": W["total_in"],
                      "decompressed_bytes_per_gpu": W["total_out"], "corpus_gen_s": round(t_gen, 1)},
           "verified": verified,
           "roofline": roofline(fmt, W, unit_bytes, spans, pk, ms_step)}
    if e2e:
        log(f"gpu: {name}: {ent['value']} GB/s; end-to-end calls")
        ent["end_to_end"] = end_to_end(eng, fmt, W, windowed=fmt in ("bzip2", "xz") and W["total_out"] <= (2 << 30))
    if cpu_line is not None:
        ent["cpu_baseline"] = cpu_line
    return ent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--format", default="gzip", choices=["gzip", "zlib", "bzip2", "xz"])
    ap.add_argument("--units", type=int, default=0)
    ap.add_argument("--unit-bytes", type=int, default=0)
    ap.add_argument("--check", default="crc64", choices=["crc64", "sha256", "crc32", "none"])
    ap.add_argument("--formats", default="default",
                    help="the `formats` object: 'default' = all five beside the default gzip line on one GPU and none otherwise, "
                         "'none', 'all', or a comma list of " + ", ".join(DEFAULT_FORMATS))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-buffer (PCIe-inclusive) calls")
    ap.add_argument("--no-spans", action="store_true",
                    help="time the steps without the per-kernel event pairs (large bzip2 batches overlap kernels on two streams "
                         "only then); the roofline object then speaks of the whole step")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="engine option for the run (semz_set_option)")
    ap.add_argument("--batch-budget", type=int, default=0,
                    help="end-to-end call: device bytes a batch may take (the library's option of that name); a smaller "
                         "value than the workload needs makes the call decode window by window")
    ap.add_argument("--workers", type=int, default=0,
                    help="corpus generation processes (default: the CPU share); 1 = no child processes at all")
    ap.add_argument("--replicate", type=int, default=0,
                    help="bzip2/xz: decode every generated stream this many times over (bounds corpus generation time)")
    ap.add_argument("--corpus-cache", default=os.environ.get("SEMZ_CORPUS_CACHE", ""),
                    help="directory that keeps generated corpora between runs on one box (runs under rocprofv3 read it "
                         "instead of starting child processes)")
    ap.add_argument("--make-corpus", action="store_true", help="fill --corpus-cache for the selected workloads and exit (no GPU)")
    ap.add_argument("--data", default="text", choices=DATA_KINDS,
                    help="value distribution of the plaintext: the Zipf text of the metric, or one of the two robustness "
                         "distributions of SURVEY.md section 8d (incompressible, highly repetitive)")
    a = ap.parse_args()
    global _DATA
    _DATA = a.data

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None

    if a.format == "gzip":
        nunits, unit_bytes = a.units or 4096, a.unit_bytes or 262144
    elif a.format == "zlib":
        nunits, unit_bytes = a.units or (1 << 20), a.unit_bytes or 4096
    elif a.format == "bzip2":
        nunits, unit_bytes = a.units or 8, a.unit_bytes or (64 << 20)
    else:
        nunits, unit_bytes = a.units or 64, a.unit_bytes or (16 << 20)
        a.replicate = a.replicate or 2
    rep = max(1, a.replicate) if a.format in ("bzip2", "xz") else 1
    workload = describe(a.format, nunits, unit_bytes, rep, a.check)
    is_default = a.format == "gzip" and not a.units and not a.unit_bytes
    if a.formats == "default":
        extra = list(DEFAULT_FORMATS) if (is_default and world == 1) else []
    elif a.formats == "none":
        extra = []
    elif a.formats == "all":
        extra = list(DEFAULT_FORMATS)
    else:
        extra = [f for f in a.formats.split(",") if f]
        for f in extra:
            if f not in SPECS:
                raise SystemExit(f"--formats: unknown workload {f}")

    from decompress_amd import _lib
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = a.workers or max(1, min(16, ncpu // max(1, world)))
    cache = a.corpus_cache or None
    seed = 0x5EED0001
    # Across processes the units shard by rank (decompress_amd/shard.py): rank r owns one contiguous range of the
    # job's units -- world x nunits of them, every unit the same size, so the ranges are equal (weak scaling) -- and
    # generates just those.
    from decompress_amd import shard
    first, last = shard.partition([unit_bytes] * (nunits * world), world)[rank]
    assert last - first == nunits

    # ---- host phase: corpora and CPU baselines, all before the GPU is initialised (pool workers are forked)
    t_gen = time.perf_counter()
    log(f"corpus: {workload}")
    W = build_workload(a.format, nunits, unit_bytes, seed, workers, a.check, rep, cache, first=first)
    t_gen = time.perf_counter() - t_gen
    extras, blob_cache, errors = {}, {}, {}
    for name in extra:
        s = SPECS[name]
        t1 = time.perf_counter()
        log(f"corpus: {name}")
        try:
            key = (s["format"], s["units"], s["unit_bytes"], s.get("check", "crc64"))
            if key not in blob_cache:
                blob_cache[key] = make_blobs(s["format"], s["units"], s["unit_bytes"], seed, workers, s.get("check", "crc64"), cache)
            extras[name] = dict(spec=s, blobs=blob_cache[key], t_gen=time.perf_counter() - t1)
        except Exception as e:                  # a missing tool (xz) must not cost the main line
            errors[name] = f"corpus: {type(e).__name__}: {e}"
    if a.make_corpus:
        print(json.dumps({"corpus_cache": cache, "workloads": [a.format] + list(extras), "errors": errors}))
        return

    cpu_line = None
    ncores = max(1, min(16, ncpu))
    if not a.no_cpu and world == 1 and rank == 0:
        log(f"cpu baseline: {a.format}")
        cpu_line = cpu_baseline(a.format, W["blobs"], W["sample_out"], ncores)
        if is_default:
            log("cpu baseline: configs[0], one 64 MiB member")
            cpu_line["c1_single_stream"] = c1_single_stream(seed)
        done = {}
        for name, x in extras.items():
            s = x["spec"]
            key = (s["format"], s["units"], s["unit_bytes"], s.get("check", "crc64"))
            if key not in done:                 # the two bzip2 workloads decode the same streams
                log(f"cpu baseline: {name}")
                done[key] = cpu_baseline(s["format"], x["blobs"], s["unit_bytes"], ncores, budget_s=6.0 if s["format"] == "bzip2" else 4.0,
                                        tool_bytes=128 << 20)
            x["cpu"] = done[key]
    if world > 1:
        # the path has no exchange step: the process group only carries the barrier and the max over ranks, on the
        # host (gloo), so RCCL is not even loaded
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")

    # ---- GPU phase
    log("gpu: open, upload")
    # one rank per GPU; on a box with fewer GPUs than ranks (a rehearsal of the multi-process path on one GPU) the
    # ranks share what is there
    ngpu = 1
    if world > 1:
        import torch
        ngpu = max(1, torch.cuda.device_count())
    eng = _lib.Engine(local % ngpu)
    for kv in a.opt:
        name, value = kv.split("=")
        eng.set_option(name, int(value))
    dw = DeviceWorkload(eng, W)

    def barrier():
        eng.sync()                                  # hipStreamSynchronize on the engine's stream + hipDeviceSynchronize
        if dist is not None:
            dist.barrier()

    log(f"gpu: {a.warmup} warmup + {a.steps} timed steps")
    for _ in range(a.warmup):
        dw.step()
    barrier()
    eng.set_option("profile", 0 if a.no_spans else 1)
    eng.profile()
    per_step = []
    dt = timed_steps(dw, a.steps, barrier, per_step)
    spans = eng.profile()
    eng.set_option("profile", 0)
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # correctness of what was timed: every unit OK, total size right, outputs byte-exact on a sample
    try:
        verified = dw.verify(a.format, unit_bytes, who=f"rank {rank}: ")
    except RuntimeError as e:
        raise SystemExit(str(e))
    if dist is not None:
        # the host-side gather of the path: per-unit (status, size) in unit order over all ranks, first error wins
        res = dw.results()
        flat, first_bad = shard.gather_outcomes(list(zip(res["status"].tolist(), res["out_len"].tolist())), dist)
        if first_bad is not None or sum(n for _, n in flat) != world * W["total_out"]:
            raise SystemExit(f"rank {rank}: gathered outcomes: first failing unit {first_bad}, {sum(n for _, n in flat)} bytes")
        verified += f"; {len(flat)} units gathered over {world} ranks, none failed"
    dw.free()

    line = None
    if rank == 0:
        ms_step = dt / a.steps * 1e3
        value = world * W["total_out"] / (dt / a.steps) / 1e9
        line = {
            "metric": METRIC, "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_step, 4), "step_ms": step_stats(per_step), "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "format": a.format, "units_per_gpu": W["nunits"], "unit_bytes": unit_bytes,
                       "compressed_bytes_per_gpu": W["total_in"], "decompressed_bytes_per_gpu": W["total_out"],
                       "sharding": f"units by rank x{world}, no collective", "corpus_gen_s": round(t_gen, 1),
                       "plaintext": a.data},
            "verified": verified,
            "roofline": roofline(a.format, W, unit_bytes, spans, a.steps, ms_step),
        }
        if a.format in ("gzip", "bzip2", "xz") and not a.no_e2e and world == 1:
            try:
                log("gpu: end-to-end calls")
                line["end_to_end"] = end_to_end(eng, a.format, W, True, a.batch_budget)
            except RuntimeError as e:
                raise SystemExit(str(e))
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line

    # ---- the other formats of the metric, same process, same protocol (one GPU)
    if rank == 0 and (extras or errors):
        formats = {}
        for name in extra:
            if name in errors:
                formats[name] = {"error": errors[name]}
                continue
            x = extras[name]
            s = x["spec"]
            try:
                t1 = time.perf_counter()
                log(f"gpu: {name}")
                Wx = build_workload(s["format"], s["units"], s["unit_bytes"], seed, workers, s.get("check", "crc64"),
                                    s.get("replicate", 1), blobs=x["blobs"])
                tg = x["t_gen"] + time.perf_counter() - t1
                per = {"bzip2": 6, "xz": 4, "zlib": 8}[s["format"]]
                formats[name] = run_format(eng, name, s, Wx, x.get("cpu"), steps=min(a.steps, per), warmup=min(a.warmup, 2),
                                           t_gen=tg, e2e=not a.no_e2e)
                del Wx
            except Exception as e:              # a failing leg is reported as such and never hides the others
                formats[name] = {"error": f"{type(e).__name__}: {e}"}
        line["formats"] = formats

    if world > 1 and not a.no_e2e:
        # one corpus, all N GPUs, inside one C call (semz_open_multi): rank 0 alone, the other ranks wait
        dist.barrier()
        if rank == 0:
            try:
                log(f"gpu: one corpus on {world} GPUs through semz_open_multi")
                eng_m = _lib.Engine([d % ngpu for d in range(world)])
                try:
                    one = end_to_end(eng, a.format, W, windowed=False)
                    multi = end_to_end(eng_m, a.format, W, windowed=False)
                finally:
                    eng_m.close()
                multi["devices"] = world
                multi["one_device"] = {"value": one["value"], "ms": one["ms"]}
                multi["path"] = ("one semz_decode_stream_into call on a semz_open_multi context over all the node's GPUs: one host "
                                 "split, contiguous unit ranges, one host thread + stream per device, host-side gather; pageable buffers")
                line.setdefault("end_to_end", {})["multi"] = multi
            except Exception as e:
                line.setdefault("end_to_end", {})["multi"] = {"error": f"{type(e).__name__}: {e}"}
        dist.barrier()

    if rank == 0:
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
