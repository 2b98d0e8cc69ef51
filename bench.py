#!/usr/bin/env python3
"""bench.py -- batched unit decode on MI355X, device-resident.

One "step" = one pass of the hot path over one batch of synthetic units: the
C-ABI call semz_decode_batch_device (decode kernels + checksum/verify kernel)
over inputs already resident in HBM, outputs left in HBM.

Default workload (N=1): BASELINE.json configs[1], 4096 independent 256 KiB gzip
members (gzip -6 of seeded Zipf text), CRC-32 verified on the GPU.
Multi-GPU (--gpus N, launched by torch.distributed.run): units shard by rank, every rank
decodes its own 4096 members (weak scaling); no data-path collective.

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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# what the SQ counters (profiles/r01_*_pmc_sq.txt) say bounds the dominant kernel in practice
ISSUE_NOTES = {
    "gzip": "instruction-issue bound, the scalar unit first (four waves per SIMD share it and take turns at its priorities): 6.8 wave-instructions per output byte, "
            "2.79 G scalar + 3.48 G vector + 0.66 G branch + 0.42 G LDS per GiB (profiles/r02_gzip_pmc_sq.txt); traffic above "
            "the algorithmic bytes is one 128-byte line per far match from the Infinity Cache (DESIGN.md section 3.1)",
    "zlib": "instruction-issue bound like gzip, plus a table build per 4 KiB record",
    "bzip2": "bz_decode_kernel: the latency of two serial waves per block (a 600-block batch is one wave per SIMD; 4800 blocks run at "
             "12 blocks per CU and 2.8 times the rate); bz_walk_kernel: one dependent 4-byte read per output byte out of the XCD's L2",
    "xz": "serial range decoder on the scalar unit: 25 instructions per decoded bit, 112 G scalar + 50 G vector instructions per "
          "launch at one scalar instruction per 4.9 clocks and SIMD (the unit's limit is 4), two waves per SIMD",
}
METRIC = "decompressed GB/s (node) batched gzip/bzip2 blocks @1/2/4/8 GPUs; % HBM peak"
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def _make_gzip_members(args):
    """Worker: members [lo, hi) of the corpus -> list of gzip -6 blobs."""
    import corpus
    lo, hi, unit_bytes, seed = args
    out = []
    for i in range(lo, hi):
        d = corpus.text(unit_bytes, seed=seed + i)
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        out.append(co.compress(d) + co.flush())
    return out


def _make_zlib_records(args):
    import corpus
    lo, hi, unit_bytes, seed = args
    big = corpus.text((hi - lo) * unit_bytes, seed=seed + lo)
    return [zlib.compress(big[k * unit_bytes:(k + 1) * unit_bytes], 6) for k in range(hi - lo)]


def _make_bzip2_streams(args):
    """bzip2 -9 streams of `unit_bytes` each (several 900k blocks per stream)."""
    import bz2
    import corpus
    lo, hi, unit_bytes, seed = args
    return [bz2.compress(corpus.text(unit_bytes, seed=seed + i), 9) for i in range(lo, hi)]


def _make_xz_streams(args):
    """xz streams with 1 MiB blocks, x86 BCJ + LZMA2 preset 6, CRC64 (or SHA-256 when seed is odd... no: see check)."""
    import subprocess
    import corpus
    lo, hi, unit_bytes, seed, check = args
    out = []
    for i in range(lo, hi):
        d = corpus.x86ish(unit_bytes, seed=seed + i)
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


def build_workload(fmt, nunits, unit_bytes, seed, workers, check="crc64", replicate=1):
    """Returns dict(data, uarr, nunits, total_out, total_in, kind, blobs, out_cap)."""
    from decompress_amd import _lib
    if fmt in ("gzip", "zlib"):
        step = max(1, nunits // (workers * 4))
        jobs = [(lo, min(nunits, lo + step), unit_bytes, seed) for lo in range(0, nunits, step)]
        parts = _pool_map(_make_gzip_members if fmt == "gzip" else _make_zlib_records, jobs, workers)
        blobs = [b for p in parts for b in p]
        kind = _lib.UNIT_GZIP if fmt == "gzip" else _lib.UNIT_ZLIB
        data = b"".join(blobs)
        uarr = (_lib.Unit * nunits)()
        off = 0
        for i, b in enumerate(blobs):
            # all-members semantics: a gzip member may look ahead to the end of the input
            lim = len(data) if fmt == "gzip" else off + len(b)
            uarr[i] = _lib.Unit(in_off=off, in_limit=lim, out_off=i * unit_bytes, out_cap=unit_bytes, kind=kind)
            off += len(b)
        return dict(data=data, uarr=uarr, nunits=nunits, total_out=nunits * unit_bytes, total_in=len(data), kind=kind,
                    blobs=blobs, out_cap=nunits * unit_bytes, sample_out=unit_bytes)
    # bzip2 / xz: `nunits` streams of `unit_bytes`; the blocks inside are found by the host splitter
    jobs = [(i, i + 1, unit_bytes, seed) + ((check,) if fmt == "xz" else ()) for i in range(nunits)]
    parts = _pool_map(_make_bzip2_streams if fmt == "bzip2" else _make_xz_streams, jobs, workers)
    blobs = [b for p in parts for b in p] * replicate
    nunits *= replicate
    data = b"".join(blobs)
    uarr, info = _split(_lib.FMT_BZIP2 if fmt == "bzip2" else _lib.FMT_XZ, data)
    total_out = nunits * unit_bytes
    kind = _lib.UNIT_BZIP2_BLOCK if fmt == "bzip2" else _lib.UNIT_XZ_BLOCK
    return dict(data=data, uarr=uarr, nunits=len(uarr), total_out=total_out, total_in=len(data), kind=kind, blobs=blobs,
                out_cap=total_out + 4096, sample_out=unit_bytes)


_CPU_FMT = {"gzip": 2, "zlib": 1, "bzip2": 3, "xz": 8}


def _oracle_worker(args):
    """One core: decode blobs[lo::stride] round and round through oracle/ until the budget is spent."""
    fmt, blobs, sample_out, budget_s, start = args
    from oracle import oracle
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 <= budget_s:
        b = blobs[(start + done) % len(blobs)]
        r, o = oracle.decompress(_CPU_FMT[fmt], b, True, cap=sample_out + 16)
        assert r.status == 0 and len(o) == sample_out
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


def cpu_baseline(fmt, blobs, sample_out, ncores, budget_s=8.0):
    """CPU rates on the GPU box's host cores, all on bounded samples of the same workload and all before the GPU is
    initialised (the workers are forked).  `value` is the oracle (C restatement of the reference's algorithm) on
    every core of this process's CPU share; `single` is one core of it; `proxies` are the system tools, one process
    each.  The reference itself is Common Lisp and no Lisp implementation exists on this box."""
    import shutil
    sample = blobs[:max(1, min(len(blobs), 4 * ncores))]
    one, dt1 = _oracle_worker((fmt, sample, sample_out, budget_s, 0))
    jobs = [(fmt, sample, sample_out, budget_s, i) for i in range(ncores)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(ncores) as pool:
        parts = pool.map(_oracle_worker, jobs)
    wall = time.perf_counter() - t0
    alln = sum(d for d, _ in parts)
    tool = {"gzip": "gzip", "zlib": None, "bzip2": "bzip2", "xz": "xz"}[fmt]
    proxies = {}
    if tool:
        k = max(1, min(len(sample), (256 << 20) // max(1, sample_out)))      # at most 256 MiB of output through the tool
        proxies[f"{tool} -dc (1 process)"] = _tool_rate(tool, b"".join(sample[:k]), k * sample_out)
    return {"value": round(alln * sample_out / wall / 1e9, 4), "unit": "GB/s", "cores": ncores, "kind": "port",
            "single": round(one * sample_out / dt1 / 1e9, 4), "all_cores": round(alln * sample_out / wall / 1e9, 4),
            "sbcl": "unavailable on this box (no Common Lisp implementation in the image): the reference itself cannot be timed",
            "proxies_GBps": proxies,
            "sample": f"oracle/ (C restatement of the reference) over {len(sample)} of the bench's {fmt} inputs, cycled: "
                      f"1 thread did {one} decodes ({one * sample_out / 2**20:.0f} MiB out) in {dt1:.1f} s; {ncores} forked workers "
                      f"did {alln} decodes ({alln * sample_out / 2**20:.0f} MiB out) in {wall:.1f} s wall; system tool on "
                      f"{'%d' % (k if tool else 0)} concatenated inputs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--format", default="gzip", choices=["gzip", "zlib", "bzip2", "xz"])
    ap.add_argument("--units", type=int, default=0)
    ap.add_argument("--unit-bytes", type=int, default=0)
    ap.add_argument("--check", default="crc64", choices=["crc64", "sha256", "crc32", "none"])
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-buffer (PCIe-inclusive) call")
    ap.add_argument("--no-spans", action="store_true",
                    help="time the steps without the per-kernel event pairs (large bzip2 batches overlap kernels on two streams "
                         "only then); the roofline object then speaks of the whole step")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="engine option for the run (semz_set_option)")
    ap.add_argument("--batch-budget", type=int, default=0,
                    help="end-to-end call: device bytes a batch may take (the library's option of that name); a smaller "
                         "value than the workload needs makes the call decode window by window")
    ap.add_argument("--workers", type=int, default=0,
                    help="corpus generation processes (default: the CPU share); 1 = no child processes at all, "
                         "which is what a run under rocprofv3 --pmc wants: pool workers forked under the profiler "
                         "inherit its SIGTERM handler and can hang when the pool is torn down")
    ap.add_argument("--replicate", type=int, default=0,
                    help="bzip2/xz: decode every generated stream this many times over (bounds corpus generation time)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None

    if a.format == "gzip":
        nunits, unit_bytes = a.units or 4096, a.unit_bytes or 262144
        workload = f"{nunits} independent {unit_bytes // 1024} KiB gzip -6 members per GPU (BASELINE configs[1]), CRC-32 + ISIZE verified"
    elif a.format == "zlib":
        nunits, unit_bytes = a.units or (1 << 20), a.unit_bytes or 4096
        workload = f"{nunits} zlib -6 records of {unit_bytes} B per GPU (BASELINE configs[4]), Adler-32 verified"
    elif a.format == "bzip2":
        nunits, unit_bytes = a.units or 8, a.unit_bytes or (64 << 20)
        workload = (f"{nunits} bzip2 -9 streams of {unit_bytes >> 20} MiB per GPU, block-parallel "
                    "(BASELINE configs[2] shape: 512 MiB of the 4 GiB per GPU), block and stream CRCs verified")
    else:
        nunits, unit_bytes = a.units or 64, a.unit_bytes or (16 << 20)
        a.replicate = a.replicate or 2
        workload = (f"{nunits * a.replicate} xz streams of {unit_bytes >> 20} MiB per GPU ({nunits} distinct, each {a.replicate} times), "
                    f"1 MiB LZMA2 blocks + x86 BCJ, check {a.check} verified (BASELINE configs[3] shape)")
    rep = max(1, a.replicate) if a.format in ("bzip2", "xz") else 1
    if a.format == "bzip2" and rep > 1:
        workload = workload.replace(" per GPU,", f" per GPU ({nunits} distinct, each {rep} times),").replace(f"{nunits} bzip2", f"{nunits * rep} bzip2", 1)

    from decompress_amd import _lib
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = a.workers or max(1, min(16, ncpu // max(1, world)))
    t_gen = time.perf_counter()
    W = build_workload(a.format, nunits, unit_bytes, 0x5EED0001 + rank * 1000003, workers, a.check, rep)
    t_gen = time.perf_counter() - t_gen
    data, uarr, nun, total_out, total_in, kind, blobs = (W[k] for k in ("data", "uarr", "nunits", "total_out", "total_in", "kind", "blobs"))

    # CPU baseline (forked workers) and then GPU initialisation, in that order
    cpu_line = None
    if not a.no_cpu and world == 1 and rank == 0:
        cpu_line = cpu_baseline(a.format, blobs, W["sample_out"], max(1, min(16, ncpu)))
    if world > 1:
        # the path has no exchange step: the process group only carries the barrier and the max over ranks, on the
        # host (gloo), so RCCL is not even loaded
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")

    eng = _lib.Engine(local)
    for kv in a.opt:
        name, value = kv.split("=")
        eng.set_option(name, int(value))
    d_in = eng.dev_alloc(total_in + 64)
    d_units = eng.dev_alloc(C.sizeof(uarr))
    d_out = eng.dev_alloc(W["out_cap"] + 64)
    d_res = eng.dev_alloc(nun * C.sizeof(_lib.Result))
    eng.upload(d_in, data)
    eng.upload(d_units, bytes(uarr))

    def step():
        eng.decode_batch_device(kind, d_in, total_in, d_units, nun, d_out, W["out_cap"], d_res)

    def barrier():
        eng.sync()                                  # hipStreamSynchronize on the engine's stream + hipDeviceSynchronize
        if dist is not None:
            dist.barrier()

    for _ in range(a.warmup):
        step()
    barrier()
    eng.set_option("profile", 0 if a.no_spans else 1)
    eng.profile()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    spans = eng.profile()
    eng.set_option("profile", 0)
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # correctness of what was timed: every unit OK, total size right, outputs byte-exact on a sample
    raw = eng.download(d_res, nun * C.sizeof(_lib.Result))
    res = (_lib.Result * nun).from_buffer_copy(raw)
    bad = [i for i in range(nun) if res[i].status != 0]
    if bad:
        raise SystemExit(f"rank {rank}: {len(bad)} units failed, first {bad[0]} status {res[bad[0]].status}")
    if sum(r.out_len for r in res) != total_out:
        raise SystemExit(f"rank {rank}: decoded {sum(r.out_len for r in res)} bytes, expected {total_out}")
    if a.format in ("gzip", "zlib"):
        for i in (0, nun // 2, nun - 1):
            got = eng.download(d_out, unit_bytes, offset=i * unit_bytes)
            if got != zlib.decompress(blobs[i], 47):
                raise SystemExit(f"rank {rank}: unit {i} output differs from zlib")
    else:
        import bz2
        import lzma
        want = (bz2.decompress if a.format == "bzip2" else lzma.decompress)(blobs[-1])
        got = eng.download(d_out, unit_bytes, offset=res[nun - 1].out_off + res[nun - 1].out_len - unit_bytes)
        if got != want:
            raise SystemExit(f"rank {rank}: output of the last stream differs from the system decoder")

    if rank == 0:
        ms_step = dt / a.steps * 1e3
        value = world * total_out / (dt / a.steps) / 1e9
        names = sorted({k for k, _ in spans})
        avg = {n: sum(m for k, m in spans if k == n) / max(1, a.steps) for n in names}   # ms per step
        if not avg:                                 # --no-spans: the whole step stands in for the dominant kernel
            names, avg = ["whole step (kernels overlap)"], {"whole step (kernels overlap)": ms_step}
        kname = max(avg, key=avg.get)
        kavg = avg[kname]
        achieved = (total_in + total_out) / (kavg * 1e-3) / 1e9
        # HBM bytes per launch of that kernel from the committed PMC passes of this same command
        # (scripts/pmc_traffic.py; rocprofv3 cannot run inside the timed process)
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        pmc_key = f"{a.format}:{nun}x{unit_bytes}:{kname}"
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                ent = json.load(f).get(pmc_key)
            if ent:
                traffic, traffic_src = ent["traffic_bytes_per_launch"], f"profiles/pmc_traffic.json[{pmc_key}]"
        line = {
            "metric": METRIC, "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "format": a.format, "units_per_gpu": nun, "unit_bytes": unit_bytes,
                       "compressed_bytes_per_gpu": total_in, "decompressed_bytes_per_gpu": total_out,
                       "sharding": f"units by rank x{world}, no collective", "corpus_gen_s": round(t_gen, 1)},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": total_in + total_out,
                         "kernel_ms_avg": round(kavg, 4),
                         "other_kernels_ms_avg": {n: round(avg[n], 4) for n in names if n != kname},
                         "note": ISSUE_NOTES.get(a.format)},
        }
        if a.format in ("gzip", "bzip2", "xz") and not a.no_e2e and world == 1:
            # the drop-in call on host buffers (pageable): split on the host, upload, kernels, download
            fmt_id = {"gzip": _lib.FMT_GZIP, "bzip2": _lib.FMT_BZIP2, "xz": _lib.FMT_XZ}[a.format]
            out_buf = (C.c_uint8 * (total_out + 64))()
            C.memset(out_buf, 1, total_out + 64)            # the caller's buffer exists and has been touched
            def c_call():                                   # the C ABI itself, as a Lisp host would call it through CFFI
                n, r, si = C.c_size_t(), _lib.Result(), _lib.StreamInfo()
                t1 = time.perf_counter()
                rc = eng.lib.semz_decode_stream_into(eng.h, fmt_id, data, len(data), 1, C.cast(out_buf, C.c_void_p), total_out + 64,
                                                     C.byref(n), C.byref(r), C.byref(si))
                dt1 = time.perf_counter() - t1
                if rc != 0 or r.status != 0 or n.value != total_out:
                    raise SystemExit(f"semz_decode_stream_into: rc {rc}, status {r.status}, {n.value} bytes")
                return dt1
            c_call()                                        # grows the context's scratch and its buffer pool
            e2e = min(c_call(), c_call())
            line["end_to_end"] = {"value": round(total_out / e2e / 1e9, 3), "unit": "GB/s", "ms": round(e2e * 1e3, 2),
                                  "path": "one semz_decode_stream_into call on pageable host buffers (input and a caller-owned, "
                                          "already touched output buffer): host split, H2D, kernels, D2H in four slices; never `value`"}
            # the same call with less device memory than the batch wants: window by window, two windows in flight
            # (half of what the batch wants, decoder scratch included: windows a quarter of the batch each.  Much smaller
            # windows starve the chip for bzip2 and xz, whose blocks take their tens and hundreds of milliseconds
            # however few of them share it)
            scratch = nun * 10_400_000 if a.format == "bzip2" else 0
            budget = a.batch_budget or (total_in + total_out + scratch) // 2
            eng.set_option("batch-budget", budget)
            c_call()
            e2e_w = min(c_call(), c_call())
            line["end_to_end"]["windowed"] = {"value": round(total_out / e2e_w / 1e9, 3), "unit": "GB/s", "ms": round(e2e_w * 1e3, 2),
                                              "batch_budget_bytes": budget, "windows": eng.get_option("windows")}
            eng.set_option("batch-budget", 0)
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
        print(json.dumps(line), flush=True)

    for p in (d_in, d_units, d_out, d_res):
        eng.dev_free(p)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
