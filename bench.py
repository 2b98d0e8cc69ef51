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


def build_workload(fmt, nunits, unit_bytes, seed, workers):
    """Returns (input bytes, [Unit], total_out, total_in, kind)."""
    from decompress_amd import _lib
    step = max(1, nunits // (workers * 4))
    jobs = [(lo, min(nunits, lo + step), unit_bytes, seed) for lo in range(0, nunits, step)]
    fn = _make_gzip_members if fmt == "gzip" else _make_zlib_records
    if workers > 1:
        with mp.get_context("fork").Pool(workers) as pool:
            parts = pool.map(fn, jobs)
    else:
        parts = [fn(j) for j in jobs]
    blobs = [b for p in parts for b in p]
    kind = _lib.UNIT_GZIP if fmt == "gzip" else _lib.UNIT_ZLIB
    data = b"".join(blobs)
    units, off = [], 0
    for i, b in enumerate(blobs):
        # all-members semantics: a member may look ahead to the end of the input
        lim = len(data) if fmt == "gzip" else off + len(b)
        units.append(_lib.Unit(in_off=off, in_limit=lim, out_off=i * unit_bytes, out_cap=unit_bytes, kind=kind))
        off += len(b)
    return data, units, nunits * unit_bytes, len(data), kind, blobs


def cpu_baseline(fmt, blobs, unit_bytes, budget_s=12.0):
    """Oracle (CPU port of the reference algorithm) on a bounded sample, one core."""
    from oracle import oracle
    t0 = time.perf_counter()
    done = 0
    for b in blobs:
        if fmt == "gzip":
            r, o, _ = oracle.gzip_member(b, unit_bytes)
        else:
            r, o, _ = oracle.zlib_member(b, False, None, unit_bytes)
        assert r.status == 0 and len(o) == unit_bytes
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done * unit_bytes / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"{done} of the same {fmt} units ({done * unit_bytes / 2**20:.0f} MiB out) through oracle/ "
                      f"(C restatement of the reference), 1 thread, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--format", default="gzip", choices=["gzip", "zlib"])
    ap.add_argument("--units", type=int, default=0)
    ap.add_argument("--unit-bytes", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None

    if a.format == "gzip":
        nunits, unit_bytes = a.units or 4096, a.unit_bytes or 262144
        workload = f"{nunits} independent {unit_bytes // 1024} KiB gzip -6 members per GPU (BASELINE configs[1]), CRC-32 + ISIZE verified"
    else:
        nunits, unit_bytes = a.units or (1 << 18), a.unit_bytes or 4096
        workload = f"{nunits} zlib -6 records of {unit_bytes} B per GPU (BASELINE configs[4] shape), Adler-32 verified"

    from decompress_amd import _lib
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(1, min(16, ncpu // max(1, world)))
    t_gen = time.perf_counter()
    data, units, total_out, total_in, kind, blobs = build_workload(a.format, nunits, unit_bytes, 0x5EED0001 + rank * 1000003, workers)
    t_gen = time.perf_counter() - t_gen

    # GPU initialisation only after the corpus workers (forked) are gone
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    eng = _lib.Engine(local)
    uarr = (_lib.Unit * nunits)(*units)
    d_in = eng.dev_alloc(total_in + 64)
    d_units = eng.dev_alloc(C.sizeof(uarr))
    d_out = eng.dev_alloc(total_out + 64)
    d_res = eng.dev_alloc(nunits * C.sizeof(_lib.Result))
    eng.upload(d_in, data)
    eng.upload(d_units, bytes(uarr))

    def step():
        eng.decode_batch_device(kind, d_in, total_in, d_units, nunits, d_out, total_out, d_res)

    def barrier():
        eng.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    for _ in range(a.warmup):
        step()
    barrier()
    eng.set_option("profile", 1)
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
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # correctness of what was timed: every unit OK, outputs byte-exact on a sample
    raw = eng.download(d_res, nunits * C.sizeof(_lib.Result))
    res = (_lib.Result * nunits).from_buffer_copy(raw)
    bad = [i for i in range(nunits) if res[i].status != 0 or res[i].out_len != unit_bytes]
    if bad:
        raise SystemExit(f"rank {rank}: {len(bad)} units failed, first {bad[0]} status {res[bad[0]].status}")
    for i in (0, nunits // 2, nunits - 1):
        got = eng.download(d_out, unit_bytes, offset=i * unit_bytes)
        want = zlib.decompress(blobs[i], 47)
        if got != want:
            raise SystemExit(f"rank {rank}: unit {i} output differs from zlib")

    if rank == 0:
        ms_step = dt / a.steps * 1e3
        value = world * total_out / (dt / a.steps) / 1e9
        kname = "inflate_kernel"
        kms = [ms for n, ms in spans if n == kname]
        kavg = sum(kms) / len(kms) if kms else float("nan")
        achieved = (total_in + total_out) / (kavg * 1e-3) / 1e9
        line = {
            "metric": METRIC, "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "format": a.format, "units_per_gpu": nunits, "unit_bytes": unit_bytes,
                       "compressed_bytes_per_gpu": total_in, "decompressed_bytes_per_gpu": total_out,
                       "sharding": f"units by rank x{world}, no collective", "corpus_gen_s": round(t_gen, 1)},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "algorithmic_bytes_per_launch": total_in + total_out,
                         "kernel_ms_avg": round(kavg, 4),
                         "other_kernels_ms_avg": {n: round(sum(m for k, m in spans if k == n) / max(1, len([1 for k, _ in spans if k == n])), 4)
                                                  for n in sorted({k for k, _ in spans}) if n != kname}},
        }
        if not a.no_cpu and world >= 1:
            line["cpu_baseline"] = cpu_baseline(a.format, blobs, unit_bytes)
        print(json.dumps(line), flush=True)

    for p in (d_in, d_units, d_out, d_res):
        eng.dev_free(p)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
