"""Developer script: where the wall time of one semz_decode_stream_into call goes (gzip config 2)."""
import ctypes as C, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from decompress_amd import _lib
W = bench.build_workload("gzip", 4096, 262144, 1, 16)
data = W["data"]; total = W["total_out"]
eng = _lib.Engine(0)
buf = (C.c_uint8 * total)()
C.memset(buf, 1, total)
os.environ["SEMZ_TRACE"] = "1"
for i in range(3):
    t = time.perf_counter()
    rc, n, res, info = eng.decode_stream_into(_lib.FMT_GZIP, data, buf, True)
    dt = time.perf_counter() - t
    print(f"call {i}: rc {rc} status {res.status} {n} bytes in {dt*1e3:.1f} ms = {n/dt/1e9:.2f} GB/s", flush=True)
assert bytes(buf[:262144]) == zlib.decompress(W["blobs"][0], 47)
