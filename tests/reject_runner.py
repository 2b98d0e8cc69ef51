"""Runs a vector of tests/vectors_reject.py through the oracle and through the GPU library."""
from oracle import oracle


def oracle_run(v):
    """(status, detail, output bytes) the CPU restatement of the reference gives for the vector."""
    if v["route"] == "stream":
        r, out = oracle.decompress(v["fmt"], v["data"], v["all"], opt=v["opt"], cap=1 << 20)
        return r.status, r.detail, out
    f = v["fields"]
    cap = f.get("cap", 1 << 16)
    if v["kind"] == 0:
        r, out = oracle.inflate(v["data"], 0, 0, f.get("param0") or 32768, f.get("prefix"), cap)
    elif v["kind"] == 1:
        r, out, _ = oracle.zlib_member(v["data"], bool(f.get("flags", 0) & 1), f.get("prefix"), cap)
    elif v["kind"] == 6:
        r, out = oracle.lzma1(v["data"], f["lc"], f["lp"], f["pb"], f["dict_size"], f.get("size"), f.get("eof_mode", 0), cap)
    else:
        raise ValueError(v["kind"])
    return r.status, r.detail, out


def gpu_run(engine, v):
    from decompress_amd import _lib
    if v["route"] == "stream":
        if v["fmt"] == 6:
            engine.set_option("window-size", v["opt"])
        out, res, _ = engine.decode_stream(v["fmt"], v["data"], v["all"])
        return res.status, res.detail, out
    f = v["fields"]
    cap = f.get("cap", 1 << 16)
    data = v["data"]
    u = _lib.Unit(in_off=0, in_limit=len(data), out_off=0, out_cap=cap, kind=v["kind"], flags=f.get("flags", 0),
                  param0=f.get("param0", 0))
    blob = data
    if f.get("prefix") is not None:
        u.prefix_off, u.prefix_len = len(data), len(f["prefix"])
        blob = data + f["prefix"]
    if v["kind"] == 6:
        u.param0 = f["dict_size"]
        u.param1 = f["lc"] | f["lp"] << 8 | f["pb"] << 16
        u.param2 = f.get("eof_mode", 0)
        if f.get("size") is not None:
            u.flags |= _lib.UF_HAVE_OUT_SIZE
            u.size_out = f["size"]
    out, res = engine.decode_batch(blob, [u], cap)
    return res[0].status, res[0].detail, out.raw[:res[0].out_len]
