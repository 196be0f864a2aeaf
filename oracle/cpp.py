"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes wrapper of oracle/zkmi_oracle.cpp
(multi-threaded C++ CPU restatement; bench.py's cpu_baseline "port")."""
import ctypes as C
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "_build", "libzkmi_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _DIR])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _buf(b):
    return (C.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if len(b) else b"\0")


def threads():
    return lib().oracle_threads()


def ntt(data, log_n, inverse=False, coset=False, nthreads=0):
    buf = _buf(data)
    lib().oracle_ntt_fr(buf, C.c_uint32(log_n), C.c_int(inverse), C.c_int(coset), C.c_int(nthreads))
    return bytes(buf)[: len(data)]


def msm_g1(scalars, bases, nthreads=0):
    out = (C.c_uint8 * 96)()
    lib().oracle_msm_g1(_buf(scalars), _buf(bases), C.c_uint64(len(scalars) // 32), out, C.c_int(nthreads))
    return bytes(out)


def msm_g2(scalars, bases, nthreads=0):
    out = (C.c_uint8 * 192)()
    lib().oracle_msm_g2(_buf(scalars), _buf(bases), C.c_uint64(len(scalars) // 32), out, C.c_int(nthreads))
    return bytes(out)


def _csr_args(mats):
    keep = []
    rp = (C.POINTER(C.c_uint32) * 3)()
    cl = (C.POINTER(C.c_uint32) * 3)()
    vl = (C.POINTER(C.c_uint8) * 3)()
    for m, (rowptr, col, val) in enumerate(mats):
        a = (C.c_uint32 * len(rowptr))(*rowptr)
        b = (C.c_uint32 * max(1, len(col)))(*col)
        c = _buf(val)
        keep += [a, b, c]
        rp[m] = C.cast(a, C.POINTER(C.c_uint32))
        cl[m] = C.cast(b, C.POINTER(C.c_uint32))
        vl[m] = C.cast(c, C.POINTER(C.c_uint8))
    return rp, cl, vl, keep


def witness_map(n_vars, n_pub, nc, log_n, mats, z, nthreads=0):
    rp, cl, vl, keep = _csr_args(mats)
    out = (C.c_uint8 * (32 << log_n))()
    lib().oracle_witness_map(C.c_uint32(n_vars), C.c_uint32(n_pub), C.c_uint32(nc), C.c_uint32(log_n), rp, cl, vl, _buf(z), out, C.c_int(nthreads))
    return bytes(out)


def groth16_prove_timed(n_vars, n_pub, nc, log_n, mats, pk, z, r, s, nthreads=0):
    """(proof bytes, seconds spent inside the C++ prover): the marshalling of matrices and key into ctypes buffers --
    Python work a native host would not do -- stays outside the clock.
    pk: dict of wire-format byte strings (alpha_g1, beta_g1, beta_g2, delta_g1,
    delta_g2, a_query, b_g1_query, b_g2_query, h_query, l_query)."""
    import time

    rp, cl, vl, keep = _csr_args(mats)
    out = (C.c_uint8 * 192)()
    names = ["alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2", "a_query", "b_g1_query", "b_g2_query", "h_query", "l_query"]
    bufs = [_buf(pk[k]) for k in names]
    zb, rb, sb = _buf(z), _buf(r), _buf(s)
    fn = lib().oracle_groth16_prove
    t0 = time.perf_counter()
    fn(C.c_uint32(n_vars), C.c_uint32(n_pub), C.c_uint32(nc), C.c_uint32(log_n), rp, cl, vl, *bufs, zb, rb, sb, out, C.c_int(nthreads))
    dt = time.perf_counter() - t0
    return bytes(out), dt


def groth16_prove(n_vars, n_pub, nc, log_n, mats, pk, z, r, s, nthreads=0):
    return groth16_prove_timed(n_vars, n_pub, nc, log_n, mats, pk, z, r, s, nthreads)[0]


def groth16_setup(n_vars, n_pub, nc, log_n, mats, toxic, nthreads=0):
    """Trusted setup with explicit toxic waste on the CPU.  Returns (vk bytes, key dict in the wire
    format groth16_prove takes)."""
    rp, cl, vl, keep = _csr_args(mats)
    N = 1 << log_n
    vk = (C.c_uint8 * (672 + 96 * n_pub))()
    beta_g1, delta_g1 = (C.c_uint8 * 96)(), (C.c_uint8 * 96)()
    a, b1, b2 = (C.c_uint8 * (96 * n_vars))(), (C.c_uint8 * (96 * n_vars))(), (C.c_uint8 * (192 * n_vars))()
    h, l = (C.c_uint8 * (96 * (N - 1)))(), (C.c_uint8 * (96 * (n_vars - n_pub)))()
    rc = lib().oracle_groth16_setup(
        C.c_uint32(n_vars), C.c_uint32(n_pub), C.c_uint32(nc), C.c_uint32(log_n), rp, cl, vl, _buf(toxic), vk, beta_g1, delta_g1,
        a, b1, b2, h, l, C.c_int(nthreads))
    if rc != 0:
        raise ValueError("tau lies in the evaluation domain")
    vkb = bytes(vk)
    key = {
        "alpha_g1": vkb[:96], "beta_g1": bytes(beta_g1), "beta_g2": vkb[96:288], "delta_g1": bytes(delta_g1),
        "delta_g2": vkb[480:672], "a_query": bytes(a), "b_g1_query": bytes(b1), "b_g2_query": bytes(b2),
        "h_query": bytes(h), "l_query": bytes(l),
    }
    return vkb, key
