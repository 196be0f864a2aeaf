"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Radix-2 NTT over the BLS12-381 scalar field.

PARITY UNPINNED: the reference holds no NTT.  This restates the
published semantics of ark_poly::Radix2EvaluationDomain::{fft,ifft}_in_place
and coset variants (ark-poly 0.4.2, pinned at
shielder/contract/Cargo.lock:267-268) = halo2_proofs::arithmetic::best_fft
(shielder/Cargo.lock:436-438):
    forward : out[k] = sum_j in[j] * w^(j k),  w = 7^((r-1)/N)
    inverse : out[j] = N^-1 sum_k in[k] * w^(-j k)
    coset   : forward pre-multiplies in[j] by g^j, inverse post-multiplies by g^-j, g = 7
The definition (O(N^2) DFT) is the pin; the fast transform is checked against it.
"""
from .bls12_381 import R, FR_GENERATOR, FR_ROOT_2_32, FR_TWO_ADICITY


def root_of_unity(log_n):
    assert 0 <= log_n <= FR_TWO_ADICITY
    return pow(FR_ROOT_2_32, 1 << (FR_TWO_ADICITY - log_n), R)


def dft_naive(a, inverse=False):
    n = len(a)
    log_n = n.bit_length() - 1
    w = root_of_unity(log_n)
    if inverse:
        w = pow(w, R - 2, R)
    out = []
    for k in range(n):
        wk = pow(w, k, R)
        acc, x = 0, 1
        for j in range(n):
            acc += a[j] * x
            x = x * wk % R
        out.append(acc % R)
    if inverse:
        ninv = pow(n, R - 2, R)
        out = [v * ninv % R for v in out]
    return out


def _bitrev(i, bits):
    return int(bin(i)[2:].zfill(bits)[::-1], 2) if bits else 0


def ntt(a, inverse=False):
    """Iterative Cooley-Tukey, natural in -> natural out."""
    n = len(a)
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    a = [a[_bitrev(i, log_n)] for i in range(n)]
    w_n = root_of_unity(log_n)
    if inverse:
        w_n = pow(w_n, R - 2, R)
    m = 1
    while m < n:
        w_m = pow(w_n, n // (2 * m), R)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = a[k + j + m] * w % R
                u = a[k + j]
                a[k + j] = (u + t) % R
                a[k + j + m] = (u - t) % R
                w = w * w_m % R
        m *= 2
    if inverse:
        ninv = pow(n, R - 2, R)
        a = [v * ninv % R for v in a]
    return a


def coset_ntt(a, g=FR_GENERATOR):
    x = 1
    b = []
    for v in a:
        b.append(v * x % R)
        x = x * g % R
    return ntt(b)


def coset_intt(a, g=FR_GENERATOR):
    b = ntt(a, inverse=True)
    gi = pow(g, R - 2, R)
    x = 1
    out = []
    for v in b:
        out.append(v * x % R)
        x = x * gi % R
    return out
