"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

BN254 (alt_bn128) G1 arithmetic and the radix-2 NTT over its scalar field, for the
KZG-commit-shaped MSM / NTT driver.  BN254 is the curve of the reference's own
proving stack (halo2curves::bn256: shielder/Cargo.toml:26, shielder/Cargo.lock:454-478); the
primitives restated are halo2_proofs::arithmetic::{best_fft, best_multiexp} and
ParamsKZG::commit (crates not in the tree).

PIN STATUS: curve constants are public known answers (p, r prime; (1, 2) on y^2 = x^3 + 3;
r * G = infinity); the 2^28-th root of unity equals halo2curves' published
bn256::Fr::ROOT_OF_UNITY = 7^((r-1)/2^28) (quoted from memory, reproduced by the computation in
tests/test_cpu_oracle.py); the NTT is pinned by its definition (O(N^2) DFT), the MSM by
double-and-add.  No vector of the reference pins them (it holds no prover): PARITY UNPINNED
against the reference itself.
"""
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # base field
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # scalar field
B = 3
G1 = (1, 2)
FR_TWO_ADICITY = 28
FR_GENERATOR = 7  # bn256::Fr::MULTIPLICATIVE_GENERATOR
FR_ROOT_2_28 = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C  # bn256::Fr::ROOT_OF_UNITY


def on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - B) % P == 0


def pt_neg(pt):
    return None if pt is None else (pt[0], (-pt[1]) % P)


def pt_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def pt_mul(pt, k):
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = pt_add(acc, acc)
        if bit == "1":
            acc = pt_add(acc, pt)
    return acc


def msm_naive(scalars, points):
    acc = None
    for s, p in zip(scalars, points):
        acc = pt_add(acc, pt_mul(p, s % R))
    return acc


def g1_to_bytes(pt):
    if pt is None:
        return bytes(64)
    return pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little")


def synthetic_bases(n, step=0xC0FFEE):
    """P_i = [1 + i * step] G (the recipe of zkmi_bn254_bases_synthetic)."""
    q = pt_mul(G1, step)
    out, cur = [], G1
    for _ in range(n):
        out.append(cur)
        cur = pt_add(cur, q)
    return out


def root_of_unity(log_n):
    assert 0 <= log_n <= FR_TWO_ADICITY
    return pow(FR_ROOT_2_28, 1 << (FR_TWO_ADICITY - log_n), R)


def dft_naive(a, inverse=False):
    n = len(a)
    w = root_of_unity(n.bit_length() - 1)
    if inverse:
        w = pow(w, -1, R)
    out = []
    for k in range(n):
        wk, acc, x = pow(w, k, R), 0, 1
        for v in a:
            acc += v * x
            x = x * wk % R
        out.append(acc % R)
    if inverse:
        ninv = pow(n, -1, R)
        out = [v * ninv % R for v in out]
    return out


def ntt(a, inverse=False, coset=False):
    """best_fft semantics; coset: pre-multiply by g^i (forward) / post-multiply by g^-i (inverse)."""
    n = len(a)
    log_n = n.bit_length() - 1
    a = list(a)
    if coset and not inverse:
        a = [v * pow(FR_GENERATOR, i, R) % R for i, v in enumerate(a)]
    w = root_of_unity(log_n)
    if inverse:
        w = pow(w, -1, R)
    # iterative Cooley-Tukey on bit-reversed input
    rev = [int(format(i, "0%db" % log_n)[::-1], 2) if log_n else 0 for i in range(n)]
    a = [a[r] for r in rev]
    m = 1
    while m < n:
        wm = pow(w, n // (2 * m), R)
        for s in range(0, n, 2 * m):
            x = 1
            for j in range(m):
                u, t = a[s + j], a[s + j + m] * x % R
                a[s + j], a[s + j + m] = (u + t) % R, (u - t) % R
                x = x * wm % R
        m *= 2
    if inverse:
        ninv = pow(n, -1, R)
        a = [v * ninv % R for v in a]
        if coset:
            ginv = pow(FR_GENERATOR, -1, R)
            a = [v * pow(ginv, i, R) % R for i, v in enumerate(a)]
    return a


def eval_polynomial(coeffs, point):
    """halo2_proofs::arithmetic::eval_polynomial: Horner from the top coefficient down (coefficients: constant term first)."""
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * point + c) % R
    return acc


def kate_division(coeffs, b):
    """halo2_proofs::arithmetic::kate_division(a, b): the quotient of a(X) by (X - b), len(a) - 1 coefficients; the remainder
    a(b) is dropped.  Synthetic division from the top: q_{n-2} = a_{n-1}, q_{i-1} = a_i + b q_i."""
    q = [0] * (len(coeffs) - 1)
    tmp = 0
    for i in range(len(coeffs) - 1, 0, -1):
        tmp = (coeffs[i] + b * tmp) % R
        q[i - 1] = tmp
    return q


def kzg_open(coeffs, point, srs):
    """(p(point), commit(q)) for q = kate_division(p, point) against the monomial SRS [tau^i] G: the opening a KZG prover
    sends for one polynomial at one point (halo2's poly::kzg::multiopen reduces to this per rotation set)."""
    q = kate_division(coeffs, point)
    return eval_polynomial(coeffs, point), msm_naive(q, srs[: len(q)])


def kzg_open_many(polys, point, v, srs):
    """k polynomials at one point (a rotation set of halo2's multiopen): ([p_j(point)], commit(q)) with q = kate_division of
    f = sum_j v^j p_j -- the verifier checks the one proof against sum_j v^j C_j and sum_j v^j p_j(point)."""
    n = len(polys[0])
    f = [0] * n
    for p in reversed(polys):
        f = [(a * v + b) % R for a, b in zip(f, p)]
    return [eval_polynomial(p, point) for p in polys], kzg_open(f, point, srs)[1]


def grand_product(num, den):
    """halo2_proofs::plonk::permutation::prover::commit's running product (PLONK's z): z_0 = 1, z_{i+1} = z_i num_i / den_i.
    Returns ([z_0 .. z_{n-1}], z_n)."""
    z, acc = [], 1
    for a, b in zip(num, den):
        z.append(acc)
        acc = acc * a % R * pow(b, -1, R) % R
    return z, acc
