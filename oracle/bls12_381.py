"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Pure-Python big-integer restatement of BLS12-381 arithmetic for the Shielder
proof-generation hot path.

PARITY UNPINNED: the reference (/root/reference, Cardinal-Cryptography/zk-apps
@ v2) contains no prover, MSM, NTT or curve arithmetic; the
arithmetic would live in un-vendored crates (halo2curves 0.6.1 /
ark-bls12-381 0.4.0, shielder/Cargo.lock:475-478,
shielder/contract/Cargo.lock:195-196).  This file restates the *published*
curve definition and is pinned only by public known-answer values
(generators, compressed generator encoding, subgroup orders, pairing
bilinearity), all re-derived in tests/test_oracle_constants.py.

Conventions:
  Fr element on the wire : 32-byte little-endian canonical integer < r
  Fq element on the wire : 48-byte little-endian canonical integer < p
  G1 affine on the wire  : x || y (96 B), all-zero = point at infinity
  G2 affine on the wire  : x.c0 || x.c1 || y.c0 || y.c1 (192 B)
  proof encoding         : zcash/IETF compressed big-endian, A(48)|B(96)|C(48)
"""

# --------------------------------------------------------------------------
# constants
# --------------------------------------------------------------------------
X = -0xD201000000010000  # BLS parameter (negative)
R = X**4 - X**2 + 1  # scalar field modulus
P = (X - 1) ** 2 * R // 3 + X  # base field modulus
assert R == 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
assert P == int(
    "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f624"
    "1eabfffeb153ffffb9feffffffffaaab",
    16,
)
FR_TWO_ADICITY = 32
FR_GENERATOR = 7
FR_ROOT_2_32 = pow(FR_GENERATOR, (R - 1) >> 32, R)

G1_X = int(
    "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac58"
    "6c55e83ff97a1aeffb3af00adb22c6bb",
    16,
)
G1_Y = int(
    "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3ed"
    "d03cc744a2888ae40caa232946c5e7e1",
    16,
)
G2_X = (
    int(
        "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d177"
        "0bac0326a805bbefd48056c8c121bdb8",
        16,
    ),
    int(
        "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049"
        "334cf11213945d57e5ac7d055d042b7e",
        16,
    ),
)
G2_Y = (
    int(
        "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c"
        "923ac9cc3baca289e193548608b82801",
        16,
    ),
    int(
        "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab"
        "3f370d275cec1da1aaa9075ff05f79be",
        16,
    ),
)


# --------------------------------------------------------------------------
# Fq / Fq2 helpers.  Fq elements are ints; Fq2 elements are (c0, c1), u^2=-1.
# --------------------------------------------------------------------------
class Fq:
    zero = 0
    one = 1

    @staticmethod
    def add(a, b):
        return (a + b) % P

    @staticmethod
    def sub(a, b):
        return (a - b) % P

    @staticmethod
    def neg(a):
        return (-a) % P

    @staticmethod
    def mul(a, b):
        return a * b % P

    @staticmethod
    def sqr(a):
        return a * a % P

    @staticmethod
    def inv(a):
        return pow(a, P - 2, P)

    @staticmethod
    def is_zero(a):
        return a % P == 0

    @staticmethod
    def muli(a, k):
        return a * k % P


class Fq2:
    zero = (0, 0)
    one = (1, 0)

    @staticmethod
    def add(a, b):
        return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)

    @staticmethod
    def sub(a, b):
        return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)

    @staticmethod
    def neg(a):
        return ((-a[0]) % P, (-a[1]) % P)

    @staticmethod
    def mul(a, b):
        return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)

    @staticmethod
    def sqr(a):
        return Fq2.mul(a, a)

    @staticmethod
    def inv(a):
        d = pow(a[0] * a[0] + a[1] * a[1], P - 2, P)
        return (a[0] * d % P, (-a[1]) * d % P)

    @staticmethod
    def is_zero(a):
        return a[0] % P == 0 and a[1] % P == 0

    @staticmethod
    def muli(a, k):
        return (a[0] * k % P, a[1] * k % P)


B_G1 = 4
B_G2 = (4, 4)  # 4(1+u)


# --------------------------------------------------------------------------
# generic short-Weierstrass (a = 0) arithmetic; affine points are (x, y) or
# None for infinity.
# --------------------------------------------------------------------------
def on_curve(F, b, pt):
    if pt is None:
        return True
    x, y = pt
    return F.sub(F.sqr(y), F.add(F.mul(F.sqr(x), x), b)) == F.zero


def pt_neg(F, pt):
    if pt is None:
        return None
    return (pt[0], F.neg(pt[1]))


def pt_add(F, p1, p2):
    """Affine chord-and-tangent addition (complete, by case analysis)."""
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if F.is_zero(F.add(y1, y2)):
            return None
        lam = F.mul(F.muli(F.sqr(x1), 3), F.inv(F.muli(y1, 2)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.sqr(lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)


# Jacobian arithmetic for speed in scalar multiplication.
def _jac_double(F, p):
    if p is None:
        return None
    X1, Y1, Z1 = p
    if F.is_zero(Y1):
        return None
    A = F.sqr(X1)
    B = F.sqr(Y1)
    C = F.sqr(B)
    D = F.muli(F.sub(F.sub(F.sqr(F.add(X1, B)), A), C), 2)
    E = F.muli(A, 3)
    Fv = F.sqr(E)
    X3 = F.sub(Fv, F.muli(D, 2))
    Y3 = F.sub(F.mul(E, F.sub(D, X3)), F.muli(C, 8))
    Z3 = F.muli(F.mul(Y1, Z1), 2)
    return (X3, Y3, Z3)


def _jac_add(F, p, q):
    if p is None:
        return q
    if q is None:
        return p
    X1, Y1, Z1 = p
    X2, Y2, Z2 = q
    Z1Z1 = F.sqr(Z1)
    Z2Z2 = F.sqr(Z2)
    U1 = F.mul(X1, Z2Z2)
    U2 = F.mul(X2, Z1Z1)
    S1 = F.mul(F.mul(Y1, Z2), Z2Z2)
    S2 = F.mul(F.mul(Y2, Z1), Z1Z1)
    if U1 == U2:
        if S1 == S2:
            return _jac_double(F, p)
        return None
    H = F.sub(U2, U1)
    Rr = F.sub(S2, S1)
    HH = F.sqr(H)
    HHH = F.mul(H, HH)
    V = F.mul(U1, HH)
    X3 = F.sub(F.sub(F.sqr(Rr), HHH), F.muli(V, 2))
    Y3 = F.sub(F.mul(Rr, F.sub(V, X3)), F.mul(S1, HHH))
    Z3 = F.mul(F.mul(Z1, Z2), H)
    return (X3, Y3, Z3)


def _to_jac(F, pt):
    return None if pt is None else (pt[0], pt[1], F.one)


def _from_jac(F, p):
    if p is None or F.is_zero(p[2]):
        return None
    zi = F.inv(p[2])
    zi2 = F.sqr(zi)
    return (F.mul(p[0], zi2), F.mul(p[1], F.mul(zi2, zi)))


def pt_mul(F, pt, k):
    """k * pt, k any integer (reduced mod r is NOT applied; callers reduce)."""
    if pt is None or k == 0:
        return None
    if k < 0:
        return pt_mul(F, pt_neg(F, pt), -k)
    acc = None
    base = _to_jac(F, pt)
    for bit in bin(k)[2:]:
        acc = _jac_double(F, acc)
        if bit == "1":
            acc = _jac_add(F, acc, base)
    return _from_jac(F, acc)


def pt_sum(F, pts):
    acc = None
    for q in pts:
        acc = _jac_add(F, acc, _to_jac(F, q))
    return _from_jac(F, acc)


def msm_naive(F, scalars, points):
    """sum_i scalars[i] * points[i] by independent double-and-add.
    Restates the *definition* that ark_ec::VariableBaseMSM::msm_bigint /
    halo2curves::msm::best_multiexp compute."""
    acc = None
    for s, q in zip(scalars, points):
        s %= R
        if s == 0 or q is None:
            continue
        acc = _jac_add(F, acc, _to_jac(F, pt_mul(F, q, s)))
    return _from_jac(F, acc)


G1 = (G1_X, G1_Y)
G2 = (G2_X, G2_Y)


def g1_mul(k, pt=None):
    return pt_mul(Fq, G1 if pt is None else pt, k % R)


def g2_mul(k, pt=None):
    return pt_mul(Fq2, G2 if pt is None else pt, k % R)


# --------------------------------------------------------------------------
# Fq12 as Fq[w] / (w^12 - 2 w^6 + 2)   (w^6 = 1 + u, u^2 = -1)
# An element is a list of 12 ints.  Deliberately a *different* representation
# from the product's Fq2->Fq6->Fq12 tower, so the two cannot share a bug.
# --------------------------------------------------------------------------
def f12_one():
    return [1] + [0] * 11


def f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    # reduce: w^12 = 2 w^6 - 2
    for k in range(22, 11, -1):
        c = t[k]
        if c:
            t[k - 6] += 2 * c
            t[k - 12] -= 2 * c
    return [v % P for v in t[:12]]


def f12_sqr(a):
    return f12_mul(a, a)


def f12_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def f12_sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def f12_pow(a, e):
    res = f12_one()
    for bit in bin(e)[2:]:
        res = f12_sqr(res)
        if bit == "1":
            res = f12_mul(res, a)
    return res


def f12_conj(a):
    """p^6-Frobenius: w -> -w (w^(p^6) = -w because w^2=v in Fq6, w not)."""
    return [(v if i % 2 == 0 else (-v) % P) for i, v in enumerate(a)]


def f12_inv(a):
    # a^(p^12-2) is far too slow; use conj/norm descent: a * conj(a) lies in
    # Fq6 = Fq[w^2]; invert there by solving a 6x6 linear system over Fq.
    c = f12_conj(a)
    n = f12_mul(a, c)  # only even powers non-zero
    assert all(n[i] == 0 for i in range(1, 12, 2))
    ninv = _f6_inv_linear([n[2 * i] for i in range(6)])
    ninv12 = [0] * 12
    for i in range(6):
        ninv12[2 * i] = ninv[i]
    return f12_mul(c, ninv12)


def _f6_mul_as12(a6, b6):
    a = [0] * 12
    b = [0] * 12
    for i in range(6):
        a[2 * i] = a6[i]
        b[2 * i] = b6[i]
    c = f12_mul(a, b)
    return [c[2 * i] for i in range(6)]


def _f6_inv_linear(a6):
    # Solve M x = e0 where M is the multiplication-by-a6 matrix (6x6 over Fq).
    cols = []
    for j in range(6):
        e = [0] * 6
        e[j] = 1
        cols.append(_f6_mul_as12(a6, e))
    M = [[cols[j][i] for j in range(6)] + [1 if i == 0 else 0] for i in range(6)]
    n = 6
    for c in range(n):
        piv = next(rw for rw in range(c, n) if M[rw][c] % P)
        M[c], M[piv] = M[piv], M[c]
        iv = pow(M[c][c], P - 2, P)
        M[c] = [v * iv % P for v in M[c]]
        for rw in range(n):
            if rw != c and M[rw][c]:
                f = M[rw][c]
                M[rw] = [(v - f * w) % P for v, w in zip(M[rw], M[c])]
    return [M[i][6] for i in range(6)]


def fq2_to_f12(a):
    """c0 + c1 u = (c0 - c1) + c1 w^6."""
    out = [0] * 12
    out[0] = (a[0] - a[1]) % P
    out[6] = a[1] % P
    return out


def _f12_scalar(k):
    return [k % P] + [0] * 11


_W = [0, 1] + [0] * 10
_W2_INV = None
_W3_INV = None


def _untwist(q):
    """E'(Fq2) -> E(Fq12): (x', y') -> (x'/w^2, y'/w^3)  (M-type twist)."""
    global _W2_INV, _W3_INV
    if _W2_INV is None:
        w2 = f12_mul(_W, _W)
        w3 = f12_mul(w2, _W)
        _W2_INV = f12_inv(w2)
        _W3_INV = f12_inv(w3)
    return (f12_mul(fq2_to_f12(q[0]), _W2_INV), f12_mul(fq2_to_f12(q[1]), _W3_INV))


def _line(p1, p2, t):
    """Evaluate at t the line through p1,p2 on E(Fq12) (py_ecc-style)."""
    x1, y1 = p1
    x2, y2 = p2
    xt, yt = t
    if x1 != x2:
        m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    if y1 == y2:
        m = f12_mul(
            f12_mul(_f12_scalar(3), f12_sqr(x1)), f12_inv(f12_mul(_f12_scalar(2), y1))
        )
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    return f12_sub(xt, x1)


def _e12_add(p1, p2):
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2 and y1 == y2:
        m = f12_mul(
            f12_mul(_f12_scalar(3), f12_sqr(x1)), f12_inv(f12_mul(_f12_scalar(2), y1))
        )
    else:
        m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    x3 = f12_sub(f12_sub(f12_sqr(m), x1), x2)
    y3 = f12_sub(f12_mul(m, f12_sub(x1, x3)), y1)
    return (x3, y3)


def miller_loop(p_g1, q_g2):
    """f_{|x|,Q}(P), conjugated because x < 0.  Returns Fq12 (unreduced)."""
    if p_g1 is None or q_g2 is None:
        return f12_one()
    Pt = (_f12_scalar(p_g1[0]), _f12_scalar(p_g1[1]))
    Q = _untwist(q_g2)
    T = Q
    f = f12_one()
    for bit in bin(-X)[3:]:
        f = f12_mul(f12_sqr(f), _line(T, T, Pt))
        T = _e12_add(T, T)
        if bit == "1":
            f = f12_mul(f, _line(T, Q, Pt))
            T = _e12_add(T, Q)
    return f12_conj(f)


FINAL_EXP = (P**12 - 1) // R


def final_exponentiation(f):
    # easy part: f^(p^6-1) then ^(p^2+1); hard part by plain square-multiply.
    f1 = f12_mul(f12_conj(f), f12_inv(f))
    f2 = f12_mul(f12_pow(f1, P * P), f1)
    return f12_pow(f2, (P**4 - P**2 + 1) // R)


def pairing(p_g1, q_g2):
    return final_exponentiation(miller_loop(p_g1, q_g2))


def pairing_product_is_one(pairs):
    f = f12_one()
    for a, b in pairs:
        f = f12_mul(f, miller_loop(a, b))
    return final_exponentiation(f) == f12_one()


# --------------------------------------------------------------------------
# wire encodings
# --------------------------------------------------------------------------
def fr_to_bytes(a):
    return (a % R).to_bytes(32, "little")


def fr_from_bytes(b):
    v = int.from_bytes(b, "little")
    if v >= R:
        raise ValueError("non-canonical Fr")
    return v


def fq_to_bytes(a):
    return (a % P).to_bytes(48, "little")


def g1_to_bytes(pt):
    if pt is None:
        return bytes(96)
    return fq_to_bytes(pt[0]) + fq_to_bytes(pt[1])


def g1_from_bytes(b):
    if b == bytes(96):
        return None
    return (int.from_bytes(b[:48], "little"), int.from_bytes(b[48:96], "little"))


def g2_to_bytes(pt):
    if pt is None:
        return bytes(192)
    (x0, x1), (y0, y1) = pt
    return fq_to_bytes(x0) + fq_to_bytes(x1) + fq_to_bytes(y0) + fq_to_bytes(y1)


def g2_from_bytes(b):
    if b == bytes(192):
        return None
    v = [int.from_bytes(b[48 * i : 48 * i + 48], "little") for i in range(4)]
    return ((v[0], v[1]), (v[2], v[3]))


def _fq_lex_larger(y):
    return y > (P - 1) // 2


def _fq2_lex_larger(y):
    # compare c1 first, then c0 (zcash convention)
    if y[1] != 0:
        return y[1] > (P - 1) // 2
    return y[0] > (P - 1) // 2


def g1_compress(pt):
    """zcash/IETF 48-byte big-endian compressed G1."""
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    out = bytearray(pt[0].to_bytes(48, "big"))
    out[0] |= 0x80
    if _fq_lex_larger(pt[1]):
        out[0] |= 0x20
    return bytes(out)


def g2_compress(pt):
    """96-byte compressed G2: x.c1 || x.c0, flags in the first byte."""
    if pt is None:
        return bytes([0xC0]) + bytes(95)
    (x0, x1), y = pt
    out = bytearray(x1.to_bytes(48, "big") + x0.to_bytes(48, "big"))
    out[0] |= 0x80
    if _fq2_lex_larger(y):
        out[0] |= 0x20
    return bytes(out)


def fq_sqrt(a):
    s = pow(a, (P + 1) // 4, P)
    return s if s * s % P == a % P else None


def fq2_sqrt(a):
    """Square root in Fq2 (p = 3 mod 4), algorithm 9 of Adj & Rodriguez-Henriquez."""
    if Fq2.is_zero(a):
        return (0, 0)
    a1 = _fq2_pow(a, (P - 3) // 4)
    alpha = Fq2.mul(a1, Fq2.mul(a1, a))
    a0 = Fq2.mul(_fq2_pow(alpha, P), alpha)
    if a0 == ((-1) % P, 0):
        return None
    x0 = Fq2.mul(a1, a)
    if alpha == ((-1) % P, 0):
        res = Fq2.mul((0, 1), x0)
    else:
        b = _fq2_pow(Fq2.add(Fq2.one, alpha), (P - 1) // 2)
        res = Fq2.mul(b, x0)
    return res if Fq2.sqr(res) == (a[0] % P, a[1] % P) else None


def _fq2_pow(a, e):
    res = Fq2.one
    for bit in bin(e)[2:]:
        res = Fq2.sqr(res)
        if bit == "1":
            res = Fq2.mul(res, a)
    return res


def g1_decompress(b):
    assert len(b) == 48 and b[0] & 0x80
    if b[0] & 0x40:
        return None
    x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:], "big")
    y = fq_sqrt((x * x * x + 4) % P)
    if y is None:
        raise ValueError("not on curve")
    if _fq_lex_larger(y) != bool(b[0] & 0x20):
        y = P - y
    return (x, y)


def g2_decompress(b):
    assert len(b) == 96 and b[0] & 0x80
    if b[0] & 0x40:
        return None
    x1 = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:48], "big")
    x0 = int.from_bytes(b[48:], "big")
    x = (x0, x1)
    y = fq2_sqrt(Fq2.add(Fq2.mul(Fq2.sqr(x), x), B_G2))
    if y is None:
        raise ValueError("not on curve")
    if _fq2_lex_larger(y) != bool(b[0] & 0x20):
        y = Fq2.neg(y)
    return (x, y)


# --------------------------------------------------------------------------
# deterministic synthetic inputs: SplitMix64
# --------------------------------------------------------------------------
class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def fr(self):
        """Uniform in [0, r) by rejection on 255-bit candidates."""
        while True:
            v = 0
            for i in range(4):
                v |= self.next() << (64 * i)
            v &= (1 << 255) - 1
            if v < R:
                return v


def synthetic_bases_g1(n, step=0xC0FFEE):
    """P0 = G, P_{i+1} = P_i + [step]G  (SURVEY.md §8d)."""
    q = g1_mul(step)
    out = [G1]
    for _ in range(n - 1):
        out.append(pt_add(Fq, out[-1], q))
    return out[:n]


def synthetic_bases_g2(n, step=0xC0FFEE):
    q = g2_mul(step)
    out = [G2]
    for _ in range(n - 1):
        out.append(pt_add(Fq2, out[-1], q))
    return out[:n]
