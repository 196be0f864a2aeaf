"""ctypes binding of include/zkmi.h (one Python method per C entry point).

Mirrors the reference's caller-side conventions: scalars are
32-byte little-endian `bytes`, points are affine wire bytes, errors surface as
`ZkmiError(code)` with the code names of `ZkpError`
(shielder/mocked_zk/src/errors.rs:3-7) where they apply.
"""
import ctypes as C
import os

ERR = {
    0: "OK",
    -1: "BAD_ARG",
    -2: "NON_CANONICAL",
    -3: "HIP",
    -4: "NO_DEVICE",
    -5: "VerificationError",
    -6: "AccountUpdateError",
    -7: "OperationCombineError",
    -8: "UNSATISFIED",
    -9: "RCCL",
}
PHASES = {
    "msm_sort": 0,
    "msm_accum_g1": 1,
    "msm_reduce_g1": 2,
    "msm_accum_g2": 3,
    "msm_reduce_g2": 4,
    "ntt": 5,
    "witness": 6,
    "misc": 7,
}
MERKLE_TREE_DEPTH = 10
MAX_TREE_HEIGHT = 32
TOKENS_NUMBER = 2


class ZkmiError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        self.name = ERR.get(code, str(code))
        super().__init__(f"zkmi error {code} ({self.name}) {detail}".strip())


def lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libzkmi.so")


class Scalar(C.Structure):
    _fields_ = [("bytes", C.c_uint8 * 32)]


class Account(C.Structure):
    _fields_ = [("balances", (Scalar * 2) * TOKENS_NUMBER)]


class OpPub(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("amount", C.c_uint8 * 16), ("token", Scalar), ("user", Scalar)]


class OpPriv(C.Structure):
    _fields_ = [("user", Scalar)]


class NoteUpdate(C.Structure):
    """zkmi_note_update (include/zkmi.h): semantic inputs of the Poseidon update_note relation."""

    _fields_ = [
        ("amount", C.c_uint8 * 32), ("token", C.c_uint8 * 32), ("user", C.c_uint8 * 32),
        ("new_note", (C.c_uint8 * 32) * 3), ("old_note", (C.c_uint8 * 32) * 3),
        ("tree_height", C.c_uint32),
        ("path_shape", C.c_uint8 * MAX_TREE_HEIGHT), ("path", (C.c_uint8 * 32) * MAX_TREE_HEIGHT),
        ("op_priv_user", C.c_uint8 * 32), ("account", (C.c_uint8 * 32) * 4),
    ]


class NoteCreate(C.Structure):
    """zkmi_note_create: semantic inputs of the creation relation."""

    _fields_ = [("tokens", (C.c_uint8 * 32) * TOKENS_NUMBER), ("note", (C.c_uint8 * 32) * 3)]


class UpdateNoteInput(C.Structure):
    """zkmi_update_note_input: semantic inputs of the withdraw-shaped relation (row a1)."""

    _fields_ = [
        ("amount", Scalar), ("token", Scalar), ("user", Scalar),
        ("old_nullifier", Scalar),
        ("new_note", Scalar * 4),
        ("old_trapdoor", Scalar), ("old_account_hash", Scalar),
        ("path_shape", C.c_uint8 * MERKLE_TREE_DEPTH),
        ("path", Scalar * MERKLE_TREE_DEPTH),
        ("old_account", Scalar * TOKENS_NUMBER),
    ]


class ZkProof(C.Structure):
    _fields_ = [
        ("id", Scalar),
        ("trapdoor_new", Scalar),
        ("trapdoor_old", Scalar),
        ("nullifier_new", Scalar),
        ("acc_old", Account),
        ("acc_new", Account),
        ("op_priv", OpPriv),
        ("merkle_proof", Scalar * MERKLE_TREE_DEPTH),
        ("merkle_proof_leaf_id", C.c_uint32),
    ]


def scalar(b):
    s = Scalar()
    b = bytes(b)
    assert len(b) == 32
    C.memmove(s.bytes, b, 32)
    return s


def scalar_u128(v):
    return scalar(int(v).to_bytes(16, "little") + bytes(16))


def _buf(b):
    return (C.c_uint8 * len(b)).from_buffer_copy(bytes(b)) if len(b) else (C.c_uint8 * 1)()


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels carry their own libamdhip64.so / libhsa-runtime64.so (RPATH $ORIGIN) and ask for them by
    the unversioned name, so a process that loads libzkmi.so (NEEDED libamdhip64.so.7 -> /opt/rocm) BEFORE importing
    torch ends up with two HIP runtimes, and the second one to initialise finds no GPU.  Loading torch's copy first
    (its SONAME is libamdhip64.so.7 too) makes both sides resolve to the same runtime whatever the import order.
    No torch installed, or torch already imported: nothing to do."""
    import importlib.util
    import sys

    if os.environ.get("ZKMI_SHARE_TORCH_HIP", "1") == "0":  # opt-out: a host that never imports torch in this process
        return
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


class Zkmi:
    """Loads libzkmi.so.  Fails loudly if it is missing (no fallback path)."""

    def __init__(self, path=None):
        path = path or os.environ.get("ZKMI_LIB") or lib_path()
        if not os.path.exists(path):
            raise ZkmiError(-4, f"{path} not built; run __graft_entry__.build()")
        _share_hip_runtime_with_torch()
        self.lib = C.CDLL(path)
        self.lib.zkmi_version.restype = C.c_char_p
        self.lib.zkmi_last_error.restype = C.c_char_p
        self.lib.zkmi_last_error.argtypes = [C.c_void_p]
        self._tlib = None

    @property
    def tlib(self):
        """The TESTING library (include/zkmi_testing.h: synthetic bases, the chain stand-in relation, host self-tests) --
        test scaffolding the product library does not export.  It is the A/B + testing build libzkmi_exp.so; the objects it
        creates are the product's own types, and a context made by the product library is the same struct there (one
        source tree), so tests manufacture inputs here and hand them to libzkmi.so."""
        if self._tlib is None:
            if hasattr(self.lib, "zkmi_selftest_fq28"):
                self._tlib = self.lib  # ZKMI_LIB selected the A/B + testing library itself
            else:
                path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libzkmi_exp.so")
                if not os.path.exists(path):
                    raise ZkmiError(-4, f"{path} (testing library) not built; run __graft_entry__.build()")
                self._tlib = C.CDLL(path)
                # objects cross between the two libraries (a product context drives testing-library code and back): refuse
                # to pair builds whose struct layouts differ instead of corrupting memory silently
                mine, theirs = self.abi_layout(self.lib), self.abi_layout(self._tlib)
                if mine != theirs:
                    self._tlib = None
                    raise ZkmiError(-1, f"libzkmi.so and libzkmi_exp.so disagree on struct layouts: {mine} vs {theirs}; rebuild both")
        return self._tlib

    @staticmethod
    def abi_layout(lib):
        """zkmi_abi_layout_probe of `lib` as a tuple."""
        out, n = (C.c_uint64 * 64)(), C.c_uint32(0)
        rc = lib.zkmi_abi_layout_probe(out, C.c_uint32(64), C.byref(n))
        if rc != 0:
            raise ZkmiError(rc, "zkmi_abi_layout_probe")
        return tuple(out[: n.value])

    def src_digest(self):
        """The source digest compiled into the loaded library (zkmi_version(): 'src:<digest>')."""
        v = self.version()
        return v.split("src:", 1)[1].split()[0] if "src:" in v else None

    def hip_versions(self):
        """(HIP_VERSION of the build, version of the HIP runtime this process bound the library to)."""
        b, r = C.c_int32(0), C.c_int32(0)
        self._chk(self.lib.zkmi_hip_versions(C.byref(b), C.byref(r)))
        return b.value, r.value

    def host_info(self):
        """zkmi_host_info: dict(cpus_granted, local_ranks, threads, pool_workers)."""
        out = (C.c_uint32 * 4)()
        self._chk(self.lib.zkmi_host_info(out))
        return dict(zip(("cpus_granted", "local_ranks", "threads", "pool_workers"), out))

    def set_host_threads(self, n):
        self._chk(self.lib.zkmi_set_host_threads(C.c_uint32(n)))

    def version(self):
        return self.lib.zkmi_version().decode()

    def device_count(self):
        n = C.c_int32(0)
        self.lib.zkmi_device_count(C.byref(n))
        return n.value

    def _chk(self, rc, ctx=None):
        if rc != 0:
            detail = ""
            if ctx is not None:
                detail = (self.lib.zkmi_last_error(ctx) or b"").decode()
            raise ZkmiError(rc, detail)

    # ---- host-only helpers --------------------------------------------------
    def g1_generator(self):
        out = (C.c_uint8 * 96)()
        self._chk(self.lib.zkmi_g1_generator(out))
        return bytes(out)

    def g2_generator(self):
        out = (C.c_uint8 * 192)()
        self._chk(self.lib.zkmi_g2_generator(out))
        return bytes(out)

    def g1_compress(self, a):
        out = (C.c_uint8 * 48)()
        self._chk(self.lib.zkmi_g1_compress(_buf(a), out))
        return bytes(out)

    def g1_decompress(self, a):
        out = (C.c_uint8 * 96)()
        self._chk(self.lib.zkmi_g1_decompress(_buf(a), out))
        return bytes(out)

    def g2_compress(self, a):
        out = (C.c_uint8 * 96)()
        self._chk(self.lib.zkmi_g2_compress(_buf(a), out))
        return bytes(out)

    def g2_decompress(self, a):
        out = (C.c_uint8 * 192)()
        self._chk(self.lib.zkmi_g2_decompress(_buf(a), out))
        return bytes(out)

    def g1_mul(self, a, k):
        out = (C.c_uint8 * 96)()
        self._chk(self.lib.zkmi_g1_mul(_buf(a), _buf(k), out))
        return bytes(out)

    def g2_mul(self, a, k):
        out = (C.c_uint8 * 192)()
        self._chk(self.lib.zkmi_g2_mul(_buf(a), _buf(k), out))
        return bytes(out)

    def g1_add(self, a, b):
        out = (C.c_uint8 * 96)()
        self._chk(self.lib.zkmi_g1_add(_buf(a), _buf(b), out))
        return bytes(out)

    def g2_add(self, a, b):
        out = (C.c_uint8 * 192)()
        self._chk(self.lib.zkmi_g2_add(_buf(a), _buf(b), out))
        return bytes(out)

    def pairing(self, p, q):
        out = (C.c_uint8 * 576)()
        self._chk(self.lib.zkmi_pairing(_buf(p), _buf(q), out))
        return bytes(out)

    def msm_plan_query(self, n, shared=False):
        """(digit bits, digits, windows | partitions, buckets per window | partition, log2 segment length, heavy threshold)"""
        out = (C.c_uint32 * 6)()
        self._chk(self.lib.zkmi_msm_plan_query(C.c_uint64(n), C.c_int32(1 if shared else 0), out))
        return tuple(out)

    def comm_unique_id(self):
        """128 bytes from ncclGetUniqueId (rank 0 calls; the host layer hands them to every rank)"""
        out = (C.c_uint8 * 128)()
        self._chk(self.lib.zkmi_comm_unique_id(out))
        return bytes(out)

    def msm_g1_combine(self, windows, n_ranks, nwin, window_bits):
        out = (C.c_uint8 * 96)()
        self._chk(self.lib.zkmi_msm_g1_combine(_buf(windows), C.c_uint32(n_ranks), C.c_uint32(nwin), C.c_uint32(window_bits), out))
        return bytes(out)

    def msm_exchange_layout(self, plan_n, n_ranks=1):
        """zkmi_msm_exchange_layout: dict of the slot geometry of the two RCCL exchanges for a plan of plan_n terms."""
        out = (C.c_uint32 * 8)()
        self._chk(self.lib.zkmi_msm_exchange_layout(C.c_uint64(plan_n), C.c_uint32(n_ranks), out))
        keys = ("nwin", "per_window", "slot_pts_points", "slot_pts_windows", "point_bytes", "c", "seg_log", "top_spread_log")
        return dict(zip(keys, out))

    def msm_g1_combine_partials(self, partials, n_ranks, plan_n, window_split=False):
        """zkmi_msm_g1_combine_partials: the host combination behind the all-gather, on caller-supplied slots."""
        out = (C.c_uint8 * 96)()
        # window_split: False / 0 = point split, True / 1 = window split, Q >= 2 = the 2-D split with Q window ranges
        self._chk(self.lib.zkmi_msm_g1_combine_partials(_buf(partials), C.c_uint32(n_ranks), C.c_uint64(plan_n),
                                                        C.c_int32(int(window_split)), out))
        return bytes(out)

    def host_info_string(self):
        self.lib.zkmi_host_info_string.restype = C.c_char_p
        return self.lib.zkmi_host_info_string().decode()

    def msm_g1_multi(self, ctxs, dptrs, counts, bases):
        """One MSM split by points over several contexts (one per GPU) of this process."""
        k = len(ctxs)
        assert k == len(dptrs) == len(counts) == len(bases) and k > 0
        a_ctx = (C.c_void_p * k)(*[c.h.value for c in ctxs])
        a_ptr = (C.c_void_p * k)(*[int(p) for p in dptrs])
        a_cnt = (C.c_uint64 * k)(*counts)
        a_bas = (C.c_void_p * k)(*[b.h.value for b in bases])
        out = (C.c_uint8 * 96)()
        self._chk(self.lib.zkmi_msm_g1_multi(a_ctx, C.c_uint32(k), a_ptr, a_cnt, a_bas, out))
        return bytes(out)

    def groth16_prove_batch_multi(self, ctxs, pks, z_ptrs, on_device, rs, ss):
        """zkmi_groth16_prove_batch_multi: proof i runs on ctxs[i % len(ctxs)] with the key replica pks[i % len(ctxs)]."""
        nd, n = len(ctxs), len(z_ptrs)
        cs = (C.c_void_p * nd)(*[c.h.value for c in ctxs])
        ks = (C.c_void_p * nd)(*[k.h.value for k in pks])
        ptrs = (C.c_void_p * max(1, n))(*z_ptrs)
        out = (C.c_uint8 * max(1, 192 * n))()
        rc = self.lib.zkmi_groth16_prove_batch_multi(cs, ks, C.c_uint32(nd), C.c_uint32(n), ptrs, C.c_int32(int(on_device)),
                                                     _buf(b"".join(rs)), _buf(b"".join(ss)), out)
        if rc != 0:
            msgs = "; ".join((self.lib.zkmi_last_error(c.h) or b"").decode() for c in ctxs)
            raise ZkmiError(rc, msgs)
        raw = bytes(out)
        return [raw[192 * i: 192 * i + 192] for i in range(n)]

    def groth16_verify(self, vk, publics, proof):
        n_pub = (len(vk) - 672) // 96
        assert len(publics) == 32 * (n_pub - 1)
        rc = self.lib.zkmi_groth16_verify(_buf(vk), C.c_uint32(n_pub), _buf(publics), _buf(proof))
        if rc == 0:
            return True
        if rc == -5:
            return False
        raise ZkmiError(rc)

    # ---- R1CS ---------------------------------------------------------------
    def shielder_r1cs(self, log_n):
        h = C.c_void_p()
        self._chk(self.tlib.zkmi_shielder_r1cs(C.c_uint32(log_n), C.byref(h)))
        return R1cs(self, h)

    def shielder_witness(self, log_n, seed):
        out = (C.c_uint8 * (32 << log_n))()
        self._chk(self.tlib.zkmi_shielder_witness(C.c_uint32(log_n), C.c_uint64(seed), out))
        return bytes(out)

    # ---- Poseidon-5 ---------------------------------------
    def poseidon_spec(self, field=0):
        """(round constants [64][5], mds [5][5]) as ints, straight from the library's generator."""
        rc = (C.c_uint8 * (32 * 64 * 5))()
        mds = (C.c_uint8 * (32 * 25))()
        self._chk(self.lib.zkmi_poseidon_spec(C.c_int32(field), rc, mds))
        r, m = bytes(rc), bytes(mds)
        ints = lambda raw, n: [int.from_bytes(raw[32 * i : 32 * i + 32], "little") for i in range(n)]
        rl, ml = ints(r, 320), ints(m, 25)
        return [rl[5 * i : 5 * i + 5] for i in range(64)], [ml[5 * i : 5 * i + 5] for i in range(5)]

    # ---- update_note relation with real Poseidon hashing --------------------
    def update_note_r1cs(self, log_n, op_kind, tree_height=MERKLE_TREE_DEPTH):
        h = C.c_void_p()
        self._chk(self.lib.zkmi_update_note_r1cs_h(C.c_uint32(log_n), C.c_int32(op_kind), C.c_uint32(tree_height), C.byref(h)))
        return R1cs(self, h)

    def create_note_r1cs(self, log_n):
        h = C.c_void_p()
        self._chk(self.lib.zkmi_create_note_r1cs(C.c_uint32(log_n), C.byref(h)))
        return R1cs(self, h)

    def note_create(self, tokens, note):
        """Integers in: tokens = (token_0, token_1), note = (zk_id, trapdoor, nullifier)."""
        i = NoteCreate()
        for k in range(2):
            C.memmove(i.tokens[k], int(tokens[k]).to_bytes(32, "little"), 32)
        for k in range(3):
            C.memmove(i.note[k], int(note[k]).to_bytes(32, "little"), 32)
        return i

    def create_note_witness(self, log_n, inp):
        """(z bytes, [h_note_new, token_0, token_1] as ints)."""
        out = (C.c_uint8 * (32 << log_n))()
        pub = (C.c_uint8 * 96)()
        self._chk(self.lib.zkmi_create_note_witness(C.c_uint32(log_n), C.byref(inp), out, pub))
        p = bytes(pub)
        return bytes(out), [int.from_bytes(p[32 * k : 32 * k + 32], "little") for k in range(3)]

    def g1_in_subgroup(self, a):
        return self.lib.zkmi_g1_in_subgroup(_buf(a)) == 0

    def g2_in_subgroup(self, a):
        return self.lib.zkmi_g2_in_subgroup(_buf(a)) == 0

    # ---- keys in arkworks' CanonicalSerialize layout --------------------------
    def ark_vk_read(self, buf, compressed):
        cap = 672 + 96 * 64
        out = (C.c_uint8 * cap)()
        n_pub, used = C.c_uint32(), C.c_uint64()
        self._chk(self.lib.zkmi_ark_vk_read(_buf(buf), C.c_uint64(len(buf)), C.c_int32(int(compressed)), out, C.c_uint64(cap),
                                            C.byref(n_pub), C.byref(used)))
        return bytes(out)[: 672 + 96 * n_pub.value], n_pub.value, used.value

    def ark_vk_write(self, vk, n_pub, compressed):
        need = C.c_uint64()
        self.lib.zkmi_ark_vk_write(_buf(vk), C.c_uint32(n_pub), C.c_int32(int(compressed)), None, C.c_uint64(0), C.byref(need))
        out = (C.c_uint8 * need.value)()
        self._chk(self.lib.zkmi_ark_vk_write(_buf(vk), C.c_uint32(n_pub), C.c_int32(int(compressed)), out, need, C.byref(need)))
        return bytes(out)

    # ---- the ZkProof surface with real proofs ---------------
    def shielder_verify_creation(self, vk_create, h_note_new, tokens, proof):
        arr = (Scalar * TOKENS_NUMBER)(*[scalar(t) for t in tokens])
        h = scalar(h_note_new)
        self._chk(self.lib.zkmi_shielder_verify_creation(_buf(vk_create), C.byref(h), arr, _buf(proof)))

    def shielder_verify_update(self, vk_deposit, vk_withdraw, op_pub, h_note_new, merkle_root, nullifier_old, proof):
        a = [scalar(x) for x in (h_note_new, merkle_root, nullifier_old)]
        vd = _buf(vk_deposit) if vk_deposit else None
        vw = _buf(vk_withdraw) if vk_withdraw else None
        self._chk(self.lib.zkmi_shielder_verify_update(vd, vw, C.byref(op_pub), *[C.byref(x) for x in a], _buf(proof)))

    def note_update(self, amount, token, user, new_note, old_note, path_shape, path, op_priv_user, account):
        """Integers in, zkmi_note_update out (new_note / old_note = (zk_id, trapdoor, nullifier),
        account = (token_0, balance_0, token_1, balance_1))."""
        i = NoteUpdate()
        put = lambda dst, v: C.memmove(dst, int(v).to_bytes(32, "little"), 32)
        put(i.amount, amount), put(i.token, token), put(i.user, user), put(i.op_priv_user, op_priv_user)
        for k in range(3):
            put(i.new_note[k], new_note[k]), put(i.old_note[k], old_note[k])
        height = len(path)
        assert len(path_shape) == height and 1 <= height <= MAX_TREE_HEIGHT
        i.tree_height = height
        for k in range(height):
            i.path_shape[k] = int(path_shape[k])
            put(i.path[k], path[k])
        for k in range(4):
            put(i.account[k], account[k])
        return i

    def update_note_witness(self, log_n, op_kind, inp, check=True):
        """(z bytes, [6 public ints], status); raises on a non-zero status unless check=False."""
        out = (C.c_uint8 * (32 << log_n))()
        pub = (C.c_uint8 * (32 * 6))()
        rc = self.lib.zkmi_update_note_witness(C.c_uint32(log_n), C.c_int32(op_kind), C.byref(inp), out, pub)
        if check or rc in (-1, -2):
            self._chk(rc)
        p = bytes(pub)
        return bytes(out), [int.from_bytes(p[32 * k : 32 * k + 32], "little") for k in range(6)], rc

    def update_note_witness_values_host(self, log_n, op_kind, inp):
        """The device code path (value-only synthesis) executed on the host; test hook."""
        out = (C.c_uint8 * (32 << log_n))()
        rc = self.lib.zkmi_update_note_witness_values_host(C.c_uint32(log_n), C.c_int32(op_kind), C.byref(inp), out)
        if rc in (-1, -2):
            self._chk(rc)
        return bytes(out), rc

    def fr_reduce(self, b32):
        out = (C.c_uint8 * 32)()
        self._chk(self.lib.zkmi_fr_reduce(_buf(b32), out))
        return bytes(out)

    def update_note_input(self, amount, token, user, old_nullifier, new_note, old_trapdoor, old_account_hash,
                          path_shape, path, old_account):
        """All scalar arguments are 32-byte canonical LE values (use fr_reduce for SHA-256 outputs)."""
        i = UpdateNoteInput()
        i.amount, i.token, i.user, i.old_nullifier = scalar(amount), scalar(token), scalar(user), scalar(old_nullifier)
        for k in range(4):
            i.new_note[k] = scalar(new_note[k])
        i.old_trapdoor, i.old_account_hash = scalar(old_trapdoor), scalar(old_account_hash)
        for k in range(MERKLE_TREE_DEPTH):
            i.path_shape[k] = int(path_shape[k])
            i.path[k] = scalar(path[k])
        for k in range(TOKENS_NUMBER):
            i.old_account[k] = scalar(old_account[k])
        return i

    def shielder_witness_from_input(self, log_n, inp):
        out = (C.c_uint8 * (32 << log_n))()
        self._chk(self.tlib.zkmi_shielder_witness_from_input(C.c_uint32(log_n), C.byref(inp), out))
        return bytes(out)

    def r1cs_create(self, n_vars, n_pub, mats):
        """mats = [(rowptr list, col list, val bytes)] * 3"""
        args = []
        keep = []
        nc = len(mats[0][0]) - 1
        for rowptr, col, val in mats:
            rp = (C.c_uint32 * len(rowptr))(*rowptr)
            cl = (C.c_uint32 * max(1, len(col)))(*col)
            vl = _buf(val)
            keep += [rp, cl, vl]
            args += [rp, cl, vl]
        h = C.c_void_p()
        self._chk(self.lib.zkmi_r1cs_create(C.c_uint32(n_vars), C.c_uint32(n_pub), C.c_uint32(nc), *args, C.byref(h)))
        return R1cs(self, h)

    # ---- reference prove/verify surface (mocked_zk mirror) -------------------
    def account_new(self, tokens):
        arr = (Scalar * TOKENS_NUMBER)(*[scalar(t) for t in tokens])
        acc = Account()
        self._chk(self.lib.zkmi_account_new(arr, C.byref(acc)))
        return acc

    def account_hash(self, acc):
        out = Scalar()
        self._chk(self.lib.zkmi_account_hash(C.byref(acc), C.byref(out)))
        return bytes(out.bytes)

    def account_update(self, acc, op_pub, op_priv):
        out = Account()
        self._chk(self.lib.zkmi_account_update(C.byref(acc), C.byref(op_pub), C.byref(op_priv), C.byref(out)))
        return out

    def note_hash(self, id_, trapdoor, nullifier, acc_hash):
        out = Scalar()
        a = [scalar(x) for x in (id_, trapdoor, nullifier, acc_hash)]
        self._chk(self.lib.zkmi_note_hash(*[C.byref(x) for x in a], C.byref(out)))
        return bytes(out.bytes)

    def combine_merkle_hash(self, first, second):
        out = Scalar()
        a, b = scalar(first), scalar(second)
        self._chk(self.lib.zkmi_combine_merkle_hash(C.byref(a), C.byref(b), C.byref(out)))
        return bytes(out.bytes)

    def op_pub(self, kind, amount, token, user):
        o = OpPub()
        o.kind = {"deposit": 0, "withdraw": 1}[kind]
        C.memmove(o.amount, int(amount).to_bytes(16, "little"), 16)
        o.token = scalar(token)
        o.user = scalar(user)
        return o

    def op_priv(self, user):
        o = OpPriv()
        o.user = scalar(user)
        return o

    def operation_combine(self, op_pub, op_priv):
        self._chk(self.lib.zkmi_operation_combine(C.byref(op_pub), C.byref(op_priv)))

    def zkproof_new(self, id_, trapdoor, nullifier, op_priv, acc):
        out = ZkProof()
        a = [scalar(x) for x in (id_, trapdoor, nullifier)]
        self._chk(self.lib.zkmi_zkproof_new(*[C.byref(x) for x in a], C.byref(op_priv), C.byref(acc), C.byref(out)))
        return out

    def zkproof_update_account(self, proof, op_pub, op_priv, trapdoor, nullifier, merkle_proof, leaf_id):
        out = ZkProof()
        h = Scalar()
        mp = (Scalar * MERKLE_TREE_DEPTH)(*[scalar(x) for x in merkle_proof])
        t, n = scalar(trapdoor), scalar(nullifier)
        self._chk(
            self.lib.zkmi_zkproof_update_account(
                C.byref(proof), C.byref(op_pub), C.byref(op_priv), C.byref(t), C.byref(n), mp, C.c_uint32(leaf_id), C.byref(h), C.byref(out)
            )
        )
        return bytes(h.bytes), out

    def zkproof_verify_creation(self, proof, h_note_new, tokens):
        arr = (Scalar * TOKENS_NUMBER)(*[scalar(t) for t in tokens])
        h = scalar(h_note_new)
        self._chk(self.lib.zkmi_zkproof_verify_creation(C.byref(proof), C.byref(h), arr))

    def zkproof_verify_update(self, proof, op_pub, h_note_new, merkle_root, nullifier_old):
        a = [scalar(x) for x in (h_note_new, merkle_root, nullifier_old)]
        self._chk(self.lib.zkmi_zkproof_verify_update(C.byref(proof), C.byref(op_pub), *[C.byref(x) for x in a]))

    def context(self, device=0):
        return Context(self, device)


class R1cs:
    def __init__(self, z, handle):
        self.z, self.h = z, handle
        nv, npub, nc, lg = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        z._chk(z.lib.zkmi_r1cs_shape(handle, C.byref(nv), C.byref(npub), C.byref(nc), C.byref(lg)))
        self.n_vars, self.n_pub, self.n_constraints, self.log_n = nv.value, npub.value, nc.value, lg.value

    def export(self, m):
        nnz = C.c_uint64()
        self.z._chk(self.z.lib.zkmi_r1cs_export(self.h, C.c_int32(m), None, None, None, C.byref(nnz)))
        rp = (C.c_uint32 * (self.n_constraints + 1))()
        cl = (C.c_uint32 * max(1, nnz.value))()
        vl = (C.c_uint8 * max(1, 32 * nnz.value))()
        self.z._chk(self.z.lib.zkmi_r1cs_export(self.h, C.c_int32(m), rp, cl, vl, None))
        return list(rp), list(cl)[: nnz.value], bytes(vl)[: 32 * nnz.value]

    def is_satisfied(self, zbytes):
        rc = self.z.lib.zkmi_r1cs_is_satisfied(self.h, _buf(zbytes))
        if rc == 0:
            return True
        if rc == -8:
            return False
        raise ZkmiError(rc)

    def free(self):
        if self.h:
            self.z.lib.zkmi_r1cs_free(self.h)
            self.h = None


class Comm:
    """zkmi_comm: an RCCL communicator behind the C ABI (include/zkmi.h)."""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def free(self):
        if self.h:
            self.ctx.lib.zkmi_comm_destroy(self.h)
            self.h = None


class Context:
    """One HIP device + stream + resident workspaces (zkmi_ctx)."""

    def __init__(self, z, device=0):
        self.z = z
        self.lib = z.lib
        self.h = C.c_void_p()
        z._chk(self.lib.zkmi_ctx_create(C.c_int32(device), C.byref(self.h)))

    def _chk(self, rc):
        self.z._chk(rc, self.h)

    def close(self):
        if self.h:
            self.lib.zkmi_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        self._chk(self.lib.zkmi_ctx_sync(self.h))

    def prof_enable(self, on=True):
        self._chk(self.lib.zkmi_prof_enable(self.h, C.c_int32(1 if on else 0)))

    def prof_reset(self):
        self._chk(self.lib.zkmi_prof_reset(self.h))

    def prof_get(self, phase):
        ms, cnt = C.c_double(), C.c_uint64()
        self._chk(self.lib.zkmi_prof_get(self.h, C.c_int32(PHASES[phase]), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    # NTT
    def ntt(self, data, log_n, inverse=False, coset=False):
        assert len(data) == 32 << log_n
        buf = _buf(data)
        self._chk(self.lib.zkmi_ntt_fr(self.h, buf, C.c_uint32(log_n), C.c_int32(inverse), C.c_int32(coset)))
        return bytes(buf)

    def ntt_dev(self, dptr, log_n, inverse=False, coset=False):
        self._chk(self.lib.zkmi_ntt_fr_dev(self.h, C.c_void_p(dptr), C.c_uint32(log_n), C.c_int32(inverse), C.c_int32(coset)))

    # bases
    def bases_g1(self, affine, check=True):
        h = C.c_void_p()
        self._chk(self.lib.zkmi_bases_g1_load(self.h, _buf(affine), C.c_uint64(len(affine) // 96), C.c_int32(check), C.byref(h)))
        return Bases(self, h, 1, len(affine) // 96)

    def bases_g2(self, affine, check=True):
        h = C.c_void_p()
        self._chk(self.lib.zkmi_bases_g2_load(self.h, _buf(affine), C.c_uint64(len(affine) // 192), C.c_int32(check), C.byref(h)))
        return Bases(self, h, 2, len(affine) // 192)

    def bases_g1_synthetic_range(self, first, n):
        h = C.c_void_p()
        self._chk(self.z.tlib.zkmi_bases_g1_synthetic_range(self.h, C.c_uint64(first), C.c_uint64(n), C.byref(h)))
        return Bases(self, h, 1, n)

    def bases_g1_synthetic(self, n):
        h = C.c_void_p()
        self._chk(self.z.tlib.zkmi_bases_g1_synthetic(self.h, C.c_uint64(n), C.byref(h)))
        return Bases(self, h, 1, n)

    def bases_g2_synthetic(self, n):
        h = C.c_void_p()
        self._chk(self.z.tlib.zkmi_bases_g2_synthetic(self.h, C.c_uint64(n), C.byref(h)))
        return Bases(self, h, 2, n)

    # MSM
    def msm_g1(self, scalars, bases):
        out = (C.c_uint8 * 96)()
        self._chk(self.lib.zkmi_msm_g1(self.h, _buf(scalars), C.c_uint64(len(scalars) // 32), bases.h, out))
        return bytes(out)

    def msm_g2(self, scalars, bases):
        out = (C.c_uint8 * 192)()
        self._chk(self.lib.zkmi_msm_g2(self.h, _buf(scalars), C.c_uint64(len(scalars) // 32), bases.h, out))
        return bytes(out)

    def msm_g1_dev(self, dptr, n, bases):
        out = (C.c_uint8 * 96)()
        self._chk(self.lib.zkmi_msm_g1_dev(self.h, C.c_void_p(dptr), C.c_uint64(n), bases.h, out))
        return bytes(out)

    def msm_g2_dev(self, dptr, n, bases):
        out = (C.c_uint8 * 192)()
        self._chk(self.lib.zkmi_msm_g2_dev(self.h, C.c_void_p(dptr), C.c_uint64(n), bases.h, out))
        return bytes(out)

    def update_note_witness_batch_dev(self, log_n, op_kind, inputs, d_ptrs):
        """Assignments of len(inputs) instances generated on the GPU into the device buffers d_ptrs
        (each 2^log_n x 32 B); returns the per-instance status codes."""
        n = len(inputs)
        assert n == len(d_ptrs)
        arr = (NoteUpdate * max(1, n))(*inputs)
        ptrs = (C.c_void_p * max(1, n))(*[int(p) for p in d_ptrs])
        st = (C.c_int32 * max(1, n))()
        self._chk(self.lib.zkmi_update_note_witness_batch_dev(self.h, C.c_uint32(log_n), C.c_int32(op_kind), arr, C.c_uint32(n), ptrs, st))
        return list(st)[:n]

    # ---- BN254 MSM / NTT / KZG commit -----------------------
    def bn254_bases(self, affine, check=True):
        h = C.c_void_p()
        self._chk(self.lib.zkmi_bn254_bases_load(self.h, _buf(affine), C.c_uint64(len(affine) // 64), C.c_int32(check), C.byref(h)))
        return BnBases(self, h, len(affine) // 64)

    def bn254_bases_synthetic(self, n):
        h = C.c_void_p()
        self._chk(self.z.tlib.zkmi_bn254_bases_synthetic(self.h, C.c_uint64(n), C.byref(h)))
        return BnBases(self, h, n)

    def bn254_msm_g1(self, scalars, bases):
        out = (C.c_uint8 * 64)()
        self._chk(self.lib.zkmi_bn254_msm_g1(self.h, _buf(scalars), C.c_uint64(len(scalars) // 32), bases.h, out))
        return bytes(out)

    def bn254_msm_g1_dev(self, dptr, n, bases):
        out = (C.c_uint8 * 64)()
        self._chk(self.lib.zkmi_bn254_msm_g1_dev(self.h, C.c_void_p(dptr), C.c_uint64(n), bases.h, out))
        return bytes(out)

    def bn254_ntt(self, data, log_n, inverse=False, coset=False):
        buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
        self._chk(self.lib.zkmi_bn254_ntt_fr(self.h, buf, C.c_uint32(log_n), C.c_int32(inverse), C.c_int32(coset)))
        return bytes(buf)

    def bn254_ntt_dev(self, dptr, log_n, inverse=False, coset=False):
        self._chk(self.lib.zkmi_bn254_ntt_fr_dev(self.h, C.c_void_p(dptr), C.c_uint32(log_n), C.c_int32(inverse), C.c_int32(coset)))

    def bn254_kzg_commit_dev(self, d_evals, log_n, srs):
        out = (C.c_uint8 * 64)()
        self._chk(self.lib.zkmi_bn254_kzg_commit_dev(self.h, C.c_void_p(d_evals), C.c_uint32(log_n), srs.h, out))
        return bytes(out)

    def bn254_kzg_open_dev(self, d_coeffs, n, zeta, srs, d_quotient=None):
        """(p(zeta) as 32 bytes, commit(q) as 64 bytes); d_quotient: device pointer that receives the n - 1 quotient coefficients."""
        ev, pf = (C.c_uint8 * 32)(), (C.c_uint8 * 64)()
        self._chk(self.lib.zkmi_bn254_kzg_open_dev(self.h, C.c_void_p(d_coeffs), C.c_uint64(n), _buf(zeta), srs.h,
                                                   C.c_void_p(d_quotient) if d_quotient else None, ev, pf))
        return bytes(ev), bytes(pf)

    def bn254_kzg_open_many_dev(self, d_polys, n, zeta, v, srs):
        """([p_j(zeta) as 32 bytes], commit(q of sum_j v^j p_j) as 64 bytes) for a list of device pointers."""
        k = len(d_polys)
        tab = (C.c_void_p * max(k, 1))(*d_polys)
        ev, pf = (C.c_uint8 * (32 * max(k, 1)))(), (C.c_uint8 * 64)()
        self._chk(self.lib.zkmi_bn254_kzg_open_many_dev(self.h, tab, C.c_uint32(k), C.c_uint64(n), _buf(zeta), _buf(v), srs.h, ev, pf))
        raw = bytes(ev)
        return [raw[32 * j: 32 * j + 32] for j in range(k)], bytes(pf)

    def bn254_grand_product_dev(self, d_num, d_den, n, d_out):
        """d_out[i] = prod_{j < i} num_j / den_j; returns the product over all n terms (32 bytes)."""
        tot = (C.c_uint8 * 32)()
        self._chk(self.lib.zkmi_bn254_grand_product_dev(self.h, C.c_void_p(d_num), C.c_void_p(d_den), C.c_uint64(n), C.c_void_p(d_out), tot))
        return bytes(tot)

    def sha256_pairs(self, inputs, n_hashes):
        assert len(inputs) == 64 * n_hashes
        out = (C.c_uint8 * (32 * max(1, n_hashes)))()
        self._chk(self.lib.zkmi_sha256_pairs(self.h, _buf(inputs), C.c_uint64(n_hashes), out))
        return bytes(out)[: 32 * n_hashes]

    def sha256_pairs_dev(self, d_in, n_hashes, d_out):
        self._chk(self.lib.zkmi_sha256_pairs_dev(self.h, C.c_void_p(d_in), C.c_uint64(n_hashes), C.c_void_p(d_out)))

    def sha256_merkle_tree_dev(self, d_nodes, log_leaves, n_filled):
        self._chk(self.lib.zkmi_sha256_merkle_tree_dev(self.h, C.c_void_p(d_nodes), C.c_uint32(log_leaves), C.c_uint64(n_filled)))

    def poseidon_hash_batch(self, inputs, n_hashes, arity, field=0):
        """n_hashes x arity canonical 32-byte inputs (bytes) -> n_hashes x 32 bytes, hashed on the GPU."""
        assert len(inputs) == 32 * n_hashes * arity
        out = (C.c_uint8 * (32 * max(1, n_hashes)))()
        self._chk(self.lib.zkmi_poseidon_hash_batch(self.h, C.c_int32(field), _buf(inputs), C.c_uint64(n_hashes), C.c_uint32(arity), out))
        return bytes(out)[: 32 * n_hashes]

    def poseidon_hash_batch_dev(self, d_in, n_hashes, arity, d_out, field=0):
        self._chk(self.lib.zkmi_poseidon_hash_batch_dev(self.h, C.c_int32(field), C.c_void_p(d_in), C.c_uint64(n_hashes), C.c_uint32(arity), C.c_void_p(d_out)))

    def poseidon_merkle_tree_dev(self, d_nodes, log_leaves, field=0):
        self._chk(self.lib.zkmi_poseidon_merkle_tree_dev(self.h, C.c_int32(field), C.c_void_p(d_nodes), C.c_uint32(log_leaves)))

    def poseidon_merkle_paths_dev(self, d_nodes, log_leaves, leaf_idx):
        """(shape bytes n x log_leaves, path bytes n x log_leaves x 32) for the given leaf indices."""
        n = len(leaf_idx)
        idx = (C.c_uint32 * max(1, n))(*leaf_idx)
        shape = (C.c_uint8 * max(1, n * log_leaves))()
        paths = (C.c_uint8 * max(1, 32 * n * log_leaves))()
        self._chk(self.lib.zkmi_poseidon_merkle_paths_dev(self.h, C.c_void_p(d_nodes), C.c_uint32(log_leaves), idx, C.c_uint32(n), shape, paths))
        return bytes(shape)[: n * log_leaves], bytes(paths)[: 32 * n * log_leaves]

    def poseidon_merkle_roots_dev(self, d_leaves, d_shape, d_paths, depth, n, d_roots, field=0):
        self._chk(self.lib.zkmi_poseidon_merkle_roots_dev(self.h, C.c_int32(field), C.c_void_p(d_leaves), C.c_void_p(d_shape), C.c_void_p(d_paths), C.c_uint32(depth), C.c_uint64(n), C.c_void_p(d_roots)))

    def comm_init(self, n_ranks, rank, unique_id):
        """RCCL communicator of this context's device (collective: every rank calls it with rank 0's id)."""
        h = C.c_void_p()
        self._chk(self.lib.zkmi_comm_init(self.h, C.c_uint32(n_ranks), C.c_uint32(rank), _buf(unique_id), C.byref(h)))
        return Comm(self, h)

    def msm_g1_allgather_combine(self, comm, dptr, n, bases, plan_n):
        """This rank's slice of a point-split MSM + the RCCL exchange + combination (collective); the full result."""
        out = (C.c_uint8 * 96)()
        self._chk(self.lib.zkmi_msm_g1_allgather_combine(self.h, comm.h, C.c_void_p(dptr), C.c_uint64(n), bases.h, C.c_uint64(plan_n), out))
        return bytes(out)

    def selftest_msm_g1_sum2_dev(self, dptr_a, dptr_b, n, bases):
        """MSM(a) + MSM(b) over the same bases through ONE shared bucket set (test hook of the L + H merge)."""
        out = (C.c_uint8 * 96)()
        self._chk(self.z.tlib.zkmi_selftest_msm_g1_sum2_dev(self.h, C.c_void_p(dptr_a), C.c_void_p(dptr_b), C.c_uint64(n), bases.h, out))
        return bytes(out)

    def msm_g1_window_range_dev(self, dptr, n, bases, plan_n, w_first, w_count):
        """(w_count x 96 B window sums over ALL n points, total windows of the plan, window bits): a rank of a WINDOW split"""
        out = (C.c_uint8 * (96 * max(1, w_count)))()
        tot, cbits = C.c_uint32(), C.c_uint32()
        self._chk(self.lib.zkmi_msm_g1_window_range_dev(self.h, C.c_void_p(dptr), C.c_uint64(n), bases.h, C.c_uint64(plan_n),
                                                        C.c_uint32(w_first), C.c_uint32(w_count), out, C.byref(tot), C.byref(cbits)))
        return bytes(out)[: 96 * w_count], tot.value, cbits.value

    def msm_g1_window_split_allgather(self, comm, dptr, n, bases):
        """BASELINE configs[3] as worded: windows split over the ranks of `comm`, RCCL all-gather, the full result (collective)"""
        out = (C.c_uint8 * 96)()
        self._chk(self.lib.zkmi_msm_g1_window_split_allgather(self.h, comm.h, C.c_void_p(dptr), C.c_uint64(n), bases.h, out))
        return bytes(out)

    def msm_g1_split2d_allgather(self, comm, dptr, n, bases, plan_n, window_groups):
        """The 2-D split: rank k = g * window_groups + q holds point group g's slice (dptr / n / bases) and computes window
        range q of the plan of plan_n terms; RCCL all-gather, the full result on every rank (collective)"""
        out = (C.c_uint8 * 96)()
        self._chk(self.lib.zkmi_msm_g1_split2d_allgather(self.h, comm.h, C.c_void_p(dptr), C.c_uint64(n), bases.h, C.c_uint64(plan_n),
                                                         C.c_uint32(window_groups), out))
        return bytes(out)

    def msm_g1_windows_dev(self, dptr, n, bases, plan_n):
        out = (C.c_uint8 * (96 * 64))()
        nwin, cbits = C.c_uint32(), C.c_uint32()
        self._chk(
            self.lib.zkmi_msm_g1_windows_dev(self.h, C.c_void_p(dptr), C.c_uint64(n), bases.h, C.c_uint64(plan_n), out, C.byref(nwin), C.byref(cbits))
        )
        return bytes(out)[: 96 * nwin.value], nwin.value, cbits.value

    # Groth16
    def set_group_size(self, group):
        """proofs per group for keys created next on this context (0 = automatic, 1 = never group)"""
        self._chk(self.lib.zkmi_ctx_set_group_size(self.h, C.c_uint32(group)))

    def groth16_setup(self, r1cs, toxic):
        assert len(toxic) == 160
        cap = 672 + 96 * r1cs.n_pub
        vk = (C.c_uint8 * cap)()
        h = C.c_void_p()
        self._chk(self.lib.zkmi_groth16_setup(self.h, r1cs.h, _buf(toxic), C.byref(h), vk, C.c_uint64(cap)))
        return ProvingKey(self, h, r1cs), bytes(vk)

    def pk_load(self, r1cs, alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2, a_q, b1_q, b2_q, h_q, l_q):
        h = C.c_void_p()
        self._chk(
            self.lib.zkmi_pk_load(
                self.h, r1cs.h, _buf(alpha_g1), _buf(beta_g1), _buf(beta_g2), _buf(delta_g1), _buf(delta_g2),
                _buf(a_q), _buf(b1_q), _buf(b2_q), _buf(h_q), _buf(l_q), C.byref(h),
            )
        )
        return ProvingKey(self, h, r1cs)

    def shielder_prove_creation(self, pk_create, knowledge, tokens, r, s):
        """(h_note_new bytes, proof bytes) from the ZkProof `knowledge` built by zkproof_new."""
        arr = (Scalar * TOKENS_NUMBER)(*[scalar(t) for t in tokens])
        h = Scalar()
        out = (C.c_uint8 * 192)()
        self._chk(self.z.lib.zkmi_shielder_prove_creation(self.h, pk_create.h, C.byref(knowledge), arr, _buf(r), _buf(s), C.byref(h), out))
        return bytes(h.bytes), bytes(out)

    def shielder_prove_update(self, pk_deposit, pk_withdraw, knowledge, op_pub, op_priv, trapdoor, nullifier, merkle_proof,
                              leaf_id, r, s):
        """ZkProof::update_account with a real proof: (h_note_new, merkle_root, new ZkProof, proof bytes)."""
        height = len(merkle_proof)
        mp = (Scalar * height)(*[scalar(x) for x in merkle_proof])
        t, n = scalar(trapdoor), scalar(nullifier)
        h, root, new = Scalar(), Scalar(), ZkProof()
        out = (C.c_uint8 * 192)()
        self._chk(self.z.lib.zkmi_shielder_prove_update(
            self.h, pk_deposit.h if pk_deposit else None, pk_withdraw.h if pk_withdraw else None, C.byref(knowledge),
            C.byref(op_pub), C.byref(op_priv), C.byref(t), C.byref(n), mp, C.c_uint32(height), C.c_uint32(leaf_id),
            _buf(r), _buf(s), C.byref(h), C.byref(root), C.byref(new), out))
        return bytes(h.bytes), bytes(root.bytes), new, bytes(out)

    def ark_pk_load(self, r1cs, buf, compressed, check_curve=True):
        """Proving key from arkworks' ProvingKey::serialize_{compressed,uncompressed} bytes -> (ProvingKey, vk bytes)."""
        h = C.c_void_p()
        cap = 672 + 96 * r1cs.n_pub
        vk = (C.c_uint8 * cap)()
        self._chk(self.lib.zkmi_ark_pk_load(self.h, r1cs.h, _buf(buf), C.c_uint64(len(buf)), C.c_int32(int(compressed)),
                                            C.c_int32(int(check_curve)), C.byref(h), vk, C.c_uint64(cap)))
        return ProvingKey(self, h, r1cs), bytes(vk)

    def ark_pk_write(self, pk, vk, compressed):
        need = C.c_uint64()
        self.lib.zkmi_ark_pk_write(self.h, pk.h, _buf(vk), C.c_int32(int(compressed)), None, C.c_uint64(0), C.byref(need))
        out = (C.c_uint8 * need.value)()
        self._chk(self.lib.zkmi_ark_pk_write(self.h, pk.h, _buf(vk), C.c_int32(int(compressed)), out, need, C.byref(need)))
        return bytes(out)

    def groth16_prove(self, pk, z, r, s):
        out = (C.c_uint8 * 192)()
        self._chk(self.lib.zkmi_groth16_prove(self.h, pk.h, _buf(z), _buf(r), _buf(s), out))
        return bytes(out)

    def groth16_prove_dev(self, pk, d_z_ptr, r, s):
        out = (C.c_uint8 * 192)()
        self._chk(self.lib.zkmi_groth16_prove_dev(self.h, pk.h, C.c_void_p(d_z_ptr), _buf(r), _buf(s), out))
        return bytes(out)

    def groth16_prove_batch_dev(self, pk, d_z_ptrs, rs, ss):
        n = len(d_z_ptrs)
        ptrs = (C.c_void_p * n)(*d_z_ptrs)
        out = (C.c_uint8 * (192 * n))()
        self._chk(self.lib.zkmi_groth16_prove_batch_dev(self.h, pk.h, C.c_uint32(n), ptrs, _buf(b"".join(rs)), _buf(b"".join(ss)), out))
        raw = bytes(out)
        return [raw[192 * i : 192 * i + 192] for i in range(n)]

    def groth16_prove_batch_host(self, pk, z_ptrs, rs, ss):
        """z_ptrs: HOST addresses of the witnesses (pinned for an asynchronous upload)."""
        n = len(z_ptrs)
        ptrs = (C.c_void_p * n)(*z_ptrs)
        out = (C.c_uint8 * (192 * n))()
        self._chk(self.lib.zkmi_groth16_prove_batch(self.h, pk.h, C.c_uint32(n), ptrs, _buf(b"".join(rs)), _buf(b"".join(ss)), out))
        raw = bytes(out)
        return [raw[192 * i : 192 * i + 192] for i in range(n)]

    def groth16_witness_map(self, pk, z):
        out = (C.c_uint8 * (32 << pk.r1cs.log_n))()
        self._chk(self.lib.zkmi_groth16_witness_map(self.h, pk.h, _buf(z), out))
        return bytes(out)


class BnBases:
    def __init__(self, ctx, h, n):
        self.ctx, self.h, self.n = ctx, h, n

    def prepare(self):
        """Fixed-base table for repeated commitments against this SRS (zkmi_bn254_srs_prepare)."""
        self.ctx._chk(self.ctx.lib.zkmi_bn254_srs_prepare(self.ctx.h, self.h))
        return self

    def read(self, first, count):
        out = (C.c_uint8 * (64 * max(1, count)))()
        self.ctx._chk(self.ctx.lib.zkmi_bn254_bases_read(self.ctx.h, self.h, C.c_uint64(first), C.c_uint64(count), out))
        return bytes(out)[: 64 * count]

    def free(self):
        if self.h:
            self.ctx.lib.zkmi_bn254_bases_free(self.h)
            self.h = None


class Bases:
    def __init__(self, ctx, h, group, n):
        self.ctx, self.h, self.group, self.n = ctx, h, group, n

    def read(self, first, count):
        w = 96 if self.group == 1 else 192
        out = (C.c_uint8 * (w * count))()
        fn = self.ctx.lib.zkmi_bases_g1_read if self.group == 1 else self.ctx.lib.zkmi_bases_g2_read
        self.ctx._chk(fn(self.ctx.h, self.h, C.c_uint64(first), C.c_uint64(count), out))
        return bytes(out)

    def prepare(self):
        """Table of 2^(c w) multiples for repeated full-length MSMs over these bases (zkmi_bases_g{1,2}_prepare)."""
        fn = self.ctx.lib.zkmi_bases_g1_prepare if self.group == 1 else self.ctx.lib.zkmi_bases_g2_prepare
        self.ctx._chk(fn(self.ctx.h, self.h))
        return self

    def free(self):
        if self.h:
            fn = self.ctx.lib.zkmi_bases_g1_free if self.group == 1 else self.ctx.lib.zkmi_bases_g2_free
            fn(self.h)
            self.h = None


class ProvingKey:
    def __init__(self, ctx, h, r1cs):
        self.ctx, self.h, self.r1cs = ctx, h, r1cs

    def export_query(self, which, first, count):
        w = 192 if which == 2 else 96
        out = (C.c_uint8 * (w * count))()
        self.ctx._chk(self.ctx.lib.zkmi_pk_export_query(self.ctx.h, self.h, C.c_int32(which), C.c_uint64(first), C.c_uint64(count), out))
        return bytes(out)

    def schedule_state(self):
        """zkmi_pk_schedule_state: (B1 folded into the L + H reduction?, non-zero digits of the last finished proof's
        assignment(s), digits a dense assignment would have placed)."""
        out = (C.c_uint64 * 3)()
        self.ctx._chk(self.ctx.lib.zkmi_pk_schedule_state(self.h, out))
        return bool(out[0]), int(out[1]), int(out[2])

    def free(self):
        if self.h:
            self.ctx.lib.zkmi_pk_free(self.h)
            self.h = None
