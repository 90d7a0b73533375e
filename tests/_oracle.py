"""ctypes bindings for the CPU checkers in oracle/ (test infrastructure only).

`oracle()` loads oracle/libgamdp_oracle.so (our plain-C restatement, travels to the GPU box);
`ref()` loads oracle/_ref/libgamref.so (the reference's own sources + shim; only exists where
/root/reference was present at build time) or returns None.
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

OK, EMPTY, OUT_OF_RANGE, INVALID = 0, 1, 2, 3
OPS = "ABMX"  # GAP_A, GAP_B, MATCH, MISMATCH


class OracleResult(C.Structure):
    _fields_ = [
        ("begin_a", C.c_uint64), ("begin_b", C.c_uint64), ("score", C.c_int64),
        ("n_match", C.c_uint64), ("length", C.c_uint64),
        ("first_a", C.c_uint64), ("first_b", C.c_uint64),
        ("last_a", C.c_uint64), ("last_b", C.c_uint64),
        ("cells", C.c_uint64), ("homology", C.c_double),
        ("first_found", C.c_uint8), ("last_found", C.c_uint8), ("status", C.c_uint8),
        ("pad_", C.c_uint8 * 5),
    ]

    def key(self):
        """Everything the reference's MyAlignment + first/last_match_pos expose."""
        return (self.status, self.begin_a, self.begin_b, self.score, self.n_match, self.length,
                self.first_a, self.first_b, self.first_found, self.last_a, self.last_b,
                self.last_found, self.homology)


class RefResult(C.Structure):
    _fields_ = [
        ("begin_a", C.c_uint64), ("begin_b", C.c_uint64), ("a_size", C.c_uint64),
        ("b_size", C.c_uint64), ("score", C.c_int64), ("homology", C.c_double),
        ("length", C.c_uint64), ("n_match", C.c_uint64),
        ("first_a", C.c_uint64), ("first_b", C.c_uint64),
        ("last_a", C.c_uint64), ("last_b", C.c_uint64),
        ("first_found", C.c_int32), ("last_found", C.c_int32), ("status", C.c_int32),
    ]


class OracleBlock(C.Structure):
    _fields_ = [("m_begin", C.c_int32), ("m_end", C.c_int32), ("s_begin", C.c_int32),
                ("s_end", C.c_int32), ("m_strand", C.c_char), ("s_strand", C.c_char),
                ("n_reads", C.c_int64)]


class OracleMB(C.Structure):
    _fields_ = [("m_ltail", C.c_uint8), ("m_rtail", C.c_uint8), ("s_ltail", C.c_uint8),
                ("s_rtail", C.c_uint8), ("align_ok", C.c_uint8), ("align_rev", C.c_uint8),
                ("status", C.c_uint8), ("touched", C.c_uint8),
                ("m_start", C.c_int32), ("m_end", C.c_int32), ("s_start", C.c_int32),
                ("s_end", C.c_int32), ("n_dp", C.c_uint32), ("cells", C.c_uint64)]


_oracle = None
_ref = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])


def oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "libgamdp_oracle.so")
        src = os.path.join(ORACLE_DIR, "gamdp_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
        lib = C.CDLL(path)
        u64, p8 = C.c_uint64, C.c_char_p
        lib.gamdp_oracle_align.argtypes = [p8, u64, p8, u64, u64, u64, u64, u64, u64, C.c_int, C.c_int,
                                           C.POINTER(OracleResult), C.c_void_p, u64]
        lib.gamdp_oracle_align.restype = C.c_int
        lib.gamdp_oracle_find_hits.argtypes = [p8, u64, u64, u64, p8, u64, u64, u64, u64, C.c_void_p, u64]
        lib.gamdp_oracle_find_hits.restype = C.c_int64
        lib.gamdp_oracle_encode.argtypes = [p8, u64, C.c_void_p]
        lib.gamdp_oracle_decode.argtypes = [C.c_void_p, u64, C.c_void_p]
        lib.gamdp_oracle_revcomp.argtypes = [C.c_void_p, u64]
        lib.gamdp_oracle_align_merge_block.argtypes = [p8, u64, p8, u64, C.POINTER(OracleBlock), C.c_uint32,
                                                       u64, C.POINTER(OracleMB), C.c_void_p, C.c_uint32]
        lib.gamdp_oracle_align_merge_block.restype = C.c_int
        lib.gamdp_oracle_synth_pair.argtypes = [u64, u64, C.c_void_p, C.c_void_p]
        lib.gamdp_oracle_synth_pair.restype = u64
        lib.gamdp_oracle_bench_pairs.argtypes = [u64, u64, u64, u64, C.c_int, C.c_void_p]
        lib.gamdp_oracle_bench_pairs.restype = u64
        _oracle = lib
    return _oracle


def ref():
    """The compiled reference, or None when it is not available (e.g. on the GPU box)."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libgamref.so")
        if not os.path.exists(path):
            if os.path.isdir("/root/reference/lib/src/alignment"):
                subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])
            else:
                return None
        lib = C.CDLL(path)
        u64, p8 = C.c_uint64, C.c_char_p
        lib.gamref_find_alignment.argtypes = [p8, u64, p8, u64, u64, u64, u64, u64, u64, C.c_int, C.c_int,
                                              C.POINTER(RefResult), C.c_void_p, u64]
        lib.gamref_find_alignment.restype = C.c_int
        lib.gamref_find_hits.argtypes = [p8, u64, u64, u64, p8, u64, u64, u64, u64, C.c_void_p, u64]
        lib.gamref_find_hits.restype = C.c_int64
        lib.gamref_reverse_complement.argtypes = [C.c_void_p, u64]
        lib.gamref_normalise.argtypes = [C.c_void_p, u64]
        _ref = lib
    return _ref


def encode(s):
    """ASCII bases (str/bytes) -> code bytes (A0 T1 C2 G3 N4) via the oracle's encoder."""
    if isinstance(s, str):
        s = s.encode()
    out = C.create_string_buffer(len(s) + 1)
    oracle().gamdp_oracle_encode(s, len(s), out)
    return out.raw[:len(s)]


def decode(codes):
    out = C.create_string_buffer(len(codes) + 1)
    oracle().gamdp_oracle_decode(codes, len(codes), out)
    return out.raw[:len(codes)].decode()


def oracle_align(a, b, band, begin_a, end_a, begin_b, end_b, fs=False, fe=False, want_ops=True):
    """a, b: code bytes. Returns (OracleResult, ops-string or None)."""
    r = OracleResult()
    cap = len(a) + len(b) + 2 * band + 64
    ops = C.create_string_buffer(cap) if want_ops else None
    oracle().gamdp_oracle_align(a, len(a), b, len(b), band, begin_a, end_a, begin_b, end_b, int(fs), int(fe),
                                C.byref(r), ops, cap if want_ops else 0)
    s = None
    if want_ops:
        s = "".join(OPS[c] for c in ops.raw[:r.length]) if r.status == OK else ""
    return r, s


def ref_align(a_chars, b_chars, band, begin_a, end_a, begin_b, end_b, fs=False, fe=False):
    """a_chars, b_chars: ASCII bytes. Returns (RefResult, ops-string)."""
    lib = ref()
    r = RefResult()
    cap = len(a_chars) + len(b_chars) + 2 * band + 64
    ops = C.create_string_buffer(cap)
    lib.gamref_find_alignment(a_chars, len(a_chars), b_chars, len(b_chars), band, begin_a, end_a, begin_b, end_b,
                              int(fs), int(fe), C.byref(r), ops, cap)
    s = "".join(OPS[c] for c in ops.raw[:r.length]) if r.status == 0 else ""
    return r, s


def ref_key(r):
    """RefResult -> the same tuple layout as OracleResult.key() (status mapped)."""
    if r.status == 2:
        return (OUT_OF_RANGE, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0)
    # the reference's MyAlignment() (all zero, empty ops) is our EMPTY
    st = EMPTY if (r.length == 0 and r.a_size == 0 and r.b_size == 0) else OK
    return (st, r.begin_a, r.begin_b, r.score, r.n_match, r.length, r.first_a, r.first_b, r.first_found,
            r.last_a, r.last_b, r.last_found, r.homology)


def oracle_find_hits(a, a_s, a_e, b, b_s, b_e, word=20):
    cap = len(a) + 1
    buf = (C.c_uint32 * cap)()
    n = oracle().gamdp_oracle_find_hits(a, len(a), a_s, a_e, b, len(b), b_s, b_e, word, buf, cap)
    return list(buf[:n])


def ref_find_hits(a_chars, a_s, a_e, b_chars, b_s, b_e, word=20):
    cap = len(a_chars) + 1
    buf = (C.c_uint32 * cap)()
    n = ref().gamref_find_hits(a_chars, len(a_chars), a_s, a_e, b_chars, len(b_chars), b_s, b_e, word, buf, cap)
    return list(buf[:n])
