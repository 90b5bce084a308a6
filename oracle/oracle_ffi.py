"""ctypes binding of the CPU ORACLE (oracle/awry_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under awry_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libawry_oracle.so")

NUCLEOTIDE, AMINO = 0, 1
PANIC = (1 << 64) - 1


def build(force=False):
    """Compile the oracle with gcc (a few seconds).  Needs no GPU."""
    if force or not os.path.exists(_SO) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
        for f in ("awry_oracle.c", "awry_oracle.h")
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


class Pos(C.Structure):
    _fields_ = [("seq_idx", C.c_uint64), ("local_pos", C.c_uint64)]


class Tally(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("queries", "steps", "block_reads", "backsteps", "hits")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None
u8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    vp, u64, u8, i32 = C.c_void_p, C.c_uint64, C.c_uint8, C.c_int

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype, f.argtypes = res, list(args)

    sig("orc_cardinality", u8, i32)
    for n in ("orc_ascii_to_index", "orc_ascii_to_code", "orc_index_to_code", "orc_code_to_index",
              "orc_index_to_ascii", "orc_code_to_ascii"):
        sig(n, u8, i32, u8)
    sig("orc_masked_popcount", C.c_uint32, u64p, u64)
    sig("orc_block_set_symbol", None, u64p, i32, u8, u64)
    sig("orc_block_code_at", u8, u64p, i32, u64)
    sig("orc_nt_block_occ", u64, u64p, u64p, u64, u8)
    sig("orc_aa_block_occ", u64, u64p, u64p, u64, u8)
    sig("orc_csa_bits_per_element", u64, u64)
    sig("orc_csa_word_len", u64, u64, u64)
    sig("orc_csa_set_value", None, u64p, u64, u64, u64)
    sig("orc_csa_reconstruct", i32, u64p, u64, u64, u64, u64p)
    sig("orc_suffix_array", i32, vp, u64, u64p)
    sig("orc_brute_count", u64, i32, vp, u64, vp, u64)
    sig("orc_brute_locate", u64, i32, vp, u64, vp, u64, u64p, u64)
    sig("orc_index_from_sa", vp, vp, u64, u64p, i32, u64, u8, u64p, C.POINTER(C.c_char_p), u64)
    sig("orc_index_build", vp, vp, u64, i32, u64, u8, u64p, C.POINTER(C.c_char_p), u64)
    sig("orc_index_from_fasta", vp, C.c_char_p, i32, u64, u8)
    sig("orc_index_free", None, vp)
    sig("orc_index_save", i32, vp, C.c_char_p)
    sig("orc_index_load", vp, C.c_char_p)
    sig("orc_alphabet", i32, vp)
    for n in ("orc_bwt_len", "orc_version", "orc_sa_ratio", "orc_num_sequences"):
        sig(n, u64, vp)
    sig("orc_kmer_len", u8, vp)
    for n in ("orc_prefix_sums", "orc_block_words", "orc_sa_words", "orc_kmer_table"):
        sig(n, u64p, vp, u64p)
    sig("orc_seq_start", u64, vp, u64)
    sig("orc_seq_header", C.c_char_p, vp, u64)
    sig("orc_text", u8p, vp)
    sig("orc_initial_range", None, vp, u8, u64p, u64p)
    sig("orc_update_range", None, vp, u64, u64, u8, u64p, u64p)
    sig("orc_backstep", u64, vp, u64)
    sig("orc_global_occurrence", u64, vp, u64, u8)
    sig("orc_symbol_at", u8, vp, u64)
    sig("orc_search_range", i32, vp, vp, u64, u64p, u64p, C.POINTER(Tally))
    sig("orc_count_string", i32, vp, vp, u64, u64p)
    sig("orc_locate_string", i32, vp, vp, u64, C.POINTER(u64p), C.POINTER(C.POINTER(Pos)), u64p, C.POINTER(Tally))
    sig("orc_free", None, vp)
    sig("orc_seq_location", None, vp, u64, C.POINTER(Pos))
    sig("orc_seq_location_ref", i32, vp, u64, C.POINTER(Pos))
    sig("orc_parallel_count", i32, vp, vp, u64p, u64, u64p, i32, C.POINTER(Tally))
    sig("orc_parallel_locate", i32, vp, vp, u64p, u64, C.POINTER(u64p), C.POINTER(u64p),
        C.POINTER(C.POINTER(Pos)), i32, C.POINTER(Tally))
    _lib = L
    return L


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(u64p)


def _bytes(b):
    if isinstance(b, str):
        b = b.encode("latin-1")
    if isinstance(b, (bytes, bytearray)):
        b = np.frombuffer(bytes(b), dtype=np.uint8)
    return np.ascontiguousarray(b, dtype=np.uint8)


def pack_queries(queries):
    """list of str/bytes -> (uint8 bytes, uint64 offsets[n+1])"""
    qs = [q.encode("latin-1") if isinstance(q, str) else bytes(q) for q in queries]
    off = np.zeros(len(qs) + 1, dtype=np.uint64)
    if qs:
        off[1:] = np.cumsum([len(q) for q in qs], dtype=np.uint64)
    return np.frombuffer(b"".join(qs), dtype=np.uint8).copy() if qs else np.zeros(0, np.uint8), off


def suffix_array(text):
    t = _bytes(text)
    sa = np.zeros(len(t), dtype=np.uint64)
    rc = lib().orc_suffix_array(t.ctypes.data, len(t), sa.ctypes.data_as(u64p))
    assert rc == 0
    return sa


class OracleIndex:
    """Mirror of the reference FmIndex surface (src/fm_index.rs) over the C oracle."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle index construction failed")
        self.h = C.c_void_p(handle)

    # ---- constructors
    @classmethod
    def from_text(cls, text, alphabet=NUCLEOTIDE, sa_ratio=8, kmer_len=0, seq_starts=(0,), headers=("seq0",), sa=None):
        t = _bytes(text)
        assert t[-1] == ord("$")
        st, stp = _u64(list(seq_starts))
        hd = (C.c_char_p * len(headers))(*[h.encode() for h in headers])
        if sa is None:
            h = lib().orc_index_build(t.ctypes.data, len(t), alphabet, sa_ratio, kmer_len, stp, hd, len(headers))
        else:
            s, sp = _u64(sa)
            h = lib().orc_index_from_sa(t.ctypes.data, len(t), sp, alphabet, sa_ratio, kmer_len, stp, hd, len(headers))
        return cls(h)

    @classmethod
    def from_fasta(cls, path, alphabet=NUCLEOTIDE, sa_ratio=8, kmer_len=0):
        return cls(lib().orc_index_from_fasta(os.fsencode(path), alphabet, sa_ratio, kmer_len))

    @classmethod
    def load(cls, path):
        return cls(lib().orc_index_load(os.fsencode(path)))

    def save(self, path):
        if lib().orc_index_save(self.h, os.fsencode(path)) != 0:
            raise IOError("oracle save failed")

    def close(self):
        if self.h:
            lib().orc_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- accessors
    def alphabet(self):
        return lib().orc_alphabet(self.h)

    def bwt_len(self):
        return int(lib().orc_bwt_len(self.h))

    def version_number(self):
        return int(lib().orc_version(self.h))

    def suffix_array_compression_ratio(self):
        return int(lib().orc_sa_ratio(self.h))

    def kmer_len(self):
        return int(lib().orc_kmer_len(self.h))

    def _arr(self, fn, mult=1):
        n = C.c_uint64()
        p = getattr(lib(), fn)(self.h, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(int(n.value) * mult,)).copy()

    def prefix_sums(self):
        return self._arr("orc_prefix_sums")

    def block_words(self):
        return self._arr("orc_block_words")

    def sa_words(self):
        return self._arr("orc_sa_words")

    def kmer_table(self):
        return self._arr("orc_kmer_table", 2).reshape(-1, 2)

    def sequences(self):
        n = int(lib().orc_num_sequences(self.h))
        return [(int(lib().orc_seq_start(self.h, i)), lib().orc_seq_header(self.h, i).decode()) for i in range(n)]

    def text(self):
        p = lib().orc_text(self.h)
        return bytes(np.ctypeslib.as_array(p, shape=(self.bwt_len(),))) if p else None

    # ---- scalar ops
    def initial_search_range(self, sym_idx):
        s, e = C.c_uint64(), C.c_uint64()
        lib().orc_initial_range(self.h, sym_idx, C.byref(s), C.byref(e))
        return int(s.value), int(e.value)

    def update_range_with_symbol(self, sp, ep, sym_idx):
        s, e = C.c_uint64(), C.c_uint64()
        lib().orc_update_range(self.h, sp, ep, sym_idx, C.byref(s), C.byref(e))
        return int(s.value), int(e.value)

    def backstep(self, p):
        return int(lib().orc_backstep(self.h, p))

    def global_occurrence(self, p, sym_idx):
        return int(lib().orc_global_occurrence(self.h, p, sym_idx))

    def symbol_at(self, p):
        return int(lib().orc_symbol_at(self.h, p))

    def search_range(self, q):
        b = _bytes(q)
        s, e = C.c_uint64(), C.c_uint64()
        rc = lib().orc_search_range(self.h, b.ctypes.data, len(b), C.byref(s), C.byref(e), None)
        if rc:
            raise ValueError("reference panics / is undefined on this query")
        return int(s.value), int(e.value)

    def count_string(self, q):
        b = _bytes(q)
        c = C.c_uint64()
        if lib().orc_count_string(self.h, b.ctypes.data, len(b), C.byref(c)):
            raise ValueError("reference panics / is undefined on this query")
        return int(c.value)

    def locate_string(self, q):
        """-> (global positions uint64[n], [(seq_idx, local_pos)] ) in ascending-BWT-row order"""
        b = _bytes(q)
        g, p, n = u64p(), C.POINTER(Pos)(), C.c_uint64()
        if lib().orc_locate_string(self.h, b.ctypes.data, len(b), C.byref(g), C.byref(p), C.byref(n), None):
            raise ValueError("reference panics / is undefined on this query")
        k = int(n.value)
        gp = np.ctypeslib.as_array(g, shape=(k,)).copy() if k else np.zeros(0, np.uint64)
        pos = [(int(p[i].seq_idx), int(p[i].local_pos)) for i in range(k)]
        lib().orc_free(g)
        lib().orc_free(p)
        return gp, pos

    def seq_location(self, gpos):
        o = Pos()
        lib().orc_seq_location(self.h, gpos, C.byref(o))
        return int(o.seq_idx), int(o.local_pos)

    def seq_location_ref(self, gpos):
        """literal reference recursion; None where it never terminates"""
        o = Pos()
        return None if lib().orc_seq_location_ref(self.h, gpos, C.byref(o)) else (int(o.seq_idx), int(o.local_pos))

    # ---- batch ops
    def parallel_count(self, qbytes, qoff, nthreads=1):
        qb, (qo, qop) = _bytes(qbytes), _u64(qoff)
        n = len(qo) - 1
        out = np.zeros(n, dtype=np.uint64)
        t = Tally()
        rc = lib().orc_parallel_count(self.h, qb.ctypes.data, qop, n, out.ctypes.data_as(u64p), nthreads, C.byref(t))
        if rc:
            raise ValueError("a query in the batch is undefined in the reference")
        return out, t.as_dict()

    def parallel_locate(self, qbytes, qoff, nthreads=1):
        qb, (qo, qop) = _bytes(qbytes), _u64(qoff)
        n = len(qo) - 1
        off, g, p = u64p(), u64p(), C.POINTER(Pos)()
        t = Tally()
        rc = lib().orc_parallel_locate(self.h, qb.ctypes.data, qop, n, C.byref(off), C.byref(g), C.byref(p), nthreads, C.byref(t))
        offs = np.ctypeslib.as_array(off, shape=(n + 1,)).copy()
        tot = int(offs[-1])
        gp = np.ctypeslib.as_array(g, shape=(tot,)).copy() if tot else np.zeros(0, np.uint64)
        pos = (np.ctypeslib.as_array(C.cast(p, u64p), shape=(tot * 2,)).copy().reshape(-1, 2)
               if tot else np.zeros((0, 2), np.uint64))
        for x in (off, g, p):
            lib().orc_free(x)
        if rc:
            raise ValueError("a query in the batch is undefined in the reference")
        return offs, gp, pos, t.as_dict()
