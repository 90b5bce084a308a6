"""Host-side mirror of the reference's public `FmIndex` surface (/root/reference src/fm_index.rs) over the
C ABI of libawry_hip.so.  Same names, argument meaning and error behaviour: query functions raise where
the reference panics (empty query, '$'/'#').  All searching happens in HIP kernels on the GPU(s) chosen
with `set_devices`; nothing here computes a count or a location on the CPU."""
import ctypes as C
import os
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence

import numpy as np

from . import _lib

NUCLEOTIDE, AMINO = 0, 1
BUILD_HOST, BUILD_AUTO = -1, -2


class SymbolAlphabet:  # src/alphabet.rs:28-31
    Nucleotide = NUCLEOTIDE
    Amino = AMINO


class AwryError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("awry error %d: %s" % (code, msg))
        self.code = code


ERR_IO, ERR_FORMAT, ERR_INVALID_QUERY, ERR_HIP, ERR_OOM, ERR_ARG, ERR_NO_DEVICE = -1, -2, -3, -4, -5, -6, -7


@dataclass
class FmBuildArgs:  # src/fm_index.rs:78-96
    input_file_src: str
    suffix_array_output_src: Optional[str] = None
    suffix_array_compression_ratio: Optional[int] = None
    lookup_table_kmer_len: Optional[int] = None
    alphabet: int = NUCLEOTIDE
    max_query_len: Optional[int] = None
    remove_intermediate_suffix_array_file: bool = False


@dataclass(frozen=True, order=True)
class LocalizedSequencePosition:  # src/sequence_index.rs:31-78
    sequence_idx: int
    local_position: int


@dataclass(frozen=True, order=True)
class SearchRange:  # src/search.rs:25-81
    start_ptr: int
    end_ptr: int

    @staticmethod
    def zero():
        return SearchRange(1, 0)

    def is_empty(self):
        return self.start_ptr > self.end_ptr

    def len(self):
        return 0 if self.is_empty() else self.end_ptr - self.start_ptr + 1

    def range_iter(self):
        return range(0) if self.is_empty() else range(self.start_ptr, self.end_ptr + 1)


def _check(rc):
    if rc != 0:
        raise AwryError(rc, _lib.load_library().awry_last_error().decode(errors="replace"))


def _as_bytes(q):
    if isinstance(q, str):
        return q.encode("latin-1")
    if isinstance(q, np.ndarray):
        return np.ascontiguousarray(q, dtype=np.uint8).tobytes()
    return bytes(q)


def pack_queries(queries: Iterable):
    """iterable of str / bytes / uint8 arrays -> (uint8[total], uint64[n+1]) CSR"""
    qs = [_as_bytes(q) for q in queries]
    off = np.zeros(len(qs) + 1, dtype=np.uint64)
    if qs:
        off[1:] = np.cumsum([len(q) for q in qs], dtype=np.uint64)
    buf = np.frombuffer(b"".join(qs), dtype=np.uint8).copy() if qs else np.zeros(0, np.uint8)
    return buf, off


_u64p = C.POINTER(C.c_uint64)


class _Owned:
    """a malloc'ed result buffer of the C ABI, exposed to numpy without a copy and released (awry_free_buffer) with
    the last array that views it"""

    def __init__(self, lib, addr, nbytes):
        self._lib, self._addr = lib, addr
        self.__array_interface__ = {"data": (addr, False), "shape": (nbytes,), "typestr": "|u1", "version": 3}

    def __del__(self):
        try:
            self._lib.awry_free_buffer(C.c_void_p(self._addr))
        except Exception:
            pass


def _adopt(lib, ptr, count, dtype):
    addr = C.cast(ptr, C.c_void_p).value
    if not count or not addr:
        if addr:
            lib.awry_free_buffer(C.c_void_p(addr))
        return np.zeros(0, dtype)
    return np.asarray(_Owned(lib, addr, count * np.dtype(dtype).itemsize)).view(dtype)


def read_query_file(path):
    """FASTA / FASTQ file -> (uint8 bytes, uint64 offsets[n+1]): one query per record (query ingestion)"""
    L = _lib.load_library()
    b, o, n = C.POINTER(C.c_uint8)(), _u64p(), C.c_uint64()
    _check(L.awry_read_query_file(os.fsencode(path), C.byref(b), C.byref(o), C.byref(n)))
    off = _adopt(L, o, int(n.value) + 1, np.uint64)
    qb = _adopt(L, b, int(off[-1]), np.uint8)
    return qb, off


class FmIndex:
    def __init__(self, handle):
        self._L = _lib.load_library()
        self._h = C.c_void_p(handle)

    # ------------------------------------------------------------------ construction / persistence
    @classmethod
    def new(cls, args: FmBuildArgs) -> "FmIndex":
        """FmIndex::new, src/fm_index.rs:142-268"""
        L = _lib.load_library()
        a = _lib.BuildArgs(os.fsencode(args.input_file_src),
                           os.fsencode(args.suffix_array_output_src) if args.suffix_array_output_src else None,
                           args.suffix_array_compression_ratio or 0, args.lookup_table_kmer_len or 0, args.alphabet,
                           args.max_query_len or 0, 1 if args.remove_intermediate_suffix_array_file else 0)
        h = C.c_void_p()
        _check(L.awry_build(C.byref(a), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_text(cls, text, alphabet=NUCLEOTIDE, sa_ratio=8, kmer_len=0, seq_starts=(0,), headers=("seq0",),
                  build_device=BUILD_AUTO) -> "FmIndex":
        """index an in-memory text that already follows the reference's text model (ends in '$');
        build_device: GPU id, BUILD_HOST (host SA-IS) or BUILD_AUTO"""
        L = _lib.load_library()
        t = np.frombuffer(_as_bytes(text), dtype=np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, np.uint8)
        st = np.ascontiguousarray(list(seq_starts), dtype=np.uint64)
        if len(st) != len(headers):
            raise ValueError("seq_starts and headers must have one entry per record (%d != %d)" % (len(st), len(headers)))
        hd = (C.c_char_p * len(headers))(*[h.encode() for h in headers])
        h = C.c_void_p()
        _check(L.awry_build_from_text_on(t.ctypes.data, len(t), alphabet, sa_ratio, kmer_len, st.ctypes.data_as(_u64p), hd,
                                         len(headers), build_device, C.byref(h)))
        return cls(h.value)

    @classmethod
    def load(cls, path) -> "FmIndex":
        """FmIndex::load, src/fm_index_file.rs:132"""
        L = _lib.load_library()
        h = C.c_void_p()
        _check(L.awry_load(os.fsencode(path), C.byref(h)))
        return cls(h.value)

    def save(self, path):
        """FmIndex::save, src/fm_index_file.rs:42"""
        _check(self._L.awry_save(self._h, os.fsencode(path)))

    def close(self):
        if self._h:
            self._L.awry_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ device placement
    def set_devices(self, device_ids: Sequence[int] = (0,)) -> "FmIndex":
        ids = (C.c_int * len(device_ids))(*device_ids)
        _check(self._L.awry_set_devices(self._h, ids, len(device_ids)))
        return self

    def set_seed_kmer_len(self, k: int):
        _check(self._L.awry_set_seed_kmer_len(self._h, k))

    def seed_kmer_len(self) -> int:
        return self._L.awry_seed_kmer_len(self._h)

    def count_schedule(self, L: int) -> str:
        """kernel(s) dev_count_nt2 launches for k-mers of length L"""
        return self._L.awry_count_schedule(self._h, L).decode()

    def num_devices(self) -> int:
        return self._L.awry_num_devices(self._h)

    # ------------------------------------------------------------------ accessors (src/fm_index.rs:302-399)
    def alphabet(self) -> int:
        return self._L.awry_alphabet(self._h)

    def bwt_len(self) -> int:
        return int(self._L.awry_bwt_len(self._h))

    def version_number(self) -> int:
        return int(self._L.awry_version(self._h))

    def suffix_array_compression_ratio(self) -> int:
        return int(self._L.awry_sa_ratio(self._h))

    def lookup_table_kmer_len(self) -> int:
        return int(self._L.awry_kmer_len(self._h))

    def sentinel_row(self) -> int:
        return int(self._L.awry_sentinel_row(self._h))

    def _arr(self, fn):
        n = C.c_uint64()
        p = getattr(self._L, fn)(self._h, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(int(n.value),)).copy() if n.value else np.zeros(0, np.uint64)

    def prefix_sums(self) -> np.ndarray:
        return self._arr("awry_prefix_sums")

    def device_block_words(self) -> np.ndarray:
        return self._arr("awry_block_words")

    def sa_words(self) -> np.ndarray:
        return self._arr("awry_sa_words")

    def reference_block_words(self) -> np.ndarray:
        """all BWT blocks converted to the reference layout (planes then milestones, src/bwt.rs:12-25)"""
        rw = 20 if self.alphabet() == NUCLEOTIDE else 44
        nb = (self.bwt_len() + 255) // 256
        out = np.zeros(nb * rw, dtype=np.uint64)
        for b in range(nb):
            _check(self._L.awry_block_reference_layout(self._h, b, out[b * rw:].ctypes.data_as(_u64p), rw))
        return out

    def sequences(self):
        n = int(self._L.awry_num_sequences(self._h))
        return [(int(self._L.awry_sequence_start(self._h, i)), self._L.awry_sequence_header(self._h, i).decode()) for i in range(n)]

    # ------------------------------------------------------------------ scalar queries
    def count_string(self, query) -> int:
        """src/fm_index.rs:499-501"""
        q = _as_bytes(query)
        c = C.c_uint64()
        _check(self._L.awry_count(self._h, q, len(q), C.byref(c)))
        return int(c.value)

    def search_range(self, query) -> SearchRange:
        """get_search_range_for_string, src/fm_index.rs:402-438"""
        q = _as_bytes(query)
        r = _lib.Range()
        _check(self._L.awry_search_range(self._h, q, len(q), C.byref(r)))
        return SearchRange(int(r.start_ptr), int(r.end_ptr))

    def locate_string(self, query) -> List[LocalizedSequencePosition]:
        """src/fm_index.rs:516-544 (ascending BWT-row order)"""
        return [LocalizedSequencePosition(int(a), int(b)) for a, b in self.locate_string_raw(query)[1]]

    def locate_string_raw(self, query):
        """-> (global positions uint64[n], (seq_idx, local_pos) uint64[n, 2])"""
        q = _as_bytes(query)
        hits, gp, n = C.POINTER(_lib.Pos)(), _u64p(), C.c_uint64()
        _check(self._L.awry_locate(self._h, q, len(q), C.byref(hits), C.byref(gp), C.byref(n)))
        k = int(n.value)
        g = np.ctypeslib.as_array(gp, shape=(k,)).copy() if k else np.zeros(0, np.uint64)
        p = np.ctypeslib.as_array(C.cast(hits, _u64p), shape=(2 * k,)).copy().reshape(-1, 2) if k else np.zeros((0, 2), np.uint64)
        self._L.awry_free_buffer(hits)
        self._L.awry_free_buffer(gp)
        return g, p

    def initial_search_range(self, symbol) -> SearchRange:
        r = _lib.Range()
        _check(self._L.awry_initial_range(self._h, ord(symbol) if isinstance(symbol, str) else int(symbol), C.byref(r)))
        return SearchRange(int(r.start_ptr), int(r.end_ptr))

    def update_range_with_symbol(self, search_range: SearchRange, symbol) -> SearchRange:
        """src/fm_index.rs:559-582"""
        r = _lib.Range()
        _check(self._L.awry_update_range(self._h, _lib.Range(search_range.start_ptr, search_range.end_ptr),
                                         ord(symbol) if isinstance(symbol, str) else int(symbol), C.byref(r)))
        return SearchRange(int(r.start_ptr), int(r.end_ptr))

    def backstep(self, search_pointer: int) -> int:
        """src/fm_index.rs:585-593"""
        o = C.c_uint64()
        _check(self._L.awry_backstep(self._h, search_pointer, C.byref(o)))
        return int(o.value)

    def get_seq_location(self, global_position: int) -> LocalizedSequencePosition:
        p = _lib.Pos()
        _check(self._L.awry_get_seq_location(self._h, global_position, C.byref(p)))
        return LocalizedSequencePosition(int(p.seq_idx), int(p.local_pos))

    # ------------------------------------------------------------------ batch queries
    def parallel_count_csr(self, qbytes: np.ndarray, qoff: np.ndarray, out: Optional[np.ndarray] = None) -> np.ndarray:
        """counts of the CSR batch in input order; `out` (uint64[n], contiguous) is filled and returned when given -- the
        C ABI's counts_out is caller-owned, and a caller that reuses its result array spares the batch the first-touch
        page faults of a fresh one"""
        qb = np.ascontiguousarray(qbytes, dtype=np.uint8)
        qo = np.ascontiguousarray(qoff, dtype=np.uint64)
        n = len(qo) - 1
        if out is None:
            out = np.empty(n, dtype=np.uint64)
        elif out.dtype != np.uint64 or out.shape != (n,) or not out.flags.c_contiguous:
            raise ValueError("out must be a contiguous uint64 array with one entry per query")
        _check(self._L.awry_count_batch(self._h, qb.ctypes.data, qo.ctypes.data_as(_u64p), n, out.ctypes.data_as(_u64p)))
        return out

    def parallel_count_packed(self, words: np.ndarray, L: int, out: Optional[np.ndarray] = None) -> np.ndarray:
        """k-mers already packed 2 bits per letter (letter j of a k-mer in bits [2j, 2j+2) of its uint64, A0 C1 G2 T3)"""
        w = np.ascontiguousarray(words, dtype=np.uint64)
        if out is None:
            out = np.empty(len(w), dtype=np.uint64)
        elif out.dtype != np.uint64 or out.shape != (len(w),) or not out.flags.c_contiguous:
            raise ValueError("out must be a contiguous uint64 array with one entry per query")
        _check(self._L.awry_count_packed_kmers(self._h, w.ctypes.data_as(_u64p), len(w), L, out.ctypes.data_as(_u64p)))
        return out

    def parallel_count(self, queries: Iterable) -> np.ndarray:
        """src/fm_index.rs:455-460: counts in input order"""
        return self.parallel_count_csr(*pack_queries(queries))

    def parallel_locate_csr(self, qbytes: np.ndarray, qoff: np.ndarray, want_pos: bool = True):
        """-> (hit_off uint64[n+1], global_pos uint64[total], pos uint64[total, 2]); want_pos=False passes hits_out = NULL
        (text positions only: 8 B per hit cross PCIe instead of 24) and returns an empty pos"""
        qb = np.ascontiguousarray(qbytes, dtype=np.uint8)
        qo = np.ascontiguousarray(qoff, dtype=np.uint64)
        n = len(qo) - 1
        off, hits, gp = _u64p(), C.POINTER(_lib.Pos)(), _u64p()
        _check(self._L.awry_locate_batch(self._h, qb.ctypes.data, qo.ctypes.data_as(_u64p), n, C.byref(off),
                                         C.byref(hits) if want_pos else None, C.byref(gp)))
        offs = _adopt(self._L, off, n + 1, np.uint64)
        tot = int(offs[-1])
        g = _adopt(self._L, gp, tot, np.uint64)
        p = _adopt(self._L, hits, 2 * tot, np.uint64).reshape(-1, 2) if want_pos else np.zeros((0, 2), np.uint64)
        return offs, g, p

    def parallel_locate(self, queries: Iterable) -> List[List[LocalizedSequencePosition]]:
        """src/fm_index.rs:479-487: outer order = input order, inner order = ascending BWT row"""
        off, _, p = self.parallel_locate_csr(*pack_queries(queries))
        return [[LocalizedSequencePosition(int(a), int(b)) for a, b in p[off[i]:off[i + 1]]] for i in range(len(off) - 1)]

    # ------------------------------------------------------------------ device-resident path
    def dev_malloc(self, nbytes, slot=0) -> int:
        p = C.c_void_p()
        _check(self._L.awry_dev_malloc(self._h, slot, nbytes, C.byref(p)))
        return p.value

    def dev_free(self, ptr, slot=0):
        _check(self._L.awry_dev_free(self._h, slot, ptr))

    def dev_upload(self, arr: np.ndarray, slot=0) -> int:
        a = np.ascontiguousarray(arr)
        p = self.dev_malloc(a.nbytes, slot)
        _check(self._L.awry_dev_memcpy_h2d(self._h, slot, p, a.ctypes.data, a.nbytes))
        return p

    def dev_download(self, ptr, shape, dtype, slot=0) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        _check(self._L.awry_dev_memcpy_d2h(self._h, slot, out.ctypes.data, ptr, out.nbytes))
        return out

    def dev_memset(self, ptr, value, nbytes, slot=0):
        _check(self._L.awry_dev_memset(self._h, slot, ptr, value, nbytes))

    def dev_synchronize(self, slot=0):
        _check(self._L.awry_dev_synchronize(self._h, slot))

    def dev_pack_nt2(self, d_ascii, n, L, d_words, d_bad, stream=None, slot=0):
        _check(self._L.awry_dev_pack_nt2(self._h, slot, d_ascii, n, L, d_words, d_bad, stream))

    def dev_count_nt2(self, d_words, n, L, d_counts, use_seed=True, stream=None, slot=0):
        _check(self._L.awry_dev_count_nt2(self._h, slot, d_words, n, L, d_counts, 1 if use_seed else 0, stream))

    def dev_count_nt2_tally(self, d_words, n, L, d_counts, d_tally, use_seed=True, stream=None, slot=0):
        _check(self._L.awry_dev_count_nt2_tally(self._h, slot, d_words, n, L, d_counts, 1 if use_seed else 0, d_tally, stream))

    def dev_count_ascii(self, d_qbytes, d_qoff, n, d_counts, d_ranges=None, d_status=None, stream=None, slot=0):
        _check(self._L.awry_dev_count_ascii(self._h, slot, d_qbytes, d_qoff, n, d_counts, d_ranges, d_status, stream))

    def dev_count_ascii_for_locate(self, d_qbytes, d_qoff, n, d_counts, d_locate_words, d_status=None, stream=None, slot=0):
        """count pass of a device-resident parallel_locate: d_locate_words[2n] are opaque words for dev_locate (range_stride 2)"""
        _check(self._L.awry_dev_count_ascii_for_locate(self._h, slot, d_qbytes, d_qoff, n, d_counts, d_locate_words, d_status, stream))

    def dev_count_ascii_uniform(self, d_qbytes, n, length, d_counts, d_status=None, stream=None, slot=0):
        """n ASCII queries of `length` bytes each, back to back (amino k-mers: the two-phase schedule)"""
        _check(self._L.awry_dev_count_ascii_uniform(self._h, slot, d_qbytes, n, length, d_counts, d_status, stream))

    def dev_count_ascii_uniform_tally(self, d_qbytes, n, length, d_counts, d_tally, stream=None, slot=0):
        """amino k-mer schedule + census: d_tally[5] += (seed probes, steps, blocks ranked, SA reads, text comparisons)"""
        _check(self._L.awry_dev_count_ascii_uniform_tally(self._h, slot, d_qbytes, n, length, d_counts, d_tally, stream))

    def dev_scan_counts(self, d_counts, n, d_hit_off, d_scratch, stream=None, slot=0):
        _check(self._L.awry_dev_scan_counts(self._h, slot, d_counts, n, d_hit_off, d_scratch, stream))

    def dev_scan_scratch_bytes(self, n) -> int:
        return int(self._L.awry_dev_scan_scratch_bytes(n))

    def dev_locate(self, d_ranges, d_hit_off, n, total, d_gpos, d_pos=None, stream=None, slot=0, range_stride=2):
        _check(self._L.awry_dev_locate(self._h, slot, d_ranges, range_stride, d_hit_off, n, total, d_gpos, d_pos, stream))

    def dev_locate_tally(self, d_ranges, d_hit_off, n, total, d_gpos, d_pos, d_tally, stream=None, slot=0, range_stride=2):
        """dev_locate + the walk kernel's census: d_tally[2] += (LF steps, hits that walked)"""
        _check(self._L.awry_dev_locate_tally(self._h, slot, d_ranges, range_stride, d_hit_off, n, total, d_gpos, d_pos, d_tally, stream))

    def dev_phase_marker(self, phase_id, stream=None, slot=0):
        """profiling aid: an empty kernel whose grid size names a phase in rocprofv3 counter output"""
        _check(self._L.awry_dev_phase_marker(self._h, slot, phase_id, stream))

    def dev_count_nt2_long(self, d_words, n, L, d_counts, d_range_start=None, use_seed=True, stream=None, slot=0):
        _check(self._L.awry_dev_count_nt2_long(self._h, slot, d_words, n, L, d_counts, d_range_start, 1 if use_seed else 0, stream))

    def set_locate_sa_ratio(self, ratio: int):
        """device-side SA density for locate (0 = the file's samples); results do not depend on it"""
        _check(self._L.awry_set_locate_sa_ratio(self._h, ratio))

    def locate_sa_ratio(self) -> int:
        return self._L.awry_locate_sa_ratio(self._h)

    def set_lcx(self, on: bool):
        """left-context index on / off (performance knob; results do not depend on it)"""
        _check(self._L.awry_set_lcx(self._h, 1 if on else 0))

    def lcx_enabled(self) -> bool:
        return bool(self._L.awry_lcx_enabled(self._h))

    def debug_lcx(self, slot=0):
        """-> (keys uint64[bwt_len], rowpos uint64[bwt_len]) copied from replica `slot`, or None when it is not resident"""
        k, rp = C.c_void_p(), C.c_void_p()
        _check(self._L.awry_debug_lcx(self._h, slot, C.byref(k), C.byref(rp)))
        if not k.value:
            return None
        n = self.bwt_len()
        keys, rowpos = np.empty(n, np.uint64), np.empty(n, np.uint64)
        _check(self._L.awry_dev_memcpy_d2h(self._h, slot, keys.ctypes.data, k, n * 8))
        _check(self._L.awry_dev_memcpy_d2h(self._h, slot, rowpos.ctypes.data, rp, n * 8))
        return keys, rowpos

    def dev_stream_copy(self, d_dst, d_src, nbytes, stream=None, slot=0):
        _check(self._L.awry_dev_stream_copy(self._h, slot, d_dst, d_src, nbytes, stream))

    def set_verify(self, after_steps: int):
        """seed-and-verify for packed nucleotide reads (-1 = off); results do not depend on it"""
        _check(self._L.awry_set_verify(self._h, after_steps))

    def set_verify_kmers(self, on: bool):
        """let the k-mer (L <= 32) kernel use seed-and-verify too (pays off on batches of present k-mers)"""
        _check(self._L.awry_set_verify_kmers(self._h, 1 if on else 0))

    def verify_enabled(self) -> bool:
        return bool(self._L.awry_verify_enabled(self._h))

    def locate_reads_nt2(self, q2d: np.ndarray, use_seed=True, slot=0):
        """fixed-length ACGT reads uint8[n, L] of any L through the packed pipeline:
        pack -> seeded quad count -> scan -> tile locate.  -> (hit_off, global_pos, pos[total, 2])"""
        q2d = np.ascontiguousarray(q2d, dtype=np.uint8)
        n, L = q2d.shape
        W = (L + 31) // 32
        d_ascii = self.dev_upload(q2d.reshape(-1), slot)
        bufs = [d_ascii]
        try:
            d_words, d_counts, d_sp, d_bad = (self.dev_malloc(8 * n * W, slot), self.dev_malloc(8 * n, slot), self.dev_malloc(8 * n, slot),
                                              self.dev_malloc(8, slot))
            d_off, d_scr = self.dev_malloc(8 * (n + 1), slot), self.dev_malloc(self.dev_scan_scratch_bytes(n), slot)
            bufs += [d_words, d_counts, d_sp, d_bad, d_off, d_scr]
            self.dev_memset(d_bad, 0, 8, slot)
            self.dev_pack_nt2(d_ascii, n, L, d_words, d_bad, None, slot)
            self.dev_count_nt2_long(d_words, n, L, d_counts, d_sp, use_seed, None, slot)
            self.dev_scan_counts(d_counts, n, d_off, d_scr, None, slot)
            self.dev_synchronize(slot)
            if int(self.dev_download(d_bad, (1,), np.uint64, slot)[0]):
                raise AwryError(ERR_INVALID_QUERY, "reads contain bytes outside ACGT; use parallel_locate")
            off = self.dev_download(d_off, (n + 1,), np.uint64, slot)
            total = int(off[-1])
            d_g, d_p = self.dev_malloc(8 * max(total, 1), slot), self.dev_malloc(16 * max(total, 1), slot)
            bufs += [d_g, d_p]
            self.dev_locate(d_sp, d_off, n, total, d_g, d_p, None, slot, range_stride=1)
            self.dev_synchronize(slot)
            return off, self.dev_download(d_g, (total,), np.uint64, slot), self.dev_download(d_p, (total, 2), np.uint64, slot)
        finally:
            for p in bufs:
                self.dev_free(p, slot)

    def dev_timer_begin(self, stream=None, slot=0):
        _check(self._L.awry_dev_timer_begin(self._h, slot, stream))

    def dev_timer_end(self, stream=None, slot=0) -> float:
        ms = C.c_float()
        _check(self._L.awry_dev_timer_end(self._h, slot, stream, C.byref(ms)))
        return float(ms.value)

    # convenience: count packed-able fixed-length ACGT k-mers given as uint8[n, L] through the hot kernel
    def count_kmers_nt2(self, q2d: np.ndarray, use_seed=True, slot=0, tally=False):
        """fixed-length ACGT k-mers uint8[n, L <= 32] through the packed kernel -> counts; tally=True also returns the
        kernel's census uint64[5] = (seed probes, LF steps, ranked blocks, verify SA reads, verify text windows)"""
        q2d = np.ascontiguousarray(q2d, dtype=np.uint8)
        n, L = q2d.shape
        d_ascii = self.dev_upload(q2d.reshape(-1), slot)
        d_words, d_counts, d_bad = self.dev_malloc(8 * n, slot), self.dev_malloc(8 * n, slot), self.dev_malloc(8, slot)
        d_tally = self.dev_malloc(64, slot) if tally else None
        try:
            self.dev_memset(d_bad, 0, 8, slot)
            self.dev_pack_nt2(d_ascii, n, L, d_words, d_bad, None, slot)
            self.dev_synchronize(slot)
            bad = int(self.dev_download(d_bad, (1,), np.uint64, slot)[0])
            if bad:
                raise AwryError(ERR_INVALID_QUERY, "%d queries contain bytes outside ACGT; use parallel_count" % bad)
            if tally:
                self.dev_memset(d_tally, 0, 64, slot)
                self.dev_count_nt2_tally(d_words, n, L, d_counts, d_tally, use_seed, None, slot)
            else:
                self.dev_count_nt2(d_words, n, L, d_counts, use_seed, None, slot)
            self.dev_synchronize(slot)
            counts = self.dev_download(d_counts, (n,), np.uint64, slot)
            return (counts, self.dev_download(d_tally, (5,), np.uint64, slot)) if tally else counts
        finally:
            for p in (d_ascii, d_words, d_counts, d_bad, d_tally):
                if p is not None:
                    self.dev_free(p, slot)
