"""Locates and loads libawry_hip.so and declares the C ABI of include/awry_hip.h for ctypes."""
import ctypes as C
import importlib.util
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "lib", "libawry_hip.so")
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "awry_hip.h")
_lib = None


def lib_path():
    return _SO


class Pos(C.Structure):
    _fields_ = [("seq_idx", C.c_uint64), ("local_pos", C.c_uint64)]


class Range(C.Structure):
    _fields_ = [("start_ptr", C.c_uint64), ("end_ptr", C.c_uint64)]


class BuildArgs(C.Structure):
    _fields_ = [("input_path", C.c_char_p), ("sa_tmp_path", C.c_char_p), ("sa_ratio", C.c_uint64),
                ("kmer_len", C.c_uint8), ("alphabet", C.c_uint8), ("max_query_len", C.c_uint64),
                ("remove_tmp", C.c_uint8)]


def header_symbols():
    """every function name declared in include/awry_hip.h"""
    src = open(_HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(awry_[a-z0-9_]+)\s*\(", src)))


def _preload_hip_runtime():
    # PyTorch's ROCm wheel bundles its own libamdhip64.so (same SONAME as /opt/rocm's).  A process that
    # uses both this library and torch must end up with ONE HIP runtime, so load torch's copy first when
    # torch is installed; libawry_hip.so's DT_NEEDED libamdhip64.so.7 then resolves to it.
    spec = importlib.util.find_spec("torch")
    if spec and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError(
            "libawry_hip.so is not built (%s).  Run `python -m awry_amd.build` (needs hipcc); "
            "there is no CPU fallback for the search path." % _SO)
    _preload_hip_runtime()
    L = C.CDLL(_SO)
    vp, u64, u8, i32, cp = C.c_void_p, C.c_uint64, C.c_uint8, C.c_int, C.c_char_p
    u64p, u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)
    vpp = C.POINTER(vp)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype, f.argtypes = res, list(args)

    sig("awry_last_error", cp)
    sig("awry_build", i32, C.POINTER(BuildArgs), vpp)
    sig("awry_build_from_text", i32, vp, u64, i32, u64, u8, u64p, C.POINTER(cp), u64, vpp)
    sig("awry_build_from_text_on", i32, vp, u64, i32, u64, u8, u64p, C.POINTER(cp), u64, i32, vpp)
    sig("awry_load", i32, cp, vpp)
    sig("awry_save", i32, vp, cp)
    sig("awry_free", None, vp)
    sig("awry_set_devices", i32, vp, C.POINTER(i32), i32)
    sig("awry_set_seed_kmer_len", i32, vp, i32)
    sig("awry_seed_kmer_len", i32, vp)
    sig("awry_debug_set_count_kernel", i32, i32)
    sig("awry_debug_force_wide_rows", i32, i32)
    sig("awry_count_schedule", cp, vp, i32)
    sig("awry_num_devices", i32, vp)
    sig("awry_replica_device", i32, vp, i32)
    sig("awry_count_batch", i32, vp, vp, u64p, u64, u64p)
    sig("awry_count_packed_kmers", i32, vp, u64p, u64, i32, u64p)
    sig("awry_locate_batch", i32, vp, vp, u64p, u64, C.POINTER(u64p), C.POINTER(C.POINTER(Pos)), C.POINTER(u64p))
    sig("awry_free_buffer", None, vp)
    sig("awry_count", i32, vp, vp, u64, u64p)
    sig("awry_search_range", i32, vp, vp, u64, C.POINTER(Range))
    sig("awry_locate", i32, vp, vp, u64, C.POINTER(C.POINTER(Pos)), C.POINTER(u64p), u64p)
    sig("awry_initial_range", i32, vp, u8, C.POINTER(Range))
    sig("awry_update_range", i32, vp, Range, u8, C.POINTER(Range))
    sig("awry_backstep", i32, vp, u64, u64p)
    sig("awry_get_seq_location", i32, vp, u64, C.POINTER(Pos))
    sig("awry_alphabet", i32, vp)
    for n in ("awry_bwt_len", "awry_version", "awry_sa_ratio", "awry_num_sequences", "awry_sentinel_row"):
        sig(n, u64, vp)
    sig("awry_kmer_len", u8, vp)
    for n in ("awry_prefix_sums", "awry_block_words", "awry_sa_words"):
        sig(n, u64p, vp, u64p)
    sig("awry_sequence_start", u64, vp, u64)
    sig("awry_sequence_header", cp, vp, u64)
    sig("awry_block_reference_layout", i32, vp, u64, u64p, u64)
    sig("awry_read_query_file", i32, cp, C.POINTER(u8p), C.POINTER(u64p), u64p)
    sig("awry_host_suffix_array", i32, vp, u64, u64p)
    sig("awry_symbol_index", u8, i32, u8)
    sig("awry_host_pack_nt2", i32, vp, u64p, u64, u64, u64p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), u64p)
    sig("awry_host_threads", i32)
    sig("awry_host_memcpy", None, vp, vp, u64)
    sig("awry_dev_pack_nt2", i32, vp, i32, vp, u64, i32, vp, vp, vp)
    sig("awry_dev_count_nt2", i32, vp, i32, vp, u64, i32, vp, i32, vp)
    sig("awry_dev_count_nt2_tally", i32, vp, i32, vp, u64, i32, vp, i32, vp, vp)
    sig("awry_dev_count_ascii", i32, vp, i32, vp, vp, u64, vp, vp, vp, vp)
    sig("awry_dev_count_ascii_for_locate", i32, vp, i32, vp, vp, u64, vp, vp, vp, vp)
    sig("awry_dev_count_ascii_uniform", i32, vp, i32, vp, u64, u64, vp, vp, vp)
    sig("awry_dev_count_ascii_uniform_tally", i32, vp, i32, vp, u64, u64, vp, vp, vp)
    sig("awry_dev_scan_scratch_bytes", u64, u64)
    sig("awry_dev_scan_counts", i32, vp, i32, vp, u64, vp, vp, vp)
    sig("awry_dev_locate", i32, vp, i32, vp, i32, vp, u64, u64, vp, vp, vp)
    sig("awry_dev_count_nt2_long", i32, vp, i32, vp, u64, i32, vp, vp, i32, vp)
    sig("awry_dev_locate_tally", i32, vp, i32, vp, i32, vp, u64, u64, vp, vp, vp, vp)
    sig("awry_dev_phase_marker", i32, vp, i32, i32, vp)
    sig("awry_set_locate_sa_ratio", i32, vp, i32)
    sig("awry_locate_sa_ratio", i32, vp)
    sig("awry_set_verify", i32, vp, i32)
    sig("awry_verify_enabled", i32, vp)
    sig("awry_set_verify_kmers", i32, vp, i32)
    sig("awry_set_lcx", i32, vp, i32)
    sig("awry_lcx_enabled", i32, vp)
    sig("awry_debug_lcx", i32, vp, i32, vpp, vpp)
    sig("awry_dev_stream_copy", i32, vp, i32, vp, vp, u64, vp)
    sig("awry_dev_malloc", i32, vp, i32, u64, vpp)
    sig("awry_dev_free", i32, vp, i32, vp)
    sig("awry_dev_memcpy_h2d", i32, vp, i32, vp, vp, u64)
    sig("awry_dev_memcpy_d2h", i32, vp, i32, vp, vp, u64)
    sig("awry_dev_memset", i32, vp, i32, vp, i32, u64)
    sig("awry_dev_synchronize", i32, vp, i32)
    sig("awry_dev_timer_begin", i32, vp, i32, vp)
    sig("awry_dev_timer_end", i32, vp, i32, vp, C.POINTER(C.c_float))
    _lib = L
    return L
