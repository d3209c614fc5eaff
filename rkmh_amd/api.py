"""ctypes binding of include/rkmh_amd.h (the C ABI of librkmh_amd.so).

Mirrors the reference's interface for the hot path: the mkmh free functions called from
/root/reference/src/rkmh.cpp (calc_hashes :860, minhashes :863, hash_intersection_size :869, ...) as
methods of `Context`, plus the batched replacements of main_stream's loops (:813-898, :904-948).
Fails loudly when the HIP library is missing -- there is no CPU implementation behind this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FOLD_SWAP32, FOLD_H1, FOLD_W2W1 = 0, 1, 2


class RkmhError(RuntimeError):
    code = 0


class NeedFullDepthMap(RkmhError):
    """RK_ERR_NEED_FULL: a compact depth map cannot answer this input (reads with more hashes than the sketch keeps); repeat the
    pass with a full Counter."""
    code = -7


class Policy(C.Structure):
    _fields_ = [("fold", C.c_int32), ("drop_last_window", C.c_int32), ("counter_counts_zero", C.c_int32),
                ("mask_strict_less", C.c_int32), ("freq_max_inclusive", C.c_int32), ("seed", C.c_uint32)]


class CallRecord(C.Structure):
    _fields_ = [("ref", C.c_int32), ("pos", C.c_int32), ("alt_depth", C.c_int32), ("avg_d", C.c_int32), ("depth", C.c_int32),
                ("orig", C.c_uint8), ("alt", C.c_uint8), ("kind", C.c_uint8), ("pad", C.c_uint8)]


class SeqSet(C.Structure):
    _fields_ = [("nseq", C.c_int64), ("bases", C.POINTER(C.c_uint8)), ("offsets", C.POINTER(C.c_uint64)),
                ("names", C.POINTER(C.c_char)), ("name_offsets", C.POINTER(C.c_uint64)),
                ("quals", C.POINTER(C.c_char))]


def library_path():
    return os.path.join(_HERE, "lib", "librkmh_amd.so")


_u8p, _u64p, _i32p, _ip = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_int)

_SIGS = {
    "rk_default_policy": (None, [C.POINTER(Policy)]),
    "rk_policy_parse": (C.c_int, [C.c_char_p, C.POINTER(Policy)]),
    "rk_policy_describe": (C.c_int, [C.POINTER(Policy), C.c_char_p, C.c_size_t]),
    "rk_policy_same_hashes": (C.c_int, [C.POINTER(Policy), C.POINTER(Policy)]),
    "rk_ctx_policy": (C.c_int, [C.c_void_p, C.POINTER(Policy)]),
    "rk_last_error": (C.c_char_p, []),
    "rk_version": (C.c_char_p, []),
    "rk_device_count": (C.c_int, []),
    "rk_ctx_create": (C.c_int, [C.c_int, C.POINTER(Policy), C.POINTER(C.c_void_p)]),
    "rk_ctx_destroy": (None, [C.c_void_p]),
    "rk_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "rk_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "rk_free": (None, [C.c_void_p]),
    "rk_counter_add": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rk_counter_copy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rk_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "rk_host_free": (None, [C.c_void_p]),
    "rk_to_upper": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "rk_calc_hashes": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, _ip, C.c_int, C.POINTER(_u64p), _ip]),
    "rk_calc_hashes_counted": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, _ip, C.c_int, C.POINTER(_u64p), _ip, C.c_void_p]),
    "rk_calc_hash": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, _u64p]),
    "rk_minhashes": (C.c_int, [C.c_void_p, _u64p, C.c_int, C.c_int, C.POINTER(_u64p), _ip]),
    "rk_mask_by_frequency": (C.c_int, [C.c_void_p, _u64p, C.c_int, C.c_void_p, C.c_int]),
    "rk_minhashes_frequency_filter": (C.c_int, [C.c_void_p, _u64p, C.c_int, C.c_int, C.POINTER(_u64p), _ip, C.c_void_p, C.c_int, C.c_int]),
    "rk_hash_intersection_size": (C.c_int, [C.c_void_p, _u64p, C.c_int, _u64p, C.c_int, _ip]),
    "rk_hash_intersection": (C.c_int, [C.c_void_p, _u64p, C.c_int, C.c_int, _u64p, C.c_int, C.c_int, C.c_int, C.POINTER(_u64p), _ip]),
    "rk_counter_create": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "rk_counter_wrap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "rk_counter_destroy": (None, [C.c_void_p]),
    "rk_counter_clear": (C.c_int, [C.c_void_p]),
    "rk_counter_increment": (C.c_int, [C.c_void_p, C.c_uint64]),
    "rk_counter_get": (C.c_int, [C.c_void_p, C.c_uint64, _i32p]),
    "rk_classify_groups_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "rk_kmer_form": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "rk_set_kmer_form": (C.c_int, [C.c_void_p, C.c_int]),
    "rk_device_props": (C.c_int, [C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rk_counter_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rk_counter_load": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rk_counter_save_tagged": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32]),
    "rk_counter_load_tagged": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32]),
    "rk_depth_map_tag": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rk_counter_device_ptr": (C.c_void_p, [C.c_void_p]),
    "rk_counter_slots": (C.c_uint64, [C.c_void_p]),
    "rk_hash_batch": (C.c_int, [C.c_void_p, _u8p, _u64p, C.c_int64, _ip, C.c_int, C.POINTER(_u64p), _u64p]),
    "rk_sketch_batch": (C.c_int, [C.c_void_p, _u8p, _u64p, C.c_int64, _ip, C.c_int, C.c_int, _u64p, _i32p]),
    "rk_set_references": (C.c_int, [C.c_void_p, _u8p, _u64p, C.c_int, _ip, C.c_int, C.c_int, C.c_int, C.c_uint64]),
    "rk_set_reference_count_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "rk_set_reference_sketches": (C.c_int, [C.c_void_p, _u64p, _i32p, C.c_int, _ip, C.c_int, C.c_int]),
    "rk_get_reference_sketches": (C.c_int, [C.c_void_p, _u64p, _i32p]),
    "rk_num_references": (C.c_int, [C.c_void_p]),
    "rk_set_depth_filter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "rk_set_min_num_bound": (C.c_int, [C.c_void_p, C.c_int]),
    "rk_set_kmer_cache": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rk_kmer_cache_state": (C.c_int, [C.c_void_p]),
    "rk_bgzf_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "rk_bgzf_close": (None, [C.c_void_p]),
    "rk_bgzf_members": (C.c_int64, [C.c_void_p]),
    "rk_bgzf_text_bytes": (C.c_uint64, [C.c_void_p]),
    "rk_bgzf_text_offset": (C.c_uint64, [C.c_void_p, C.c_int64]),
    "rk_bgzf_first_byte": (C.c_int, [C.c_void_p]),
    "rk_bgzf_image": (C.c_void_p, [C.c_void_p]),
    "rk_bgzf_lead_member": (C.c_int64, [C.c_void_p, C.c_int64]),
    "rk_bgzf_member": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rk_fastq_slot_set_source": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rk_host_register_readonly": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rk_host_unregister": (None, [C.c_void_p]),
    "rk_warm_up": (C.c_int, [C.c_int, C.c_int]),
    "rk_bgzf_plan": (C.c_int64, [C.c_void_p, C.c_uint64, C.POINTER(C.c_int64), C.c_int64]),
    "rk_bgzf_plan_members": (C.c_int64, [C.c_void_p, C.c_uint64, C.c_int64, C.POINTER(C.c_int64), C.c_int64]),
    "rk_bgzf_fastq_records": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rk_fastq_slot_load_bgzf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rk_fasta_load_put_gzip": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "rk_fasta_load_put_newline": (C.c_int, [C.c_void_p, C.c_uint64]),
    "rk_fasta_load_put_bgzf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_uint64]),
    "rk_gzip_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "rk_gzip_close": (None, [C.c_void_p]),
    "rk_gzip_image": (C.c_void_p, [C.c_void_p]),
    "rk_gzip_file_bytes": (C.c_uint64, [C.c_void_p]),
    "rk_gzip_first_byte": (C.c_int, [C.c_void_p]),
    "rk_gzip_text_bytes_hint": (C.c_uint64, [C.c_void_p]),
    "rk_gzip_plan": (C.c_int64, [C.c_void_p, C.c_uint64]),
    "rk_gzip_calls": (C.c_int64, [C.c_void_p]),
    "rk_gzip_release_device": (None, [C.c_void_p]),
    "rk_gzip_stretch_bytes": (C.c_uint64, [C.c_void_p]),
    "rk_fastq_slot_reserve_gzip": (C.c_int, [C.c_void_p, C.c_uint64]),
    "rk_fastq_slot_load_gzip": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rk_bgzf_file_bytes": (C.c_uint64, [C.c_void_p]),
    # packed reads (`rkmh pack`, -F): include/rkmh_amd.h "PACKED READS"
    "rk_packed_encode": (C.c_int64, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]),
    "rk_packed_decode": (None, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "rk_classify_batch_device_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                                  C.c_uint32, C.c_void_p]),
    "rk_count_batch_device_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                               C.c_uint32, C.c_void_p]),
    "rk_packed_slot_create": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]),
    "rk_packed_slot_destroy": (None, [C.c_void_p]),
    "rk_packed_slot_classify": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rk_packed_slot_count": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rk_packed_filter_records_bound": (C.c_uint64, [C.c_void_p]),
    "rk_packed_filter_records": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64]),
    "rk_fastq_slot_create2": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]),
    "rk_fastq_slot_set_filter_output": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rk_fastq_slot_spans_base": (C.c_void_p, [C.c_void_p]),
    "rk_counter_create_compact": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_void_p)]),
    "rk_counter_compact_entries": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "rk_counter_entries": (C.c_uint64, [C.c_void_p]),
    "rk_counter_is_compact": (C.c_int, [C.c_void_p]),
    "rk_min_num_bound": (C.c_int, [C.c_void_p]),
    "rk_count_batch": (C.c_int, [C.c_void_p, _u8p, _u64p, C.c_int64, C.c_void_p]),
    "rk_count_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "rk_classify_batch": (C.c_int, [C.c_void_p, _u8p, _u64p, C.c_int64, _i32p]),
    "rk_classify_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "rk_classify_batch_device_all": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "rk_call": (C.c_int, [C.c_void_p, _u8p, _u64p, C.c_int, _u8p, _u64p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.POINTER(CallRecord)), C.POINTER(C.c_int64)]),
    "rk_format_stream_line": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rk_parse_files": (C.c_int, [C.POINTER(C.c_char_p), C.c_int, C.POINTER(SeqSet)]),
    "rk_seqset_free": (None, [C.POINTER(SeqSet)]),
    "rk_pool_trim": (None, []),
    "rk_reader_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "rk_reader_next": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.POINTER(SeqSet)]),
    "rk_reader_set_options": (None, [C.c_void_p, C.c_int]),
    "rk_reader_close": (None, [C.c_void_p]),
    "rk_reader_open_at": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "rk_reader_open_range": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]),
    "rk_reader_strict": (C.c_int, [C.c_void_p]),
    "rk_fastq_slot_create": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "rk_fastq_slot_text": (C.c_void_p, [C.c_void_p]),
    "rk_fastq_slot_classify": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "rk_fastq_slot_destroy": (None, [C.c_void_p]),
    "rk_fastq_slot_submit": (C.c_int, [C.c_void_p, C.c_uint64]),
    "rk_fastq_slot_finish": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rk_line_parts_create": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint64), C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rk_line_parts_destroy": (None, [C.c_void_p]),
    "rk_fastq_stream_lines_bound": (C.c_uint64, [C.c_void_p, C.c_void_p]),
    "rk_fastq_stream_lines": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "rk_fastq_filter_records_bound": (C.c_uint64, [C.c_void_p]),
    "rk_fastq_filter_records": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64]),
    "rk_fasta_load_create": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "rk_fasta_load_put": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]),
    "rk_fasta_load_finish": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "rk_fasta_load_get_bases": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rk_set_references_fasta": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_uint64]),
    "rk_fasta_load_destroy": (None, [C.c_void_p]),
    "rk_fastq_slot_count": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "rk_fastq_cut": (C.c_int64, [C.c_void_p, C.c_uint64]),
    "rk_synth_reads": (C.c_int, [_u8p, _u64p, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, _u8p, C.c_int]),
}


def _preload_torch_hip_runtime():
    """One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (same SONAME as /opt/rocm's).
    If librkmh_amd.so pulled in the system copy first, a later `import torch` would bring a second runtime
    that finds no GPU.  So when torch is installed, its runtime is loaded first and the library binds to it."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load_library():
    """Loads librkmh_amd.so (built in-tree by `make` / __graft_entry__.build()). No fallback."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise RkmhError("%s is missing: build it with `make` (hipcc --offload-arch=gfx950). "
                            "rkmh_amd has no CPU fallback." % path)
        _preload_torch_hip_runtime()
        lib = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)  # AttributeError if the ABI drifted
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def _chk(rc):
    if rc != 0:
        cls = NeedFullDepthMap if rc == -7 else RkmhError
        e = cls("rkmh_amd error %d: %s" % (rc, load_library().rk_last_error().decode()))
        e.code = rc
        raise e


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _ks(ks):
    if np.isscalar(ks):
        ks = [ks]
    return np.ascontiguousarray(ks, dtype=np.int32)


def _padded(bases):
    """uint8 copy with >= 8 readable bytes past the end (device staging reads aligned dwords)."""
    b = np.zeros(len(bases) + 16, dtype=np.uint8)
    b[: len(bases)] = np.frombuffer(bases, dtype=np.uint8) if isinstance(bases, (bytes, bytearray)) else bases
    return b


def pack(seqs):
    """list[bytes] -> (uint8 bases (padded), uint64 offsets[n+1])"""
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    return _padded(b"".join(seqs)), offs


def _seqset_to_py(ss):
    n = ss.nseq
    offs = np.ctypeslib.as_array(ss.offsets, shape=(n + 1,)).copy()
    nb = int(offs[-1])
    bases = np.zeros(nb + 16, dtype=np.uint8)
    if nb:
        bases[:nb] = np.ctypeslib.as_array(ss.bases, shape=(nb,))
    noffs = np.ctypeslib.as_array(ss.name_offsets, shape=(n + 1,)).copy()
    names_raw = (C.c_char * int(noffs[-1])).from_address(C.cast(ss.names, C.c_void_p).value).raw if n else b""
    names = [names_raw[int(noffs[i]): int(noffs[i + 1]) - 1] for i in range(n)]
    quals = None
    if ss.quals:
        q = (C.c_char * nb).from_address(C.cast(ss.quals, C.c_void_p).value).raw if nb else b""   # (string_at takes a C int: 2 GB at most)
        quals = [q[int(offs[i]): int(offs[i + 1])] for i in range(n)]
    return {"bases": bases, "offsets": offs, "names": names, "quals": quals, "nseq": n}


def _pinned_block(nbytes):
    """A ctypes byte array over rk_host_alloc memory that frees it when the LAST reference to the array goes away: numpy views made
    with np.frombuffer keep the ctypes object alive through their .base chain, so the page-locked memory lives exactly as long as
    any view of it (the DMA engine may still be reading or writing it until then)."""
    import weakref
    lib = load_library()
    ptr = C.c_void_p()
    _chk(lib.rk_host_alloc(max(nbytes, 1), C.byref(ptr)))
    buf = (C.c_uint8 * max(nbytes, 1)).from_address(ptr.value)
    weakref.finalize(buf, lib.rk_host_free, C.c_void_p(ptr.value))
    return buf


class PinnedArray:
    """A numpy array in page-locked host memory from rk_host_alloc: the host entry points read and write such buffers by DMA where
    they lie.  The memory belongs to `.array` (and to every view of it): it is freed when the last of them is collected, not when
    this wrapper is."""

    def __init__(self, shape, dtype):
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        buf = _pinned_block(nbytes)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def pinned_array(shape, dtype):
    return PinnedArray(shape, dtype)


def parse_files(paths):
    """parse_fastas (rkmh.cpp:238-292): FASTA/FASTQ(.gz) files -> dict(bases, offsets, names, quals)."""
    lib = load_library()
    arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
    ss = SeqSet()
    _chk(lib.rk_parse_files(arr, len(paths), C.byref(ss)))
    try:
        return _seqset_to_py(ss)
    finally:
        lib.rk_seqset_free(C.byref(ss))


class Reader:
    """Streaming FASTA/FASTQ(.gz) reader (batches ready for Context.classify)."""

    def __init__(self, path, keep_quals=True):
        self._lib = load_library()
        self._h = C.c_void_p()
        _chk(self._lib.rk_reader_open(os.fsencode(path), C.byref(self._h)))
        if not keep_quals:
            self._lib.rk_reader_set_options(self._h, 1)

    def next_batch(self, max_records=1 << 20, max_bases=1 << 28):
        ss = SeqSet()
        _chk(self._lib.rk_reader_next(self._h, max_records, max_bases, C.byref(ss)))
        try:
            return _seqset_to_py(ss) if ss.nseq else None
        finally:
            self._lib.rk_seqset_free(C.byref(ss))

    def close(self):
        if self._h:
            self._lib.rk_reader_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pool_trim():
    """Returns the parser's parked batch buffers (up to 1.5 GB) to the system (rk_pool_trim)."""
    load_library().rk_pool_trim()


def parse_file_range(path, lo, hi):
    """The records that START in bytes [lo, hi) of a regular uncompressed FASTQ file (rk_reader_open_range): how each rank of a
    multi-process run reads only its own part of the reads.  Returns the same dict as parse_files plus "strict": False means the
    text is not four lines per record, the range boundaries cannot be trusted and the caller must parse the whole file instead.
    Raises RkmhError for input that cannot be split at all (gzip, FASTA, STDIN)."""
    lib = load_library()
    h = C.c_void_p()
    _chk(lib.rk_reader_open_range(os.fsencode(path), int(lo), int(hi), C.byref(h)))
    parts = []
    try:
        while True:
            ss = SeqSet()
            _chk(lib.rk_reader_next(h, 1 << 22, 1 << 30, C.byref(ss)))
            n = ss.nseq
            if n:
                parts.append(_seqset_to_py(ss))
            lib.rk_seqset_free(C.byref(ss))
            if n == 0:
                break
        strict = bool(lib.rk_reader_strict(h))
    finally:
        lib.rk_reader_close(h)
    if not parts:
        return {"bases": np.zeros(16, np.uint8), "offsets": np.zeros(1, np.uint64), "names": [], "quals": [], "nseq": 0, "strict": strict}
    out = parts[0]
    for p in parts[1:]:
        nb = int(out["offsets"][-1])
        out["bases"] = np.concatenate([out["bases"][:nb], p["bases"]])
        out["offsets"] = np.concatenate([out["offsets"], p["offsets"][1:] + np.uint64(nb)])
        out["names"] += p["names"]
        out["quals"] = None if out["quals"] is None or p["quals"] is None else out["quals"] + p["quals"]
        out["nseq"] += p["nseq"]
    out["strict"] = strict
    return out


def format_stream_line(ref_name, read_name, max_shared, diff, min_num, sketch_size, min_matches=-1, min_diff=0):
    lib = load_library()
    cap = len(ref_name) + len(read_name) + 128
    buf = C.create_string_buffer(cap)
    n = lib.rk_format_stream_line(buf, cap, ref_name, read_name, max_shared, diff, min_num, sketch_size,
                                  min_matches, min_diff)
    if n < 0:
        _chk(n)
    return buf.raw[:n]


def device_props(device=0) -> dict:
    """{'compute_units', 'clock_khz', 'l2_bytes', 'hbm_bytes'} of a GPU (hipGetDeviceProperties)."""
    lib = load_library()
    cu, khz, l2, hbm = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
    _chk(lib.rk_device_props(device, C.byref(cu), C.byref(khz), C.byref(l2), C.byref(hbm)))
    return {"compute_units": cu.value, "clock_khz": khz.value, "l2_bytes": l2.value, "hbm_bytes": hbm.value}


class FastqResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("nrec", C.c_int64), ("out4", C.POINTER(C.c_int32)), ("name_off", C.POINTER(C.c_uint32)),
                ("name_len", C.POINTER(C.c_uint32)), ("seq_off", C.POINTER(C.c_uint32)), ("seq_len", C.POINTER(C.c_uint32)),
                ("qual_off", C.POINTER(C.c_uint32))]


class FastaIndex(C.Structure):
    _fields_ = [("status", C.c_int32), ("nseq", C.c_int64), ("offsets", C.POINTER(C.c_uint64)), ("names", C.c_void_p),
                ("name_offsets", C.POINTER(C.c_uint64))]


class FastaLoad:
    """Reference FASTA text stripped on the device (rk_fasta_load_*): put() the raw text of the -r files block by block through a
    FastqSlot's page-locked buffer, finish() returns (status, names, offsets) -- status != 0: parse on the host --, then
    set_references() sketches the packed bases where they lie."""

    def __init__(self, ctx, text_bytes):
        self._lib = load_library()
        self._ctx = ctx
        self._h = C.c_void_p()
        self.text_bytes = text_bytes
        _chk(self._lib.rk_fasta_load_create(ctx._h, text_bytes, C.byref(self._h)))

    def put(self, slot, offset, text: bytes):
        if len(text) > slot.max_bytes:
            raise ValueError("block larger than the slot")
        C.memmove(self._lib.rk_fastq_slot_text(slot._h), text, len(text))
        _chk(self._lib.rk_fasta_load_put(self._h, slot._h, offset, len(text)))

    def put_raw(self, slot, offset, nbytes):
        """The first nbytes of the slot's text_buffer() (filled by the caller) become text[offset ..)."""
        _chk(self._lib.rk_fasta_load_put(self._h, slot._h, offset, nbytes))

    def finish(self, total_bytes=None):
        res = FastaIndex()
        _chk(self._lib.rk_fasta_load_finish(self._h, self.text_bytes if total_bytes is None else total_bytes, C.byref(res)))
        if res.status != 0:
            return int(res.status), [], np.zeros(1, np.uint64)
        n = int(res.nseq)
        offs = np.ctypeslib.as_array(res.offsets, shape=(n + 1,)).copy()
        noff = np.ctypeslib.as_array(res.name_offsets, shape=(n + 1,))
        blob = C.string_at(res.names, int(noff[n]))
        names = [blob[int(noff[i]): int(noff[i + 1]) - 1] for i in range(n)]
        self._total = int(offs[n])
        return 0, names, offs

    def bases(self):
        """The packed bases on the host (rk_fasta_load_get_bases), as the text spells them."""
        out = np.zeros(self._total + 16, dtype=np.uint8)
        _chk(self._lib.rk_fasta_load_get_bases(self._h, out.ctypes.data_as(C.c_void_p)))
        return out[: self._total]

    def set_references(self, ks, sketch_size, max_samples=-1, counter_slots=0):
        arr = (C.c_int * len(ks))(*ks)
        _chk(self._lib.rk_set_references_fasta(self._ctx._h, self._h, arr, len(ks), sketch_size, max_samples, counter_slots))
        self._ctx.sketch_size, self._ctx.ks = sketch_size, _ks(ks)

    def destroy(self):
        if self._h:
            self._lib.rk_fasta_load_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def fastq_cut(text: bytes) -> int:
    """Offset of the last record start in text[1:] under the four-line rule, or -1 (rk_fastq_cut)."""
    b = (C.c_char * len(text)).from_buffer_copy(text)
    return int(load_library().rk_fastq_cut(C.cast(b, C.c_void_p), len(text)))


class LineParts:
    """What does not depend on the read in a stream / classify line (rk_line_parts): reference names with their tab, the eight tails."""

    def __init__(self, ref_names, sketch_size, min_matches, min_diff):
        self._lib = load_library()
        self._h = C.c_void_p()
        blob = b"".join(bytes(n) + b"\0" for n in ref_names)
        offs = np.zeros(len(ref_names) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(n) + 1 for n in ref_names])
        _chk(self._lib.rk_line_parts_create(blob, offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(ref_names), sketch_size, min_matches, min_diff,
                                            C.byref(self._h)))

    def destroy(self):
        if self._h:
            self._lib.rk_line_parts_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class FastqSlot:
    """One block of raw FASTQ text in flight (rk_fastq_slot_*): the text is split into records, checked, packed and classified on the device."""

    def __init__(self, ctx, max_bytes=1 << 26, device_text=False):
        """device_text: RK_SLOT_DEVICE_TEXT -- the block's text stays on the device (blocks come from load_bgzf), only rows and the
        packed names (or, after set_filter_output, the records filter prints) come back."""
        self._lib = load_library()
        self._h = C.c_void_p()
        self.max_bytes = max_bytes
        self.device_text = bool(device_text)
        _chk(self._lib.rk_fastq_slot_create2(ctx._h, max_bytes, 1 if device_text else 0, C.byref(self._h)))

    def set_filter_output(self, min_matches, min_diff):
        _chk(self._lib.rk_fastq_slot_set_filter_output(self._h, min_matches, min_diff))

    def classify(self, text: bytes):
        """Returns (status, rows [n,4] int32, names [n] bytes, seqs [n] bytes); status != 0: the block must be parsed on the host."""
        self.submit(text)
        return self.finish()

    def count(self, text: bytes, counter):
        """Pass 1 of -M on a block (rk_fastq_slot_count): returns (status, records); status != 0: nothing was counted."""
        n = len(text)
        if n > self.max_bytes:
            raise ValueError("block larger than the slot")
        C.memmove(self._lib.rk_fastq_slot_text(self._h), text, n)
        st, nrec = C.c_int32(), C.c_int64()
        _chk(self._lib.rk_fastq_slot_count(self._h, n, counter._h, C.byref(st), C.byref(nrec)))
        return st.value, nrec.value

    # -- the zero-copy forms the one-process-per-GPU front end uses (rkmh_amd/cli.py): the caller reads file bytes straight into the
    # slot's page-locked buffer and the output text is formatted in C from the result where it lies
    def text_buffer(self):
        """The slot's page-locked text buffer as a writable ctypes array (max_bytes + 64 bytes): os.preadv() into it."""
        return (C.c_char * (self.max_bytes + 64)).from_address(self._lib.rk_fastq_slot_text(self._h))

    def load_bgzf(self, bz, b0, b1):
        """Members [b0, b1) of a Bgzf file inflated on the device into this slot (rk_fastq_slot_load_bgzf) -> (status, nbytes, text
        offset); status 1: take the host route.  Follow with classify_raw(nbytes) / count_raw(nbytes, counter)."""
        n, off = C.c_uint64(), C.c_uint64()
        rc = self._lib.rk_fastq_slot_load_bgzf(self._h, bz._h, b0, b1, C.byref(n), C.byref(off))
        if rc < 0:
            _chk(rc)
        return rc, int(n.value), int(off.value)

    def load_gzip(self, gz, call):
        """Stretch number `call` of a Gzip file (an ordinary one-stream .gz) inflated on the device into this slot
        (rk_fastq_slot_load_gzip) -> (status, nbytes, text offset); status 1: the sequential reader continues from the text offset."""
        n, off = C.c_uint64(), C.c_uint64()
        rc = self._lib.rk_fastq_slot_load_gzip(self._h, gz._h, call, C.byref(n), C.byref(off))
        if rc < 0:
            _chk(rc)
        return rc, int(n.value), int(off.value)

    def classify_raw(self, nbytes):
        """rk_fastq_slot_classify on the first nbytes of text_buffer(); returns the FastqResult structure (valid until the next call)."""
        res = FastqResult()
        _chk(self._lib.rk_fastq_slot_classify(self._h, nbytes, C.byref(res)))
        return res

    def count_raw(self, nbytes, counter):
        st, nrec = C.c_int32(), C.c_int64()
        _chk(self._lib.rk_fastq_slot_count(self._h, nbytes, counter._h, C.byref(st), C.byref(nrec)))
        return st.value, nrec.value

    def _out_buffer(self, cap):
        # one growing buffer per slot (a slot serves one thread): no allocation or zero-fill per block
        if getattr(self, "_obuf", None) is None or len(self._obuf) < cap:
            self._obuf = bytearray(cap + cap // 8)
        return (C.c_char * len(self._obuf)).from_buffer(self._obuf)

    def stream_lines(self, parts, res):
        """The stream / classify lines of a classified block (rk_fastq_stream_lines), as bytes."""
        cap = int(self._lib.rk_fastq_stream_lines_bound(parts._h, C.byref(res)))
        buf = self._out_buffer(cap)
        n = self._lib.rk_fastq_stream_lines(parts._h, C.byref(res), self._lib.rk_fastq_slot_spans_base(self._h), buf, len(self._obuf))
        del buf
        if n < 0:
            _chk(int(n))
        return bytes(memoryview(self._obuf)[:n])

    def filter_records(self, res, min_matches, min_diff):
        """filter's records of a classified block (rk_fastq_filter_records), as bytes."""
        cap = int(self._lib.rk_fastq_filter_records_bound(C.byref(res)))
        buf = self._out_buffer(cap)
        n = self._lib.rk_fastq_filter_records(C.byref(res), self._lib.rk_fastq_slot_spans_base(self._h), min_matches, min_diff, buf, len(self._obuf))
        del buf
        if n < 0:
            _chk(int(n))
        return bytes(memoryview(self._obuf)[:n])

    def submit(self, text: bytes):
        """First half (rk_fastq_slot_submit): the upload and the splitting kernels are enqueued; returns at once."""
        n = len(text)
        if n > self.max_bytes:
            raise ValueError("block larger than the slot")
        C.memmove(self._lib.rk_fastq_slot_text(self._h), text, n)
        self._text = text
        _chk(self._lib.rk_fastq_slot_submit(self._h, n))

    def finish(self):
        """Second half (rk_fastq_slot_finish): waits, classifies, collects; same return value as classify()."""
        text = self._text
        res = FastqResult()
        _chk(self._lib.rk_fastq_slot_finish(self._h, C.byref(res)))
        if res.status != 0 or res.nrec == 0:
            return int(res.status), np.zeros((0, 4), np.int32), [], []
        m = int(res.nrec)
        rows = np.ctypeslib.as_array(res.out4, shape=(m, 4)).copy()
        no = np.ctypeslib.as_array(res.name_off, shape=(m,)); nl = np.ctypeslib.as_array(res.name_len, shape=(m,))
        so = np.ctypeslib.as_array(res.seq_off, shape=(m,)); sl = np.ctypeslib.as_array(res.seq_len, shape=(m,))
        names = [text[int(no[i]): int(no[i]) + int(nl[i])] for i in range(m)]
        seqs = [text[int(so[i]): int(so[i]) + int(sl[i])] for i in range(m)]
        qo = np.ctypeslib.as_array(res.qual_off, shape=(m,))
        self.last_quals = [text[int(qo[i]): int(qo[i]) + int(sl[i])] for i in range(m)]  # the quality strings of the block just finished
        return 0, rows, names, seqs

    def destroy(self):
        if self._h:
            self._lib.rk_fastq_slot_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Gzip:
    """An ordinary gzip file -- one deflate stream -- that the device inflates stretch after stretch (rk_gzip_*, rk_gunzip.hip).
    Gzip.open returns None for anything that is not a gzip file."""

    def __init__(self, handle):
        self._lib = load_library()
        self._h = handle

    @staticmethod
    def open(path):
        lib = load_library()
        h = C.c_void_p()
        rc = lib.rk_gzip_open(os.fsencode(path), C.byref(h))
        if rc != 0:
            return None
        return Gzip(h)

    def first_byte(self):
        return int(self._lib.rk_gzip_first_byte(self._h))

    @property
    def text_bytes_hint(self):
        return int(self._lib.rk_gzip_text_bytes_hint(self._h))

    def plan(self, slot_bytes):
        """how many load_gzip calls the file takes with slots of slot_bytes; rewinds the stream"""
        n = int(self._lib.rk_gzip_plan(self._h, slot_bytes))
        if n < 0:
            _chk(-1)
        return n

    def release_device(self):
        self._lib.rk_gzip_release_device(self._h)

    def close(self):
        if self._h:
            self._lib.rk_gzip_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Bgzf:
    """A BGZF (bgzip) file whose members any number of threads may inflate (rk_bgzf_*): the compressed form of the device FASTQ
    front end's input.  Bgzf.open returns None for anything that is not BGZF."""

    def __init__(self, handle):
        self._lib = load_library()
        self._h = handle

    @staticmethod
    def open(path):
        lib = load_library()
        h = C.c_void_p()
        rc = lib.rk_bgzf_open(os.fsencode(path), C.byref(h))
        if rc == -1:         # RK_ERR_ARG: not BGZF
            return None
        _chk(rc)
        return Bgzf(h)

    @property
    def members(self):
        return int(self._lib.rk_bgzf_members(self._h))

    @property
    def text_bytes(self):
        return int(self._lib.rk_bgzf_text_bytes(self._h))

    def text_offset(self, member):
        return int(self._lib.rk_bgzf_text_offset(self._h, member))

    def first_byte(self):
        return int(self._lib.rk_bgzf_first_byte(self._h))

    def plan(self, target_bytes):
        """first members of the jobs of about target_bytes of text each, and the member count last"""
        cap = self.members + 2
        first = (C.c_int64 * cap)()
        n = int(self._lib.rk_bgzf_plan(self._h, target_bytes, first, cap))
        if n < 0:
            _chk(n)
        return [int(first[i]) for i in range(n + 1)]

    def fastq_records(self, b0, b1, dst, cap):
        """the whole FASTQ records that start in members [b0, b1) -> (status, bytes written to dst, their offset in the text);
        status 1: the text does not begin with '@'.  dst: address or ctypes buffer of cap bytes."""
        n, off = C.c_uint64(), C.c_uint64()
        rc = self._lib.rk_bgzf_fastq_records(self._h, b0, b1, dst, cap, C.byref(n), C.byref(off))
        if rc < 0:
            _chk(rc)
        return rc, int(n.value), int(off.value)

    def close(self):
        if self._h:
            self._lib.rk_bgzf_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Counter:
    """HASHTCounter (rkmh.cpp:739): int32 table in HBM, slot = key % slots."""

    def __init__(self, ctx, slots=None, device_ptr=None, compact=False):
        """compact=True: the compact depth map of rk_counter_create_compact (only the slots of index keys; needs the references set
        and min_num bound 0); device_ptr then names zeroed memory of Counter.compact_entries(ctx, slots) int32, or None."""
        self._lib = load_library()
        self._ctx = ctx
        self._h = C.c_void_p()
        if compact:
            _chk(self._lib.rk_counter_create_compact(ctx._h, slots, C.c_void_p(device_ptr) if device_ptr is not None else None, C.byref(self._h)))
        elif device_ptr is not None:
            _chk(self._lib.rk_counter_wrap(ctx._h, C.c_void_p(device_ptr), slots, C.byref(self._h)))
        else:
            _chk(self._lib.rk_counter_create(ctx._h, slots, C.byref(self._h)))

    def increment(self, key):
        _chk(self._lib.rk_counter_increment(self._h, int(key)))

    def get(self, key):
        v = C.c_int32()
        _chk(self._lib.rk_counter_get(self._h, int(key), C.byref(v)))
        return v.value

    def clear(self):
        _chk(self._lib.rk_counter_clear(self._h))

    def add(self, other):
        """self += other (tables of equal size, possibly on two devices of this process): the reduce step of a multi-device -M run."""
        _chk(self._lib.rk_counter_add(self._h, other._h))

    def copy_from(self, other):
        """self = other (the broadcast step)."""
        _chk(self._lib.rk_counter_copy(self._h, other._h))

    def save(self, path, tag=None):
        """tag: bytes from Context.depth_map_tag (provenance of a read-depth map) or None for an untagged file."""
        if tag is None:
            _chk(self._lib.rk_counter_save(self._h, os.fsencode(path)))
        else:
            _chk(self._lib.rk_counter_save_tagged(self._h, os.fsencode(path), C.c_char_p(bytes(tag)), len(tag)))

    def load(self, path, tag=None):
        """Refuses (RkmhError) a file whose provenance tag differs from `tag` (tagged vs untagged included)."""
        if tag is None:
            _chk(self._lib.rk_counter_load(self._h, os.fsencode(path)))
        else:
            _chk(self._lib.rk_counter_load_tagged(self._h, os.fsencode(path), C.c_char_p(bytes(tag)), len(tag)))

    @staticmethod
    def compact_entries(ctx, slots):
        n = C.c_uint64()
        _chk(load_library().rk_counter_compact_entries(ctx._h, slots, C.byref(n)))
        return int(n.value)

    @property
    def slots(self):
        return int(self._lib.rk_counter_slots(self._h))

    @property
    def entries(self):
        return int(self._lib.rk_counter_entries(self._h))

    @property
    def compact(self):
        return bool(self._lib.rk_counter_is_compact(self._h))

    @property
    def device_ptr(self):
        return int(self._lib.rk_counter_device_ptr(self._h))

    def destroy(self):
        if self._h:
            self._lib.rk_counter_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def parse_policy(spec=None, base=None):
    """rk_policy_parse: `spec` (presets default / mash, or fold= / windows= / zero= / mask= / freqmax= / seed=) applied onto `base`
    (default: the build's defaults).  Raises RkmhError on text it does not know."""
    lib = load_library()
    p = Policy()
    if base is None:
        lib.rk_default_policy(C.byref(p))
    else:
        C.memmove(C.byref(p), C.byref(base), C.sizeof(Policy))
    if spec:
        _chk(lib.rk_policy_parse(spec.encode() if isinstance(spec, str) else spec, C.byref(p)))
    return p


def describe_policy(p):
    """rk_policy_describe: the canonical text of a policy, every key spelled out."""
    buf = C.create_string_buffer(160)
    n = load_library().rk_policy_describe(C.byref(p), buf, len(buf))
    if n < 0:
        _chk(n)
    return buf.value.decode()


class Context:
    """One GPU. Methods are named after the reference's functions they replace."""

    def __init__(self, device=0, policy_spec=None, **policy):
        """policy_spec: the text form (`--hash-policy` of the command lines: presets default / mash, key=value; rk_policy_parse),
        applied to the defaults first; keyword arguments then set single fields of the struct."""
        self._lib = load_library()
        p = parse_policy(policy_spec)
        for k, v in policy.items():
            setattr(p, k, v)
        self.policy = p
        self._h = C.c_void_p()
        _chk(self._lib.rk_ctx_create(device, C.byref(p), C.byref(self._h)))
        self.sketch_size = None
        self.ks = None

    def close(self):
        if self._h:
            self._lib.rk_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _chk(self._lib.rk_ctx_synchronize(self._h))

    # ---- inner boundary (mkmh mirrors) -------------------------------------------------------
    def to_upper(self, s: bytes) -> bytes:
        b = C.create_string_buffer(s, len(s))
        _chk(self._lib.rk_to_upper(self._h, b, len(s)))
        return b.raw

    def calc_hashes(self, seq: bytes, ks, counter=None) -> np.ndarray:
        ks = _ks(ks)
        out = _u64p()
        n = C.c_int()
        if counter is None:
            _chk(self._lib.rk_calc_hashes(self._h, seq, len(seq), _p(ks, C.c_int), len(ks), C.byref(out), C.byref(n)))
        else:
            _chk(self._lib.rk_calc_hashes_counted(self._h, seq, len(seq), _p(ks, C.c_int), len(ks), C.byref(out),
                                                  C.byref(n), counter._h))
        r = np.ctypeslib.as_array(out, shape=(max(n.value, 1),))[: n.value].copy()
        self._lib.rk_free(out)
        return r

    def calc_hash(self, kmer: bytes) -> int:
        v = C.c_uint64()
        _chk(self._lib.rk_calc_hash(self._h, kmer, len(kmer), C.byref(v)))
        return int(v.value)

    def minhashes(self, h: np.ndarray, sketch_size: int, counter=None, min_count=0, max_count=0):
        """Returns (mins, sorted_input) -- the reference sorts its input in place."""
        h = np.ascontiguousarray(h, dtype=np.uint64).copy()
        out = _u64p()
        m = C.c_int()
        if counter is None:
            _chk(self._lib.rk_minhashes(self._h, _p(h, C.c_uint64), len(h), sketch_size, C.byref(out), C.byref(m)))
        else:
            _chk(self._lib.rk_minhashes_frequency_filter(self._h, _p(h, C.c_uint64), len(h), sketch_size, C.byref(out),
                                                         C.byref(m), counter._h, min_count, max_count))
        r = np.ctypeslib.as_array(out, shape=(max(sketch_size, 1),))[: m.value].copy()
        self._lib.rk_free(out)
        return r, h

    def mask_by_frequency(self, h: np.ndarray, counter, min_occ: int) -> np.ndarray:
        h = np.ascontiguousarray(h, dtype=np.uint64).copy()
        _chk(self._lib.rk_mask_by_frequency(self._h, _p(h, C.c_uint64), len(h), counter._h, min_occ))
        return h

    def hash_intersection_size(self, a, b) -> int:
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        out = C.c_int()
        _chk(self._lib.rk_hash_intersection_size(self._h, _p(a, C.c_uint64), len(a), _p(b, C.c_uint64), len(b), C.byref(out)))
        return out.value

    def hash_intersection(self, a, a_start, a_len, b, b_start, b_len, sketch_size):
        """mkmh's 7-argument hash_intersection (equiv.hpp:308,340,364): the matches themselves (<= sketch_size)."""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        if a_start < 0 or b_start < 0 or a_start + a_len > len(a) or b_start + b_len > len(b):
            raise ValueError("range outside the array")
        out, n = _u64p(), C.c_int()
        _chk(self._lib.rk_hash_intersection(self._h, _p(a, C.c_uint64), a_start, a_len, _p(b, C.c_uint64), b_start, b_len,
                                            sketch_size, C.byref(out), C.byref(n)))
        r = np.ctypeslib.as_array(out, shape=(max(n.value, 1),))[: n.value].copy()
        self._lib.rk_free(out)
        return r

    # ---- outer boundary (batches) ------------------------------------------------------------
    def hash_batch(self, bases, offsets, ks):
        ks = _ks(ks)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        ho = np.zeros(n + 1, dtype=np.uint64)
        out = _u64p()
        _chk(self._lib.rk_hash_batch(self._h, _p(bases, C.c_uint8), _p(offsets, C.c_uint64), n, _p(ks, C.c_int), len(ks),
                                     C.byref(out), _p(ho, C.c_uint64)))
        tot = int(ho[-1])
        r = np.ctypeslib.as_array(out, shape=(max(tot, 1),))[:tot].copy()
        self._lib.rk_free(out)
        return r, ho

    def sketch_batch(self, bases, offsets, ks, sketch_size):
        ks = _ks(ks)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        sk = np.zeros((n, sketch_size), dtype=np.uint64)
        ln = np.zeros(n, dtype=np.int32)
        _chk(self._lib.rk_sketch_batch(self._h, _p(bases, C.c_uint8), _p(offsets, C.c_uint64), n, _p(ks, C.c_int), len(ks),
                                       sketch_size, _p(sk, C.c_uint64), _p(ln, C.c_int32)))
        return sk, ln

    def set_references(self, bases, offsets, ks, sketch_size, max_samples=None, counter_slots=0, count_distinct=False):
        ks = _ks(ks)
        _chk(self._lib.rk_set_reference_count_mode(self._h, 1 if count_distinct else 0))
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        _chk(self._lib.rk_set_references(self._h, _p(bases, C.c_uint8), _p(offsets, C.c_uint64), len(offsets) - 1,
                                         _p(ks, C.c_int), len(ks), sketch_size,
                                         -1 if max_samples is None else int(max_samples), counter_slots))
        self.sketch_size, self.ks = sketch_size, ks

    def set_reference_sketches(self, sketches, lens, ks, sketch_size):
        ks = _ks(ks)
        sketches = np.ascontiguousarray(sketches, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        _chk(self._lib.rk_set_reference_sketches(self._h, _p(sketches, C.c_uint64), _p(lens, C.c_int32), len(lens),
                                                 _p(ks, C.c_int), len(ks), sketch_size))
        self.sketch_size, self.ks = sketch_size, ks

    def get_reference_sketches(self):
        n = self._lib.rk_num_references(self._h)
        sk = np.zeros((n, self.sketch_size), dtype=np.uint64)
        ln = np.zeros(n, dtype=np.int32)
        _chk(self._lib.rk_get_reference_sketches(self._h, _p(sk, C.c_uint64), _p(ln, C.c_int32)))
        return sk, ln

    def set_depth_filter(self, counter, min_kmer_occ):
        _chk(self._lib.rk_set_depth_filter(self._h, counter._h if counter is not None else None, min_kmer_occ))
        self._depth = counter

    def set_min_num_bound(self, bound):
        """Row field 3 under a depth filter: exact min_num (bound < 0, default) or min(min_num, bound) -- see rkmh_amd.h."""
        _chk(self._lib.rk_set_min_num_bound(self._h, int(bound)))

    def count_batch(self, bases, offsets, counter):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        _chk(self._lib.rk_count_batch(self._h, _p(bases, C.c_uint8), _p(offsets, C.c_uint64), len(offsets) - 1, counter._h))

    def depth_map_tag(self, ks, bases, offsets) -> bytes:
        """Provenance tag of the read-depth map these reads produce under this context's policy (for Counter.save/load)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        kk = (C.c_int * len(ks))(*ks)
        tag = (C.c_uint8 * 128)()
        _chk(self._lib.rk_depth_map_tag(self._h, kk, len(ks), _p(bases, C.c_uint8), _p(offsets, C.c_uint64), len(offsets) - 1, tag))
        return bytes(tag)

    def classify(self, bases, offsets, out=None) -> np.ndarray:
        """main_stream's per-read loop for a host batch -> int32 [n,4] (max_id, max_shared, diff, min_num).
        Page-locked arrays (pinned_array) are read / written by DMA in place; anything else goes through staging buffers."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        if out is None:
            out = np.zeros((n, 4), dtype=np.int32)
        _chk(self._lib.rk_classify_batch(self._h, _p(bases, C.c_uint8), _p(offsets, C.c_uint64), n, _p(out, C.c_int32)))
        return out

    def set_kmer_form(self, enable):
        """Allow (default) or forbid the k-mer-space form for the references set next."""
        _chk(self._lib.rk_set_kmer_form(self._h, 1 if enable else 0))

    def set_kmer_cache(self, path):
        """File that keeps the k-mer enumeration of the next set_references between runs (rk_set_kmer_cache); None: no cache."""
        _chk(self._lib.rk_set_kmer_cache(self._h, os.fsencode(path) if path else None))

    def kmer_cache_state(self):
        """0 none, 1 loaded, 2 enumerated and written, 3 enumerated (file not writable) -- of the last set_references"""
        return int(self._lib.rk_kmer_cache_state(self._h))

    def kmer_form(self):
        """(active, k-mers found by the enumeration): is the k-mer-space form of the fused kernel in use for these references?"""
        n = C.c_uint32()
        r = self._lib.rk_kmer_form(self._h, C.byref(n))
        _chk(min(r, 0))
        return bool(r), n.value

    def classify_groups(self, bases, offsets, argmax_refs):
        """hpv16's per-read loop: (out4 [n,4] over the first argmax_refs references, counts [n, nref - argmax_refs] of the rest)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        out = np.zeros((n, 4), dtype=np.int32)
        tail = np.zeros((n, max(int(self._lib.rk_num_references(self._h)) - argmax_refs, 0)), dtype=np.int32)
        _chk(self._lib.rk_classify_groups_batch(self._h, _p(bases, C.c_uint8), _p(offsets, C.c_uint64), n, argmax_refs,
                                                _p(out, C.c_int32), _p(tail, C.c_int32) if tail.size else None))
        return out, tail

    @property
    def stream(self):
        return int(self._lib.rk_ctx_stream(self._h) or 0)

    def call(self, ref_bases, ref_offsets, read_bases, read_offsets, k, window_len=100):
        """main_call's candidate records (rkmh.cpp:1772-1865): list of dicts, one per passing SNP / deletion k-mer."""
        ref_offsets = np.ascontiguousarray(ref_offsets, dtype=np.uint64)
        read_offsets = np.ascontiguousarray(read_offsets, dtype=np.uint64)
        out = C.POINTER(CallRecord)()
        n = C.c_int64()
        _chk(self._lib.rk_call(self._h, _p(ref_bases, C.c_uint8), _p(ref_offsets, C.c_uint64), len(ref_offsets) - 1,
                               _p(read_bases, C.c_uint8), _p(read_offsets, C.c_uint64), len(read_offsets) - 1, k, window_len,
                               C.byref(out), C.byref(n)))
        recs = [dict(ref=out[i].ref, pos=out[i].pos, alt_depth=out[i].alt_depth, avg_d=out[i].avg_d, depth=out[i].depth,
                     orig=chr(out[i].orig), alt=chr(out[i].alt), kind=out[i].kind) for i in range(n.value)]
        self._lib.rk_free(out)
        return recs

    def classify_device(self, d_bases_ptr, d_offsets_ptr, nreads, d_out_ptr, max_read_len=0, stream=None):
        """Same with inputs resident in HBM (raw device pointers, e.g. torch tensors' data_ptr()).
        stream: a hipStream_t handle (e.g. torch.cuda.current_stream().cuda_stream; 0 = null stream);
        None = the context's own stream."""
        if stream is None:
            stream = self.stream
        _chk(self._lib.rk_classify_batch_device(self._h, C.c_void_p(d_bases_ptr), C.c_void_p(d_offsets_ptr), nreads,
                                                C.c_void_p(d_out_ptr), max_read_len, C.c_void_p(stream)))

    def classify_device_all(self, d_bases_ptr, d_offsets_ptr, nreads, d_out_ptr, max_read_len=0, stream=None):
        """classify_device without flagged rows: reads the fused kernel hands back are answered by the general kernels
        on the resident bases (synchronises the stream)."""
        if stream is None:
            stream = self.stream
        _chk(self._lib.rk_classify_batch_device_all(self._h, C.c_void_p(d_bases_ptr), C.c_void_p(d_offsets_ptr), nreads,
                                                    C.c_void_p(d_out_ptr), max_read_len, C.c_void_p(stream)))

    def count_device(self, d_bases_ptr, d_offsets_ptr, nreads, counter, stream=None):
        if stream is None:
            stream = self.stream
        _chk(self._lib.rk_count_batch_device(self._h, C.c_void_p(d_bases_ptr), C.c_void_p(d_offsets_ptr), nreads,
                                             counter._h, C.c_void_p(stream)))
