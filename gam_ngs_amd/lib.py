"""ctypes binding of libgamdp.so: one Python declaration per symbol of include/gamdp.h."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))

# every function include/gamdp.h declares (checked by tests/test_cabi_symbols.py)
SYMBOLS = [
    "gamdp_ctx_create", "gamdp_ctx_destroy", "gamdp_ctx_set_arena_bytes", "gamdp_last_error", "gamdp_ctx_stream",
    "gamdp_ctx_kernel_time", "gamdp_ctx_launch_info", "gamdp_seqset_create", "gamdp_seqset_destroy", "gamdp_seqset_size",
    "gamdp_seqset_length", "gamdp_align_batch", "gamdp_align_merge_blocks", "gamdp_find_hits", "gamdp_encode",
    "gamdp_decode", "gamdp_revcomp", "gamdp_synth_pair", "gamdp_seqset_create_synth", "gamdp_seqset_create_synth_strided",
    "gamdp_fasta_open", "gamdp_fasta_close", "gamdp_fasta_count", "gamdp_fasta_name", "gamdp_fasta_codes",
    "gamdp_seqset_create_from_fasta",
    "gamdp_fasta_create", "gamdp_merge_lists_prepare", "gamdp_zscore_vote", "gamdp_pctgs_create", "gamdp_pctgs_destroy",
    "gamdp_pctgs_last_error", "gamdp_pctgs_add_graph", "gamdp_pctgs_finish", "gamdp_pctgs_count",
    "gamdp_pctgs_merged_count", "gamdp_pctgs_codes", "gamdp_pctgs_rows", "gamdp_pctgs_contig_use",
    "gamdp_pctgs_write_fasta", "gamdp_pctgs_write_descriptors",
    "gamdp_task_preflight", "gamdp_build_info", "gamdp_ctx_l1_stats",
    "gamdp_multi_create", "gamdp_multi_destroy", "gamdp_multi_size", "gamdp_multi_ctx", "gamdp_multi_last_error",
    "gamdp_multi_seqset_create", "gamdp_multi_seqset_create_from_fasta", "gamdp_multi_seqset_destroy",
    "gamdp_multi_seqset_on", "gamdp_multi_align_batch", "gamdp_multi_align_merge_blocks", "gamdp_partition_lpt",
    "gamdp_blocks_open", "gamdp_blocks_close", "gamdp_blocks_count", "gamdp_blocks_data", "gamdp_blocks_write",
    "gamdp_no_blocks_contigs", "gamdp_no_blocks_after_filter", "gamdp_pctgs_not_merged", "gamdp_fasta_write_selected",
]

EINVAL, ENODEV, ENOMEM, ENOTSUP, EHIP = -1, -2, -3, -4, -5
ST_OK, ST_EMPTY, ST_OUT_OF_RANGE, ST_INVALID = 0, 1, 2, 3
ST_DIAG_RANGE = 9   # diagnostics build only: a packed-f16 block left the exact range (never expected)


class GamdpError(RuntimeError):
    pass


class Task(C.Structure):
    _fields_ = [("a_id", C.c_uint32), ("b_id", C.c_uint32), ("a_off", C.c_uint64), ("b_off", C.c_uint64),
                ("a_rc", C.c_uint8), ("b_rc", C.c_uint8), ("force_start", C.c_uint8), ("force_end", C.c_uint8),
                ("band", C.c_uint32), ("begin_a", C.c_uint64), ("end_a", C.c_uint64), ("begin_b", C.c_uint64),
                ("end_b", C.c_uint64)]


class Result(C.Structure):
    _fields_ = [("begin_a", C.c_uint64), ("begin_b", C.c_uint64), ("score", C.c_int64), ("n_match", C.c_uint64),
                ("length", C.c_uint64), ("first_a", C.c_uint64), ("first_b", C.c_uint64), ("last_a", C.c_uint64),
                ("last_b", C.c_uint64), ("cells", C.c_uint64), ("homology", C.c_double),
                ("first_found", C.c_uint8), ("last_found", C.c_uint8), ("status", C.c_uint8), ("pad_", C.c_uint8 * 5)]

    def key(self):
        """Same tuple layout as the oracle's / golden vectors' keys."""
        return (self.status, self.begin_a, self.begin_b, self.score, self.n_match, self.length, self.first_a,
                self.first_b, self.first_found, self.last_a, self.last_b, self.last_found, self.homology)


class Ops(C.Structure):
    _fields_ = [("ops_buf", C.c_void_p), ("ops_off", C.POINTER(C.c_uint64)), ("ops_cap", C.POINTER(C.c_uint64))]


class BlockC(C.Structure):
    _fields_ = [("m_begin", C.c_int32), ("m_end", C.c_int32), ("s_begin", C.c_int32), ("s_end", C.c_int32),
                ("m_strand", C.c_char), ("s_strand", C.c_char), ("n_reads", C.c_int64)]


class MbIn(C.Structure):
    _fields_ = [("m_id", C.c_int32), ("s_id", C.c_int32), ("m_ltail", C.c_uint8), ("m_rtail", C.c_uint8),
                ("s_ltail", C.c_uint8), ("s_rtail", C.c_uint8), ("n_blocks", C.c_uint32),
                ("blocks", C.POINTER(BlockC))]


class MbOut(C.Structure):
    _fields_ = [("align_ok", C.c_uint8), ("align_rev", C.c_uint8), ("status", C.c_uint8), ("coords_set", C.c_uint8),
                ("m_start", C.c_int32), ("m_end", C.c_int32), ("s_start", C.c_int32), ("s_end", C.c_int32),
                ("n_dp", C.c_uint32), ("cells", C.c_uint64)]


class L1Stats(C.Structure):
    _fields_ = [("merge_blocks", C.c_uint64), ("dp_calls", C.c_uint64), ("cells", C.c_uint64), ("rounds", C.c_uint32),
                ("cohorts", C.c_uint32), ("launches", C.c_uint32), ("pad_", C.c_uint32), ("wall_ms", C.c_double),
                ("gpu_busy_ms", C.c_double), ("kernel_sum_ms", C.c_double), ("host_pending_ms", C.c_double),
                ("host_feed_ms", C.c_double)]


class LaunchInfo(C.Structure):
    """gamdp_launch_info: one kernel launch of the last gamdp_align_batch call, as the library accounts for it."""
    _fields_ = [("kernel", C.c_char * 40), ("n_aware", C.c_uint32), ("tasks_per_wavefront", C.c_uint32), ("tasks", C.c_uint32),
                ("units", C.c_uint32), ("slots", C.c_uint32), ("band_max", C.c_uint32), ("units_dirfree", C.c_uint32),
                ("units_packed_top", C.c_uint32), ("units_packed_top_mixed", C.c_uint32), ("strips", C.c_uint32),
                ("piece", C.c_uint32), ("units_top_wanted", C.c_uint32), ("rounds", C.c_double), ("kernel_ms", C.c_double)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "pad_"}
        d["kernel"] = d["kernel"].decode()
        return d


class MBlock(C.Structure):
    _fields_ = [("m_id", C.c_int32), ("m_start", C.c_int32), ("m_end", C.c_int32), ("s_id", C.c_int32),
                ("s_start", C.c_int32), ("s_end", C.c_int32), ("align_rev", C.c_uint8), ("align_ok", C.c_uint8),
                ("m_ltail", C.c_uint8), ("m_rtail", C.c_uint8), ("s_ltail", C.c_uint8), ("s_rtail", C.c_uint8),
                ("ext_slave_next", C.c_uint8), ("ext_slave_prev", C.c_uint8), ("m_rev", C.c_uint8),
                ("s_rev", C.c_uint8), ("pad_", C.c_uint8 * 2)]


class PctgRow(C.Structure):
    _fields_ = [("start", C.c_int64), ("end", C.c_int64), ("ctg_id", C.c_int32), ("reversed", C.c_uint8),
                ("is_master", C.c_uint8), ("pad_", C.c_uint8 * 2)]


class BlockRec(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("m_block_reads_len", C.c_uint64), ("m_reads_len", C.c_uint64),
                ("s_block_reads_len", C.c_uint64), ("s_reads_len", C.c_uint64), ("m_ctg", C.c_int32),
                ("m_begin", C.c_int32), ("m_end", C.c_int32), ("s_ctg", C.c_int32), ("s_begin", C.c_int32),
                ("s_end", C.c_int32), ("m_strand", C.c_char), ("s_strand", C.c_char), ("pad_", C.c_uint8 * 6)]

    KEYS = ("n_reads", "m_ctg", "m_strand", "m_begin", "m_end", "m_block_reads_len", "m_reads_len",
            "s_ctg", "s_strand", "s_begin", "s_end", "s_block_reads_len", "s_reads_len")

    def as_dict(self):
        d = {k: getattr(self, k) for k in self.KEYS}
        d["m_strand"], d["s_strand"] = d["m_strand"].decode("latin1"), d["s_strand"].decode("latin1")
        return d


REGION_VOTE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32)

_lib = None


def library_path():
    # GAMDP_LIB lets experiments (tools/) load an alternative build; the default is the in-tree library
    return os.environ.get("GAMDP_LIB") or os.path.join(HERE, "libgamdp.so")


def load_library():
    """Loads libgamdp.so (raises GamdpError if it has not been built: there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise GamdpError("%s is missing: build it with `make -C gam_ngs_amd/csrc` "
                         "(or __graft_entry__.build()); there is no CPU fallback" % path)
    lib = C.CDLL(path)
    u64, u32, vp = C.c_uint64, C.c_uint32, C.c_void_p
    lib.gamdp_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.gamdp_ctx_create.restype = C.c_int
    lib.gamdp_ctx_destroy.argtypes = [vp]
    lib.gamdp_ctx_destroy.restype = None
    lib.gamdp_ctx_set_arena_bytes.argtypes = [vp, u64]
    lib.gamdp_last_error.argtypes = [vp]
    lib.gamdp_last_error.restype = C.c_char_p
    lib.gamdp_ctx_stream.argtypes = [vp]
    lib.gamdp_ctx_stream.restype = vp
    lib.gamdp_ctx_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u64), C.c_int]
    lib.gamdp_ctx_launch_info.argtypes = [vp, C.POINTER(LaunchInfo), C.c_size_t, C.POINTER(C.c_size_t)]
    lib.gamdp_seqset_create.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(u64), u32, C.c_int, C.POINTER(vp)]
    lib.gamdp_seqset_destroy.argtypes = [vp]
    lib.gamdp_seqset_destroy.restype = None
    lib.gamdp_seqset_size.argtypes = [vp]
    lib.gamdp_seqset_size.restype = u32
    lib.gamdp_seqset_length.argtypes = [vp, u32]
    lib.gamdp_seqset_length.restype = u64
    lib.gamdp_align_batch.argtypes = [vp, vp, vp, C.POINTER(Task), C.c_size_t, C.POINTER(Result), C.POINTER(Ops)]
    lib.gamdp_align_merge_blocks.argtypes = [vp, vp, vp, C.POINTER(MbIn), C.c_size_t, u32, C.POINTER(MbOut),
                                             C.POINTER(Result), u32]
    lib.gamdp_task_preflight.argtypes = [u64, u64, u32, u64, u64, u64, u64, C.c_int, C.c_int, C.POINTER(u64)]
    lib.gamdp_ctx_l1_stats.argtypes = [vp, C.POINTER(L1Stats)]
    lib.gamdp_build_info.argtypes = []
    lib.gamdp_build_info.restype = C.c_uint
    lib.gamdp_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    lib.gamdp_multi_destroy.argtypes = [vp]
    lib.gamdp_multi_destroy.restype = None
    lib.gamdp_multi_size.argtypes = [vp]
    lib.gamdp_multi_ctx.argtypes = [vp, C.c_int]
    lib.gamdp_multi_ctx.restype = vp
    lib.gamdp_multi_last_error.argtypes = [vp]
    lib.gamdp_multi_last_error.restype = C.c_char_p
    lib.gamdp_multi_seqset_create.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(u64), u32, C.c_int, C.POINTER(vp)]
    lib.gamdp_multi_seqset_create_from_fasta.argtypes = [vp, vp, C.POINTER(vp)]
    lib.gamdp_multi_seqset_destroy.argtypes = [vp]
    lib.gamdp_multi_seqset_destroy.restype = None
    lib.gamdp_multi_seqset_on.argtypes = [vp, C.c_int]
    lib.gamdp_multi_seqset_on.restype = vp
    lib.gamdp_multi_align_batch.argtypes = [vp, vp, vp, C.POINTER(Task), C.c_size_t, C.POINTER(Result)]
    lib.gamdp_multi_align_merge_blocks.argtypes = [vp, vp, vp, C.POINTER(MbIn), C.c_size_t, u32, C.POINTER(MbOut),
                                                   C.POINTER(Result), u32]
    lib.gamdp_partition_lpt.argtypes = [C.POINTER(u64), C.c_size_t, C.c_int, C.POINTER(u32)]
    lib.gamdp_find_hits.argtypes = [C.c_char_p, u64, u64, u64, C.c_char_p, u64, u64, u64, u64, vp, u64]
    lib.gamdp_find_hits.restype = C.c_int64
    lib.gamdp_encode.argtypes = [C.c_char_p, u64, vp]
    lib.gamdp_encode.restype = None
    lib.gamdp_decode.argtypes = [vp, u64, vp]
    lib.gamdp_decode.restype = None
    lib.gamdp_revcomp.argtypes = [vp, u64]
    lib.gamdp_revcomp.restype = None
    lib.gamdp_synth_pair.argtypes = [u64, u64, vp, vp]
    lib.gamdp_synth_pair.restype = u64
    lib.gamdp_seqset_create_synth.argtypes = [vp, u64, u32, u64, C.POINTER(vp)]
    lib.gamdp_seqset_create_synth_strided.argtypes = [vp, u64, u64, u32, u64, C.POINTER(vp)]
    lib.gamdp_fasta_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.gamdp_fasta_close.argtypes = [vp]
    lib.gamdp_fasta_close.restype = None
    lib.gamdp_fasta_count.argtypes = [vp]
    lib.gamdp_fasta_count.restype = u32
    lib.gamdp_fasta_name.argtypes = [vp, u32]
    lib.gamdp_fasta_name.restype = C.c_char_p
    lib.gamdp_fasta_codes.argtypes = [vp, u32, C.POINTER(u64)]
    lib.gamdp_fasta_codes.restype = C.POINTER(C.c_uint8)
    lib.gamdp_seqset_create_from_fasta.argtypes = [vp, vp, C.POINTER(vp)]
    lib.gamdp_fasta_create.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(u64), u32, C.c_int,
                                       C.POINTER(vp)]
    lib.gamdp_merge_lists_prepare.argtypes = [vp, vp, C.POINTER(MBlock), C.POINTER(u32), u32, C.c_uint,
                                              C.POINTER(MBlock), u64, C.POINTER(u32), u32, C.POINTER(u32)]
    lib.gamdp_zscore_vote.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t]
    lib.gamdp_pctgs_create.argtypes = [vp, vp, C.POINTER(vp)]
    lib.gamdp_pctgs_destroy.argtypes = [vp]
    lib.gamdp_pctgs_destroy.restype = None
    lib.gamdp_pctgs_last_error.argtypes = [vp]
    lib.gamdp_pctgs_last_error.restype = C.c_char_p
    lib.gamdp_pctgs_add_graph.argtypes = [vp, C.POINTER(MBlock), C.POINTER(u32), u32, REGION_VOTE_FN, vp]
    lib.gamdp_pctgs_finish.argtypes = [vp]
    lib.gamdp_pctgs_count.argtypes = [vp]
    lib.gamdp_pctgs_count.restype = u32
    lib.gamdp_pctgs_merged_count.argtypes = [vp]
    lib.gamdp_pctgs_merged_count.restype = u32
    lib.gamdp_pctgs_codes.argtypes = [vp, u32, C.POINTER(u64)]
    lib.gamdp_pctgs_codes.restype = C.POINTER(C.c_uint8)
    lib.gamdp_pctgs_rows.argtypes = [vp, u32, C.POINTER(PctgRow), u32]
    lib.gamdp_pctgs_rows.restype = u32
    lib.gamdp_pctgs_contig_use.argtypes = [vp, vp, vp]
    lib.gamdp_pctgs_write_fasta.argtypes = [vp, C.c_char_p]
    lib.gamdp_pctgs_write_descriptors.argtypes = [vp, C.c_char_p]
    lib.gamdp_blocks_open.argtypes = [C.c_char_p, C.c_int64, C.POINTER(vp)]
    lib.gamdp_blocks_close.argtypes = [vp]
    lib.gamdp_blocks_close.restype = None
    lib.gamdp_blocks_count.argtypes = [vp]
    lib.gamdp_blocks_count.restype = u64
    lib.gamdp_blocks_data.argtypes = [vp]
    lib.gamdp_blocks_data.restype = C.POINTER(BlockRec)
    lib.gamdp_blocks_write.argtypes = [C.c_char_p, C.POINTER(BlockRec), u64]
    lib.gamdp_no_blocks_contigs.argtypes = [C.POINTER(BlockRec), u64, u32, u32, vp, vp]
    lib.gamdp_no_blocks_after_filter.argtypes = [C.POINTER(BlockRec), u64, u32, u32, vp, vp, vp, vp]
    lib.gamdp_pctgs_not_merged.argtypes = [vp, vp, vp, vp]
    lib.gamdp_fasta_write_selected.argtypes = [vp, vp, C.c_char_p]
    _lib = lib
    return lib
