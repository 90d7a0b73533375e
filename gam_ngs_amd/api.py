"""Host-side mirror of the reference's operator interface for the alignment path, on top of the C ABI.

Names and argument meaning follow the reference so the parity tests read like its call sites:

  BandedSmithWaterman(band).find_alignment(a, begin_a, end_a, b, begin_b, end_b, force_start, force_end)
      lib/include/alignment/banded_smith_waterman.hpp:66-71
  MyAlignment + first_match_pos / last_match_pos      lib/include/alignment/my_alignment.hpp:65-131
  ABlast(word).findHits(a, a_start, a_end, b, b_start, b_end)   lib/include/alignment/ablast.hpp:116-117
  PctgBuilder.alignMergeBlock(graph, mb)              lib/include/pctg/PctgBuilder.hpp:162
  MergeBlock                                          lib/include/pctg/MergeDescriptor.hpp:40-69

Everything that computes goes through libgamdp.so on the GPU; nothing here falls back to the CPU.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

from . import lib as L

GAP_A, GAP_B, MATCH, MISMATCH = 0, 1, 2, 3
OPS_CHARS = "ABMX"
DEFAULT_BAND_SIZE = 150


def _check(ctx, rc, what):
    if rc != 0:
        msg = L.load_library().gamdp_last_error(ctx.handle) if ctx is not None and ctx.handle else b""
        raise L.GamdpError("%s failed with code %d: %s" % (what, rc, (msg or b"").decode()))


class Context:
    """One GPU + one HIP stream (gamdp_ctx)."""

    def __init__(self, device: int = 0):
        self.lib = L.load_library()
        h = C.c_void_p()
        rc = self.lib.gamdp_ctx_create(device, C.byref(h))
        if rc != 0:
            raise L.GamdpError("gamdp_ctx_create(device=%d) failed with code %d (no gfx950 GPU?); "
                               "libgamdp has no CPU fallback" % (device, rc))
        self.handle = h
        self.device = device

    def set_arena_bytes(self, nbytes: int):
        _check(self, self.lib.gamdp_ctx_set_arena_bytes(self.handle, nbytes), "gamdp_ctx_set_arena_bytes")

    def last_error(self):
        return (self.lib.gamdp_last_error(self.handle) or b"").decode()

    def kernel_time(self, reset=False):
        ms, n = C.c_double(), C.c_uint64()
        self.lib.gamdp_ctx_kernel_time(self.handle, C.byref(ms), C.byref(n), int(reset))
        return ms.value, n.value

    def launch_info(self):
        """The launches of the last gamdp_align_batch call on this context, as dicts (gamdp_ctx_launch_info)."""
        n = C.c_size_t()
        self.lib.gamdp_ctx_launch_info(self.handle, None, 0, C.byref(n))
        arr = (L.LaunchInfo * max(1, n.value))()
        self.lib.gamdp_ctx_launch_info(self.handle, arr, n.value, C.byref(n))
        return [arr[i].as_dict() for i in range(n.value)]

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gamdp_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SequenceSet:
    """RefSequence equivalent: contigs packed (2 bit + N mask) and resident in HBM (gamdp_seqset)."""

    def __init__(self, ctx: Context, seqs: Sequence[bytes], ascii: bool = True):
        self.ctx = ctx
        n = len(seqs)
        self._keep = [bytes(s) for s in seqs]
        arr = (C.c_char_p * max(1, n))(*self._keep)
        lens = (C.c_uint64 * max(1, n))(*[len(s) for s in self._keep])
        h = C.c_void_p()
        rc = ctx.lib.gamdp_seqset_create(ctx.handle, arr, lens, n, int(ascii), C.byref(h))
        _check(ctx, rc, "gamdp_seqset_create")
        self.handle = h
        self.lengths = [len(s) for s in self._keep]

    @classmethod
    def from_fasta(cls, ctx: "Context", path: str):
        """loadSequences equivalent: every record of a FASTA file, in file order (names in .names)."""
        names, codes = load_fasta(path)
        self = cls(ctx, codes, ascii=False)
        self.names = names
        return self

    @classmethod
    def synthetic(cls, ctx: "Context", first_pair: int, n_pairs: int, length: int, stride: int = 1):
        """Pairs first_pair + k*stride, k < n_pairs, of the benchmark generator, built and packed inside the
        library (sequence 2k = master, 2k+1 = slave); no host copy of the bases is kept."""
        self = cls.__new__(cls)
        self.ctx = ctx
        self._keep = []
        h = C.c_void_p()
        _check(ctx, ctx.lib.gamdp_seqset_create_synth_strided(ctx.handle, first_pair, stride, n_pairs, length, C.byref(h)),
               "gamdp_seqset_create_synth_strided")
        self.handle = h
        self.lengths = [ctx.lib.gamdp_seqset_length(h, i) for i in range(2 * n_pairs)]
        return self

    def __len__(self):
        return len(self.lengths)

    def contig(self, idx, rc=False, off=0):
        return Contig(self, idx, rc, off)

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            self.ctx.lib.gamdp_seqset_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiContext:
    """Several GPUs of one node behind one handle (gamdp_multi): the worker pool of ThreadedBuildPctg.cc:143-197.
    `devices` may name a device more than once (one context each)."""

    def __init__(self, devices: Sequence[int]):
        self.lib = L.load_library()
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = self.lib.gamdp_multi_create(arr, len(devices), C.byref(h))
        if rc != 0:
            raise L.GamdpError("gamdp_multi_create(%r) failed with code %d; libgamdp has no CPU fallback" % (list(devices), rc))
        self.handle = h
        self.devices = list(devices)

    def last_error(self):
        return (self.lib.gamdp_multi_last_error(self.handle) or b"").decode()

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gamdp_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiSequenceSet:
    """The same contigs resident on every device of a MultiContext (gamdp_multi_seqset)."""

    def __init__(self, mctx: MultiContext, seqs: Sequence[bytes], ascii: bool = True):
        self.ctx = mctx
        n = len(seqs)
        self._keep = [bytes(s) for s in seqs]
        arr = (C.c_char_p * max(1, n))(*self._keep)
        lens = (C.c_uint64 * max(1, n))(*[len(s) for s in self._keep])
        h = C.c_void_p()
        rc = mctx.lib.gamdp_multi_seqset_create(mctx.handle, arr, lens, n, int(ascii), C.byref(h))
        if rc != 0:
            raise L.GamdpError("gamdp_multi_seqset_create failed with code %d: %s" % (rc, mctx.last_error()))
        self.handle = h
        self.lengths = [len(s) for s in self._keep]

    def __len__(self):
        return len(self.lengths)

    def contig(self, idx, rc=False, off=0):
        return Contig(self, idx, rc, off)

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            self.ctx.lib.gamdp_multi_seqset_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def partition_lpt(weights: Sequence[int], parts: int) -> List[int]:
    """gamdp_partition_lpt: part of every item (deterministic longest-processing-time-first)."""
    n = len(weights)
    w = (C.c_uint64 * max(1, n))(*weights)
    out = (C.c_uint32 * max(1, n))()
    if L.load_library().gamdp_partition_lpt(w, n, parts, out) != 0:
        raise L.GamdpError("gamdp_partition_lpt failed")
    return list(out[:n])


def task_preflight(alen, blen, band, begin_a, end_a, begin_b, end_b, fs=False, fe=False):
    """(status, cells) the batch call settles on before launching anything (gamdp_task_preflight); status ST_OK means
    the call needs the DP."""
    m64 = (1 << 64) - 1
    cells = C.c_uint64()
    st = L.load_library().gamdp_task_preflight(alen, blen, band, begin_a & m64, end_a & m64, begin_b & m64, end_b & m64,
                                               int(fs), int(fe), C.byref(cells))
    return st, cells.value


@dataclass(frozen=True)
class Contig:
    """A view of one sequence of a SequenceSet: optionally reverse-complemented (the reference's
    in-place reverse_complement) and/or a suffix (its chop_begin copy)."""
    seqset: SequenceSet
    idx: int
    rc: bool = False
    off: int = 0

    def size(self):
        return self.seqset.lengths[self.idx] - self.off


@dataclass
class MyAlignment:
    _begin_a: int = 0
    _begin_b: int = 0
    _score: int = 0
    _homology: float = 0.0
    _length: int = 0
    n_match: int = 0
    first_match: tuple = (0, 0)
    first_found: bool = False
    last_match: tuple = (0, 0)
    last_found: bool = False
    status: int = L.ST_EMPTY
    cells: int = 0
    ops: Optional[str] = None  # edit string over "ABMX" (GAP_A, GAP_B, MATCH, MISMATCH) when requested

    def begin_a(self): return self._begin_a
    def begin_b(self): return self._begin_b
    def score(self): return self._score
    def homology(self): return self._homology
    def length(self): return self._length

    def sequence(self):
        if self.ops is None:
            raise L.GamdpError("edit string was not requested (want_ops=False)")
        return [OPS_CHARS.index(ch) for ch in self.ops]

    def key(self):
        return (self.status, self._begin_a, self._begin_b, self._score, self.n_match, self._length,
                self.first_match[0], self.first_match[1], int(self.first_found), self.last_match[0],
                self.last_match[1], int(self.last_found), self._homology)

    @staticmethod
    def from_result(r: "L.Result", ops=None):
        return MyAlignment(r.begin_a, r.begin_b, r.score, r.homology, r.length, r.n_match, (r.first_a, r.first_b),
                           bool(r.first_found), (r.last_a, r.last_b), bool(r.last_found), r.status, r.cells, ops)


def first_match_pos(aln: MyAlignment):
    """(found, (a, b)) -- my_alignment.cc:167-193"""
    return aln.first_found, aln.first_match


def last_match_pos(aln: MyAlignment):
    """(found, (a, b)) -- my_alignment.cc:228-262"""
    return aln.last_found, aln.last_match


class BandedSmithWaterman:
    """BandedSmithWaterman(band): find_alignment for one call, find_alignments for a batch."""

    def __init__(self, ctx: Context, band: int = DEFAULT_BAND_SIZE):
        self.ctx = ctx
        self.band = band

    def find_alignment(self, a: Contig, begin_a, end_a, b: Contig, begin_b, end_b, force_start=False, force_end=False,
                       want_ops=False) -> MyAlignment:
        return self.find_alignments([(a, begin_a, end_a, b, begin_b, end_b, force_start, force_end)], want_ops)[0]

    def find_alignments(self, calls, want_ops=False, bands=None) -> List[MyAlignment]:
        """calls: list of (a, begin_a, end_a, b, begin_b, end_b[, force_start[, force_end]]); all a's
        must come from one SequenceSet and all b's from one SequenceSet."""
        n = len(calls)
        if n == 0:
            return []
        sa, sb = calls[0][0].seqset, calls[0][3].seqset
        tasks = (L.Task * n)()
        m64 = (1 << 64) - 1
        for i, cl in enumerate(calls):
            a, begin_a, end_a, b, begin_b, end_b = cl[:6]
            fs = bool(cl[6]) if len(cl) > 6 else False
            fe = bool(cl[7]) if len(cl) > 7 else False
            if a.seqset is not sa or b.seqset is not sb:
                raise L.GamdpError("all a / all b contigs of one batch must share a SequenceSet")
            t = tasks[i]
            t.a_id, t.b_id, t.a_off, t.b_off = a.idx, b.idx, a.off, b.off
            t.a_rc, t.b_rc, t.force_start, t.force_end = int(a.rc), int(b.rc), int(fs), int(fe)
            t.band = self.band if bands is None else bands[i]
            t.begin_a, t.end_a, t.begin_b, t.end_b = begin_a & m64, end_a & m64, begin_b & m64, end_b & m64
        out = (L.Result * n)()
        ops_struct = None
        # want_ops: one flag for the batch, or one per call (a capacity of 0 = no edit string for that call)
        per_call = list(want_ops) if isinstance(want_ops, (list, tuple)) else [bool(want_ops)] * n
        want_ops = any(per_call)
        if want_ops:
            caps = []
            for i, cl in enumerate(calls):
                band = tasks[i].band
                caps.append(cl[0].size() + cl[3].size() + 2 * band + 64 if per_call[i] else 0)
            offs = [0] * n
            tot = 0
            for i in range(n):
                offs[i] = tot
                tot += caps[i]
            buf = C.create_string_buffer(max(1, tot))
            offs_c = (C.c_uint64 * n)(*offs)
            caps_c = (C.c_uint64 * n)(*caps)
            ops_struct = L.Ops(C.cast(buf, C.c_void_p), offs_c, caps_c)
        if isinstance(self.ctx, MultiContext):
            if want_ops:
                raise L.GamdpError("edit strings are a single-context (test) feature")
            rc = self.ctx.lib.gamdp_multi_align_batch(self.ctx.handle, sa.handle, sb.handle, tasks, n, out)
            if rc != 0:
                raise L.GamdpError("gamdp_multi_align_batch failed with code %d: %s" % (rc, self.ctx.last_error()))
        else:
            rc = self.ctx.lib.gamdp_align_batch(self.ctx.handle, sa.handle, sb.handle, tasks, n, out,
                                                C.byref(ops_struct) if ops_struct else None)
            _check(self.ctx, rc, "gamdp_align_batch")
        res = []
        for i in range(n):
            ops = None
            if want_ops and per_call[i]:
                ops = ""
                if out[i].status == L.ST_OK:
                    raw = buf.raw[offs[i]:offs[i] + out[i].length]
                    ops = "".join(OPS_CHARS[c] for c in raw)
            res.append(MyAlignment.from_result(out[i], ops))
        return res


class ABlast:
    """ABlast(word_size).findHits on code arrays (host function of libgamdp)."""

    def __init__(self, word_size: int = 20):
        self.word_size = word_size
        self.lib = L.load_library()

    def findHits(self, a: bytes, a_start, a_end, b: bytes, b_start, b_end):
        cap = len(a) + 1
        buf = (C.c_uint32 * cap)()
        m64 = (1 << 64) - 1
        n = self.lib.gamdp_find_hits(a, len(a), a_start & m64, a_end & m64, b, len(b), b_start & m64, b_end & m64,
                                     self.word_size, buf, cap)
        if n < 0:
            raise L.GamdpError("gamdp_find_hits failed")
        return list(buf[:n])


def load_fasta(path: str):
    """(names, code arrays) of a FASTA file through the library's loader (reference rules, no GPU needed)."""
    lib = L.load_library()
    h = C.c_void_p()
    rc = lib.gamdp_fasta_open(path.encode(), C.byref(h))
    if rc != 0:
        raise L.GamdpError("gamdp_fasta_open(%s) failed with code %d" % (path, rc))
    try:
        names, codes = [], []
        for i in range(lib.gamdp_fasta_count(h)):
            names.append(lib.gamdp_fasta_name(h, i).decode())
            n = C.c_uint64()
            p = lib.gamdp_fasta_codes(h, i, C.byref(n))
            codes.append(bytes(bytearray(p[:n.value])) if n.value else b"")
        return names, codes
    finally:
        lib.gamdp_fasta_close(h)


def encode(s) -> bytes:
    if isinstance(s, str):
        s = s.encode()
    out = C.create_string_buffer(len(s) + 1)
    L.load_library().gamdp_encode(s, len(s), out)
    return out.raw[:len(s)]


def decode(codes: bytes) -> str:
    out = C.create_string_buffer(len(codes) + 1)
    L.load_library().gamdp_decode(codes, len(codes), out)
    return out.raw[:len(codes)].decode()


def reverse_complement(codes: bytes) -> bytes:
    buf = C.create_string_buffer(codes, max(1, len(codes)))
    L.load_library().gamdp_revcomp(buf, len(codes))
    return buf.raw[:len(codes)]


def synth_pair(k: int, length: int):
    """(master codes, slave codes) of synthetic pair k (gamdp_synth_pair)."""
    m = C.create_string_buffer(length)
    s = C.create_string_buffer(length + length // 8 + 64)
    sl = L.load_library().gamdp_synth_pair(k, length, m, s)
    return m.raw[:length], s.raw[:sl]


@dataclass
class Block:
    """The Block/Frame fields the driver reads."""
    m_begin: int
    m_end: int
    s_begin: int
    s_end: int
    m_strand: str = "+"
    s_strand: str = "+"
    n_reads: int = 1


@dataclass
class MergeBlock:
    """MergeDescriptor.hpp:40-69: inputs (ids, tails, block list) and the fields alignMergeBlock writes."""
    m_id: int
    s_id: int
    blocks: List[Block] = field(default_factory=list)
    m_ltail: bool = False
    m_rtail: bool = False
    s_ltail: bool = False
    s_rtail: bool = False
    # outputs
    align_ok: bool = False
    align_rev: bool = False
    m_start: int = 0
    m_end: int = 0
    s_start: int = 0
    s_end: int = 0
    status: int = L.ST_OK
    coords_set: bool = False
    n_dp: int = 0
    cells: int = 0
    audit: Optional[list] = None


class PctgBuilder:
    """alignMergeBlock for a whole list of merge blocks (the loop of BuildPctgFunctions.cc:82-84)."""

    def __init__(self, ctx: Context, masterRef: SequenceSet, slaveRef: SequenceSet, band: int = DEFAULT_BAND_SIZE):
        self.ctx, self.masterRef, self.slaveRef, self.band = ctx, masterRef, slaveRef, band

    def alignMergeBlock(self, mb: MergeBlock, audit=0):
        self.alignMergeBlocks([mb], audit)
        return mb

    def alignMergeBlocks(self, mbs: List[MergeBlock], audit=0):
        n = len(mbs)
        if n == 0:
            return mbs
        ins = (L.MbIn * n)()
        keep = []
        for i, mb in enumerate(mbs):
            nb = len(mb.blocks)
            arr = (L.BlockC * max(1, nb))()
            for k, b in enumerate(mb.blocks):
                arr[k].m_begin, arr[k].m_end, arr[k].s_begin, arr[k].s_end = b.m_begin, b.m_end, b.s_begin, b.s_end
                arr[k].m_strand, arr[k].s_strand, arr[k].n_reads = b.m_strand.encode(), b.s_strand.encode(), b.n_reads
            keep.append(arr)
            x = ins[i]
            x.m_id, x.s_id = mb.m_id, mb.s_id
            x.m_ltail, x.m_rtail, x.s_ltail, x.s_rtail = int(mb.m_ltail), int(mb.m_rtail), int(mb.s_ltail), int(mb.s_rtail)
            x.n_blocks = nb
            x.blocks = C.cast(arr, C.POINTER(L.BlockC))
        outs = (L.MbOut * n)()
        aud = (L.Result * (n * audit))() if audit else None
        if isinstance(self.ctx, MultiContext):
            rc = self.ctx.lib.gamdp_multi_align_merge_blocks(self.ctx.handle, self.masterRef.handle, self.slaveRef.handle,
                                                             ins, n, self.band, outs, aud, audit)
            if rc != 0:
                raise L.GamdpError("gamdp_multi_align_merge_blocks failed with code %d: %s" % (rc, self.ctx.last_error()))
        else:
            rc = self.ctx.lib.gamdp_align_merge_blocks(self.ctx.handle, self.masterRef.handle, self.slaveRef.handle, ins, n,
                                                       self.band, outs, aud, audit)
            _check(self.ctx, rc, "gamdp_align_merge_blocks")
        for i, mb in enumerate(mbs):
            o = outs[i]
            mb.align_ok, mb.status, mb.coords_set = bool(o.align_ok), o.status, bool(o.coords_set)
            mb.n_dp, mb.cells = o.n_dp, o.cells
            if o.coords_set:  # the reference leaves these untouched otherwise (PctgBuilder.cc:825-829)
                mb.align_rev = bool(o.align_rev)
                mb.m_start, mb.m_end, mb.s_start, mb.s_end = o.m_start, o.m_end, o.s_start, o.s_end
            if audit:
                mb.audit = [MyAlignment.from_result(aud[i * audit + k]) for k in range(min(audit, o.n_dp))]
        return mbs


def load_blocks(path: str, min_block_size: int = 1):
    """Block::loadBlocks (lib/src/assembly/Block.cc:669-690): the blocks of a .blocks file as dicts."""
    lib = L.load_library()
    h = C.c_void_p()
    if lib.gamdp_blocks_open(str(path).encode(), min_block_size, C.byref(h)):
        raise L.GamdpError("cannot read " + str(path))
    try:
        n = lib.gamdp_blocks_count(h)
        p = lib.gamdp_blocks_data(h)
        return [p[i].as_dict() for i in range(n)]
    finally:
        lib.gamdp_blocks_close(h)


def write_blocks(path: str, blocks):
    """Block::writeBlocks (Block.cc:737-747)."""
    lib = L.load_library()
    arr = (L.BlockRec * max(1, len(blocks)))()
    for r, b in zip(arr, blocks):
        for k in L.BlockRec.KEYS:
            v = b[k]
            setattr(r, k, v.encode("latin1") if isinstance(v, str) else v)
    if lib.gamdp_blocks_write(str(path).encode(), arr, len(blocks)):
        raise L.GamdpError("cannot write " + str(path))


def _block_array(blocks):
    arr = (L.BlockRec * max(1, len(blocks)))()
    for r, b in zip(arr, blocks):
        for k in L.BlockRec.KEYS:
            v = b[k]
            setattr(r, k, v.encode("latin1") if isinstance(v, str) else v)
    return arr


def no_blocks_contigs(blocks, n_master: int, n_slave: int):
    """getNoBlocksContigs (Block.cc:810-862): (master flags, slave flags), 1 = no block lies on the contig."""
    lib = L.load_library()
    m, s = (C.c_uint8 * max(1, n_master))(), (C.c_uint8 * max(1, n_slave))()
    rc = lib.gamdp_no_blocks_contigs(_block_array(blocks), len(blocks), n_master, n_slave, m, s)
    if rc:
        raise L.GamdpError("a block names a contig outside the assemblies (the reference exits here)")
    return list(m[:n_master]), list(s[:n_slave])


def no_blocks_after_filter(filtered, n_master: int, n_slave: int, master_nbc, slave_nbc):
    """getNoBlocksAfterFilterContigs (Block.cc:865-925): contigs that lost all their blocks in the coverage filter."""
    lib = L.load_library()
    m, s = (C.c_uint8 * max(1, n_master))(), (C.c_uint8 * max(1, n_slave))()
    mb, sb = (C.c_uint8 * max(1, n_master))(*master_nbc), (C.c_uint8 * max(1, n_slave))(*slave_nbc)
    rc = lib.gamdp_no_blocks_after_filter(_block_array(filtered), len(filtered), n_master, n_slave, mb, sb, m, s)
    if rc:
        raise L.GamdpError("a block names a contig outside the assemblies (the reference exits here)")
    return list(m[:n_master]), list(s[:n_slave])


def write_selected_fasta(assembly, select, path):
    """The contigs with select[i] != 0 as gam-merge writes .noblocks.BF/.AF.fasta and .notmerged.fasta
    (src/Merge.cc:336-373, 414-431): `stream << contig << std::endl`."""
    lib = L.load_library()
    sel = (C.c_uint8 * max(1, len(select)))(*[int(bool(x)) for x in select])
    if lib.gamdp_fasta_write_selected(assembly.handle, sel, str(path).encode()):
        raise L.GamdpError("cannot write " + str(path))

