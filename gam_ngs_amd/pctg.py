"""Python mirror of the post-alignment stage of gam-merge (host only; thin wrappers over the C ABI of
gam_ngs_amd/csrc/gamdp_pctg.cpp): merge-list surgery, buildPctgs, ids + single-contig pctgs, writers.
Reference: lib/src/pctg/BuildPctgFunctions.cc:86-92, PctgBuilder.cc:102-723, src/Merge.cc:380-465."""
import ctypes as C

from . import lib as L

MB_KEYS = ("m_id", "m_start", "m_end", "s_id", "s_start", "s_end", "align_rev", "align_ok", "m_ltail", "m_rtail",
           "s_ltail", "s_rtail", "ext_slave_next", "ext_slave_prev", "m_rev", "s_rev")
STAGE_ALIGN, STAGE_DIRECTION, STAGE_SORT, STAGE_INCLUSIONS, STAGE_ALL = 1, 2, 4, 8, 15


class Assembly:
    """Host-side RefSequence: names + base codes (A0 T1 C2 G3 N4), from memory or from a FASTA file."""

    def __init__(self, names=None, seqs=None, path=None):
        self.lib = L.load_library()
        h = C.c_void_p()
        if path is not None:
            rc = self.lib.gamdp_fasta_open(str(path).encode(), C.byref(h))
        else:
            n = len(seqs)
            bufs = [bytes(bytearray(s)) for s in seqs]
            arr = (C.c_char_p * n)(*bufs)
            nm = (C.c_char_p * n)(*[x.encode() for x in names])
            lens = (C.c_uint64 * n)(*[len(b) for b in bufs])
            rc = self.lib.gamdp_fasta_create(nm, arr, lens, n, 0, C.byref(h))
        if rc:
            raise L.GamdpError("cannot build the assembly (rc %d)" % rc)
        self.handle = h

    def __len__(self):
        return self.lib.gamdp_fasta_count(self.handle)

    def name(self, i):
        return self.lib.gamdp_fasta_name(self.handle, i).decode()

    def codes(self, i):
        n = C.c_uint64()
        p = self.lib.gamdp_fasta_codes(self.handle, i, C.byref(n))
        return bytes(p[:n.value]) if n.value else b""

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gamdp_fasta_close(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _flatten(lists):
    flat = [b for l in lists for b in l]
    arr = (L.MBlock * max(1, len(flat)))()
    for k, b in enumerate(flat):
        for key in MB_KEYS:
            setattr(arr[k], key, int(b.get(key, 0)))
    sizes = (C.c_uint32 * max(1, len(lists)))(*[len(l) for l in lists])
    return arr, sizes, len(flat)


def _unflatten(arr, sizes, n_lists):
    out, at = [], 0
    for i in range(n_lists):
        out.append([{key: int(getattr(arr[at + k], key)) for key in MB_KEYS} for k in range(sizes[i])])
        at += sizes[i]
    return out


def prepare_merge_lists(master, slave, lists, stages=STAGE_ALL):
    """splitMergeBlocksByAlign / ByDirection / sortMergeBlocksByDirection / splitMergeBlocksByInclusions."""
    lib = L.load_library()
    arr, sizes, n = _flatten(lists)
    cap_b, cap_l = n + 1, n + len(lists) + 1
    out = (L.MBlock * cap_b)()
    out_sizes = (C.c_uint32 * cap_l)()
    n_out = C.c_uint32()
    rc = lib.gamdp_merge_lists_prepare(master.handle, slave.handle, arr, sizes, len(lists), stages, out, cap_b,
                                       out_sizes, cap_l, C.byref(n_out))
    if rc:
        raise L.GamdpError("gamdp_merge_lists_prepare failed (rc %d)" % rc)
    return _unflatten(out, out_sizes, n_out.value)


class PairedContigs:
    """std::list<PairedContig> of one run: add_graph per assembly graph, finish, then read or write."""

    def __init__(self, master, slave):
        self.lib = L.load_library()
        self.master, self.slave = master, slave
        h = C.c_void_p()
        if self.lib.gamdp_pctgs_create(master.handle, slave.handle, C.byref(h)):
            raise L.GamdpError("gamdp_pctgs_create failed")
        self.handle = h

    def add_graph(self, lists, vote=None):
        """vote(m_id, m_start, m_end, s_id, s_start, s_end) -> 0 master / 1 slave (the host's z-score evidence)."""
        arr, sizes, _ = _flatten(lists)
        if vote is None:
            cb = C.cast(None, L.REGION_VOTE_FN)
        else:
            cb = L.REGION_VOTE_FN(lambda user, *a: int(vote(*a)))
        rc = self.lib.gamdp_pctgs_add_graph(self.handle, arr, sizes, len(lists), cb, None)
        if rc:
            raise L.GamdpError("gamdp_pctgs_add_graph: %s (rc %d)" % (self.lib.gamdp_pctgs_last_error(self.handle).decode(), rc))

    def finish(self):
        if self.lib.gamdp_pctgs_finish(self.handle):
            raise L.GamdpError("gamdp_pctgs_finish failed")

    def __len__(self):
        return self.lib.gamdp_pctgs_count(self.handle)

    @property
    def merged(self):
        return self.lib.gamdp_pctgs_merged_count(self.handle)

    def codes(self, i):
        n = C.c_uint64()
        p = self.lib.gamdp_pctgs_codes(self.handle, i, C.byref(n))
        return bytes(p[:n.value]) if n.value else b""

    def rows(self, i):
        n = self.lib.gamdp_pctgs_rows(self.handle, i, None, 0)
        buf = (L.PctgRow * max(1, n))()
        self.lib.gamdp_pctgs_rows(self.handle, i, buf, n)
        return [(r.ctg_id, r.start, r.end, bool(r.reversed), bool(r.is_master)) for r in buf[:n]]

    def contig_use(self):
        m = (C.c_uint8 * max(1, len(self.master)))()
        s = (C.c_uint8 * max(1, len(self.slave)))()
        self.lib.gamdp_pctgs_contig_use(self.handle, m, s)
        return list(m[:len(self.master)]), list(s[:len(self.slave)])

    def not_merged(self, slave_nbc_bf, slave_nbc_af):
        """Merge.cc:416-429: slave contigs in no paired contig and in neither no-blocks set (-> .notmerged.fasta)."""
        n = len(self.slave)
        a, b = (C.c_uint8 * max(1, n))(*slave_nbc_bf), (C.c_uint8 * max(1, n))(*slave_nbc_af)
        out = (C.c_uint8 * max(1, n))()
        if self.lib.gamdp_pctgs_not_merged(self.handle, a, b, out):
            raise L.GamdpError("gamdp_pctgs_not_merged failed")
        return list(out[:n])

    def write_fasta(self, path):
        if self.lib.gamdp_pctgs_write_fasta(self.handle, str(path).encode()):
            raise L.GamdpError("cannot write " + str(path))

    def write_descriptors(self, path):
        if self.lib.gamdp_pctgs_write_descriptors(self.handle, str(path).encode()):
            raise L.GamdpError("cannot write " + str(path))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gamdp_pctgs_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def zscore_vote(master_z, slave_z):
    lib = L.load_library()
    n = len(master_z)
    return lib.gamdp_zscore_vote((C.c_double * max(1, n))(*master_z), (C.c_double * max(1, n))(*slave_z), n)
