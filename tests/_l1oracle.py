"""The CPU oracle's merge-block driver (oracle/gamdp_oracle.c, restating PctgBuilder::alignMergeBlock) on one
tests/_l1cases.py-style scenario."""
import ctypes as C

import _oracle as O


def oracle_mb(sc, band=150, audit_cap=16):
    m, s = O.encode(sc["master"]), O.encode(sc["slave"])
    nb = len(sc["blocks"])
    arr = (O.OracleBlock * max(1, nb))()
    for k, b in enumerate(sc["blocks"]):
        arr[k].m_begin, arr[k].m_end, arr[k].s_begin, arr[k].s_end = b[0], b[1], b[2], b[3]
        arr[k].m_strand, arr[k].s_strand, arr[k].n_reads = b[4].encode(), b[5].encode(), b[6]
    mb = O.OracleMB()
    mb.m_ltail, mb.m_rtail, mb.s_ltail, mb.s_rtail = [int(x) for x in sc["tails"]]
    aud = (O.OracleResult * audit_cap)()
    O.oracle().gamdp_oracle_align_merge_block(m, len(m), s, len(s), arr, nb, band, C.byref(mb), aud, audit_cap)
    return mb, [aud[i].key() for i in range(min(audit_cap, mb.n_dp))]
