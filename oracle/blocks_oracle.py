"""TEST INFRASTRUCTURE ONLY -- CPU restatement (pure Python) of the reference's .blocks reader / writer:
Block::loadBlocks / writeBlocks and the stream operators of Block and Frame (lib/src/assembly/Block.cc:669-690,
737-747, 795-807; lib/src/assembly/Frame.cc:197-222).

PARITY UNPINNED: Frame.hpp pulls in Read.hpp -> google/sparse_hash_map, which is not in the image, so these few
reference lines cannot be compiled here and no golden vectors exist; the format itself is trivial (15 numbers and two
characters per line).  The reference parses with std::istream's formatted extraction; this file restates those rules
(num_get: optional sign, decimal digits, failbit on none or on overflow; unsigned fields accept a minus sign and wrap)
token by token."""

WS = " \t\n\v\f\r"
HEADER = ("# MasterAssemblyID\tMasterContigID\tMasterStrand\tMasterBegin\tMasterEnd\tMasterBlockReadsLength\tMasterReadsLength\t"
          "SlaveAssemblyID\tSlaveContigID\tSlaveStrand\tSlaveBegin\tSlaveEnd\tSlaveBlockReadsLength\tSlaveReadsLength\n")


class _In:
    def __init__(self, text):
        self.t, self.p, self.ok = text, 0, True

    def _skip(self):
        while self.p < len(self.t) and self.t[self.p] in WS:
            self.p += 1

    def integer(self, lo, hi, unsigned=False):
        """operator>>(int32_t&) / (long long&) / (unsigned long long&)"""
        if not self.ok:
            return 0
        self._skip()
        q, neg = self.p, False
        if q < len(self.t) and self.t[q] in "+-":
            neg = self.t[q] == "-"
            q += 1
        d = q
        while q < len(self.t) and self.t[q].isdigit():
            q += 1
        if q == d:
            self.ok = False
            return 0
        self.p = q
        v = int(self.t[d:q])
        if unsigned:
            if v > hi:
                self.ok = False
                return hi
            return (-v) % (hi + 1) if neg else v
        v = -v if neg else v
        if v < lo or v > hi:
            self.ok = False
            return lo if v < lo else hi
        return v

    def char(self):
        if not self.ok:
            return "\0"
        self._skip()
        if self.p >= len(self.t):
            self.ok = False
            return "\0"
        self.p += 1
        return self.t[self.p - 1]


I32 = (-2 ** 31, 2 ** 31 - 1)
I64 = (-2 ** 63, 2 ** 63 - 1)
U64 = 2 ** 64 - 1


def parse_line(line):
    """operator>>(istream&, Block&): returns the block as a dict, or None when the extraction fails"""
    s = _In(line)
    b = {"n_reads": s.integer(*I64)}
    for side in "ms":
        s.integer(*I32)                                   # assembly id, unused
        b[side + "_ctg"] = s.integer(*I32)
        b[side + "_strand"] = s.char()
        b[side + "_begin"] = s.integer(*I32)
        b[side + "_end"] = s.integer(*I32)
        b[side + "_block_reads_len"] = s.integer(0, U64, unsigned=True)
        b[side + "_reads_len"] = s.integer(0, U64, unsigned=True)
    return b if s.ok else None


def load_blocks(text, min_block_size=1):
    """loadBlocks, Block.cc:669-690"""
    out = []
    for line in text.split("\n"):
        if line == "" or line[0] == "#":
            continue
        b = parse_line(line)
        if b is not None and b["n_reads"] >= min_block_size:
            out.append(b)
    return out


def render(blocks):
    """writeBlocks, Block.cc:737-747 with the two operator<<"""
    out = [HEADER]
    for b in blocks:
        f = [str(b["n_reads"])]
        for side in "ms":
            f += ["0", str(b[side + "_ctg"]), b[side + "_strand"], str(b[side + "_begin"]), str(b[side + "_end"]),
                  str(b[side + "_block_reads_len"]), str(b[side + "_reads_len"])]
        out.append("\t".join(f) + "\n")
    return "".join(out)
