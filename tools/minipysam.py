"""Column-replay stand-ins for `pysam` and `pybedtools`, used ONLY by tools/make_pileup_goldens.py to drive the
reference's own BaseCellCounter.py / SplitBamCellTypes.py in the build container (neither package is installed,
there is no network).  Test-fixture tooling: nothing under longsom_amd/ imports this file and it never travels as
part of the product path.

What runs for real when the goldens are made: every line of the reference's Python —
  meta_to_dict, split_bam and its report            scripts/PreProcessing/SplitBamCellTypes.py:16-36,39-192
  MakeWindows, run_interval, EasyReadPileup,
  collect_result, concatenate_sort_temp_files...    scripts/SNVCalling/BaseCellCounter.py:12-320,344-409
What is restated here (third-party semantics that are NOT in /root/reference; SURVEY.md §8a rows a4-a5, hand-derived
from htslib sam.c `resolve_cigar2` / `bam_plp_push` and pysam `libcalignedsegment.pyx` / `libcalignmentfile.pyx`):
  * the column iterator behind AlignmentFile.pileup(): stepper "samtools" read filter (flag_filter UNMAP|SECONDARY|
    QCFAIL|DUP, min_mapping_quality, ignore_orphans), per-(read, column) qpos / is_del / is_refskip / indel;
  * PileupColumn accessors with the base-quality skip (pileup_base_qual_skip) and the string form of
    get_query_sequences(add_indels=True);
  * AlignedSegment.opt / flags, AlignmentFile "wb" writer (re-emits the raw records), FastaFile, BedTool.window_maker.
Written independently of longsom_amd/csrc/hostio/bamio.cpp and oracle/plp_oracle.c (per-read cursor objects walked
column by column, python strings built the way pysam builds them), so agreement of the three is a real check.
max_depth is not modelled (fixtures stay far below 200000 except the dedicated cap case, which uses `max_depth`).
"""
import struct
import types
import zlib

_NT16 = "=ACMGRSVTWYHKDBN"
# htslib's resolve_cigar2 changed in 1.11: before it, the LAST column of ANY operation that is followed by a deletion carried
# indel = -(length of that next D) — also the last column of a D operation itself, so inside "1D2D" the first deletion's column is
# printed "*-2NN" (EasyReadPileup: 'D') where htslib >= 1.11 prints "*" ('O').  The reference's environment pins neither pysam nor
# htslib (workflow/envs/SComatic.yaml:9,23: python=3.7 resolves pysam 0.15-0.22, i.e. htslib 1.9-1.18).  Default: >= 1.11.
LEGACY_DEL_MERGE = False
REF_OPS = (0, 2, 3, 7, 8)          # M D N = X consume the reference
MATCH_OPS = (0, 7, 8)


def read_bgzf(path):
    raw = open(path, "rb").read()
    out, off = [], 0
    while off + 18 <= len(raw):
        xlen = struct.unpack_from("<H", raw, off + 10)[0]
        bsize = None
        p = off + 12
        while p < off + 12 + xlen:
            si1, si2, slen = raw[p], raw[p + 1], struct.unpack_from("<H", raw, p + 2)[0]
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", raw, p + 4)[0] + 1
            p += 4 + slen
        data = raw[off + 12 + xlen: off + bsize - 8]
        out.append(zlib.decompress(data, -15) if len(data) else b"")
        off += bsize
    return b"".join(out)


class AlignedSegment:
    """One BAM record: the accessors the reference's scripts use."""

    def __init__(self, raw):
        self.raw = raw                     # record bytes without the leading block_size
        (self.tid, self.pos, l_name, self.mapq, _bin, n_cigar, self.flag, self.l_seq, _mt, _mp, _tl) = struct.unpack_from("<iiBBHHHIiii", raw, 0)
        p = 32
        self.query_name = raw[p:p + l_name - 1].decode(); p += l_name
        self.cigartuples = [(c & 0xF, c >> 4) for c in struct.unpack_from("<%dI" % n_cigar, raw, p)] if n_cigar else None
        p += 4 * n_cigar
        self._seq = raw[p:p + (self.l_seq + 1) // 2]; p += (self.l_seq + 1) // 2
        self._qual = raw[p:p + self.l_seq]; p += self.l_seq
        self._tags = {}
        while p + 3 <= len(raw):
            tag, ty = raw[p:p + 2].decode(), chr(raw[p + 2]); p += 3
            if ty in "Aa":
                self._tags[tag] = chr(raw[p]); p += 1
            elif ty in "cC":
                self._tags[tag] = struct.unpack_from("<b" if ty == "c" else "<B", raw, p)[0]; p += 1
            elif ty in "sS":
                self._tags[tag] = struct.unpack_from("<h" if ty == "s" else "<H", raw, p)[0]; p += 2
            elif ty in "iI":
                self._tags[tag] = struct.unpack_from("<i" if ty == "i" else "<I", raw, p)[0]; p += 4
            elif ty == "f":
                self._tags[tag] = struct.unpack_from("<f", raw, p)[0]; p += 4
            elif ty in "ZH":
                e = raw.index(b"\0", p); self._tags[tag] = raw[p:e].decode(); p = e + 1
            elif ty == "B":
                st, cnt = chr(raw[p]), struct.unpack_from("<I", raw, p + 1)[0]
                p += 5 + cnt * {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[st]
            else:
                raise ValueError("aux type %r" % ty)
        end = self.pos
        for op, ln in self.cigartuples or []:
            if op in REF_OPS:
                end += ln
        self.reference_end = end if end > self.pos else self.pos + 1
        self._k = -1; self._x = 0; self._y = 0      # pileup cursor (op index, reference / query position of the op's start)

    # pysam: KeyError when the tag is absent
    def opt(self, tag):
        return self._tags[tag]

    mapping_quality = property(lambda s: s.mapq)
    is_reverse = property(lambda s: bool(s.flag & 0x10))
    is_secondary = property(lambda s: bool(s.flag & 0x100))
    is_duplicate = property(lambda s: bool(s.flag & 0x400))
    is_supplementary = property(lambda s: bool(s.flag & 0x800))

    def base(self, q):
        return _NT16[(self._seq[q >> 1] >> (0 if q & 1 else 4)) & 0xF]

    def qual(self, q):
        return self._qual[q]

    # ---- htslib resolve_cigar2: place the read on column `pos` --------------------------------------------------
    def resolve(self, pos):
        cg = self.cigartuples
        if self._k < 0:
            self._x, self._y = self.pos, 0
            k = 0
            while k < len(cg) and cg[k][0] not in REF_OPS:
                if cg[k][0] in (1, 4):
                    self._y += cg[k][1]
                k += 1
            self._k = k
        else:
            op, ln = cg[self._k]
            if pos - self._x >= ln:
                if op in MATCH_OPS:
                    self._y += ln
                self._x += ln
                k = self._k + 1
                while k < len(cg) and cg[k][0] not in REF_OPS:
                    if cg[k][0] in (1, 4):
                        self._y += cg[k][1]
                    k += 1
                self._k = k
        op, ln = cg[self._k]
        indel = 0
        if self._x + ln - 1 == pos and self._k + 1 < len(cg):          # last column of this op: what follows?
            op2, l2 = cg[self._k + 1]
            if op2 == 2 and LEGACY_DEL_MERGE:
                indel = -l2                     # htslib <= 1.10: whatever the current operation is, consecutive D's not summed
            elif op2 == 2 and op != 2:
                indel = -l2
                for o3, l3 in cg[self._k + 2:]:
                    if o3 == 2:
                        indel -= l3
                    else:
                        break
            elif op2 == 1:
                indel = l2
                for o3, l3 in cg[self._k + 2:]:
                    if o3 == 1:
                        indel += l3
                    elif o3 != 6:
                        break
            elif op2 == 6 and self._k + 2 < len(cg):
                d = 0
                for o3, l3 in cg[self._k + 2:]:
                    if o3 == 1:
                        d += l3
                    elif o3 in REF_OPS:
                        break
                if d > 0:
                    indel = d
        if op in MATCH_OPS:
            return self._y + (pos - self._x), False, False, indel
        return self._y, True, op == 3, indel


class PileupRead:
    def __init__(self, aln, qpos, is_del, is_refskip, indel):
        self.alignment, self.query_position_or_next, self.is_del, self.is_refskip, self.indel = aln, qpos, is_del, is_refskip, indel


class PileupColumn:
    def __init__(self, tid, pos, entries, min_base_quality):
        self.reference_id, self.pos, self.reference_pos = tid, pos, pos
        self._all = entries
        self._bq = min_base_quality

    def _kept(self):
        for e in self._all:                                           # pileup_base_qual_skip
            q = e.alignment.qual(e.query_position_or_next) if e.query_position_or_next < e.alignment.l_seq else 0
            if q >= self._bq:
                yield e, q

    def get_num_aligned(self):
        return sum(1 for _ in self._kept())

    def get_query_names(self):
        return [e.alignment.query_name for e, _ in self._kept()]

    def get_query_qualities(self):
        return [q for _, q in self._kept()]

    @property
    def pileups(self):
        return [e for e, _ in self._kept()]

    def get_query_sequences(self, mark_matches=False, mark_ends=False, add_indels=False):
        out = []
        for e, _ in self._kept():
            a, rev = e.alignment, bool(e.alignment.flag & 0x10)
            if not e.is_del:
                cc = a.base(e.query_position_or_next) if e.query_position_or_next < a.l_seq else "N"
                s = cc.lower() if rev else cc.upper()                 # no reference attached: mark_matches is inert
            elif e.is_refskip:
                s = "<" if rev else ">"
            else:
                s = "*"
            if add_indels and e.indel > 0:
                ins = "".join(a.base(e.query_position_or_next + 1 + j) if e.query_position_or_next + 1 + j < a.l_seq else "N" for j in range(e.indel))
                s += "+%d%s" % (e.indel, ins.lower() if rev else ins.upper())
            elif add_indels and e.indel < 0:
                s += "-%d%s" % (-e.indel, ("n" if rev else "N") * (-e.indel))
            out.append(s)
        return out


class AlignmentFile:
    """Reader (any mode starting with 'r' or no mode) or writer ('wb' with template=)."""

    def __init__(self, path, mode="rb", template=None):
        self.path, self.mode = path, mode
        if mode.startswith("w"):
            self._header_bytes, self.references, self.lengths, self._out = template._header_bytes, template.references, template.lengths, []
            return
        d = read_bgzf(path)
        assert d[:4] == b"BAM\1", path
        l_text = struct.unpack_from("<I", d, 4)[0]
        p = 8 + l_text
        n_ref = struct.unpack_from("<I", d, p)[0]; p += 4
        self.references, self.lengths = [], []
        for _ in range(n_ref):
            l_name = struct.unpack_from("<I", d, p)[0]
            self.references.append(d[p + 4:p + 4 + l_name - 1].decode()); p += 4 + l_name
            self.lengths.append(struct.unpack_from("<I", d, p)[0]); p += 4
        self._header_bytes = d[:p]
        self._reads = []
        while p + 4 <= len(d):
            bs = struct.unpack_from("<I", d, p)[0]
            self._reads.append(d[p + 4:p + 4 + bs]); p += 4 + bs

    def fetch(self, *a, **k):
        for raw in self._reads:
            r = AlignedSegment(raw)
            if r.tid >= 0:            # fetch() without a region and without until_eof = IteratorRowAllRefs: one index query per
                yield r               # contig, which returns every record PLACED on it (an unmapped mate with coordinates included)

    def write(self, read):
        self._out.append(read.raw)

    def close(self):
        if self.mode.startswith("w") and self._out is not None:
            from tests.support.bamwrite import _bgzf_block
            blob = self._header_bytes + b"".join(struct.pack("<I", len(r)) + r for r in self._out)
            with open(self.path, "wb") as f:
                for i in range(0, len(blob), 0xFF00):
                    f.write(_bgzf_block(blob[i:i + 0xFF00]))
                f.write(_bgzf_block(b""))
            self._out = None

    def pileup(self, contig=None, start=None, stop=None, min_base_quality=13, min_mapping_quality=0, ignore_overlaps=True,
               max_depth=8000, ignore_orphans=True, flag_filter=0x4 | 0x100 | 0x200 | 0x400, truncate=False, **kw):
        tid = self.references.index(contig)
        reads = []
        for raw in self._reads:                                       # the index fetch: records overlapping [start, stop)
            r = AlignedSegment(raw)
            if r.tid != tid or r.flag & 0x4 or r.pos >= stop or r.reference_end <= start:
                continue
            # stepper "samtools" (__advance_samtools) + bam_plp_push
            if r.flag & flag_filter or (ignore_orphans and (r.flag & 0x1) and not (r.flag & 0x2)) or r.mapq < min_mapping_quality:
                continue
            if not r.cigartuples:
                continue
            reads.append(r)
        head, active = 0, []
        pos = 0
        while head < len(reads) or active:
            if not active:
                pos = reads[head].pos
            # htslib bam_plp_push drops a read iff it starts at the iterator's current column (iter->tid == tid && iter->pos == pos)
            # while mp->cnt > maxcnt.  The iterator only reaches column P once a read starting at P has been pushed (bam_plp_next
            # emits columns below max_pos and then waits for more input), so the FIRST read of a start position is never tested;
            # the later ones are, against mp->cnt = buffered reads + the pre-allocated tail node, where the buffer still holds the
            # reads whose last base was column P - 1 (they are freed while column P is swept)
            first_here = True
            while head < len(reads) and reads[head].pos <= pos:
                if reads[head].pos == pos and not first_here and len(active) + 1 > max_depth:
                    head += 1
                    continue
                first_here = False
                active.append(reads[head]); head += 1
            active = [r for r in active if r.reference_end > pos]
            if not active:
                continue
            ents = [PileupRead(r, *r.resolve(pos)) for r in active]
            if not truncate or start <= pos < stop:
                yield PileupColumn(tid, pos, ents, min_base_quality)
            pos += 1


class FastaFile:
    def __init__(self, path):
        self._seq, name = {}, None
        for line in open(path):
            line = line.rstrip("\n")
            if line.startswith(">"):
                name = line[1:].split()[0]; self._seq[name] = []
            elif name is not None:
                self._seq[name].append(line)
        self._seq = {k: "".join(v) for k, v in self._seq.items()}
        self.references = list(self._seq)

    def get_reference_length(self, name):
        return len(self._seq[name])

    def fetch(self, reference=None, start=None, end=None):
        if reference not in self._seq:
            raise KeyError(reference)
        if start is not None and start < 0:
            raise ValueError("start out of range (%i)" % start)
        return self._seq[reference][start:end]

    def close(self):
        pass


class _Interval(tuple):
    chrom = property(lambda s: s[0])
    start = property(lambda s: int(s[1]))
    end = property(lambda s: int(s[2]))


class BedTool:
    """BedTool([(chrom, start, end), ...]) with the three calls MakeWindows makes on the default path."""

    def __init__(self, rows):
        self._rows = [_Interval((str(r[0]), int(r[1]), int(r[2]))) for r in rows]

    def filter(self, fn):
        return BedTool([r for r in self._rows if fn(r)])

    def __iter__(self):
        return iter(self._rows)

    def window_maker(self, b=None, w=None):
        """bedtools makewindows -b <b> -w <w>: [s, s+w), [s+w, s+2w), ... clipped at e; fields come back as strings."""
        out = []
        for c, s, e in (b or self)._rows:
            x = s
            while x < e:
                out.append(_Interval((c, str(x), str(min(x + w, e)))))
                x += w
        wm = BedTool([])
        wm._rows = out
        return wm


def install():
    import sys
    pysam = types.ModuleType("pysam")
    pysam.AlignmentFile, pysam.Samfile, pysam.FastaFile = AlignmentFile, AlignmentFile, FastaFile
    pysam.index = lambda *a, **k: None
    sys.modules["pysam"] = pysam
    pb = types.ModuleType("pybedtools")
    pb.BedTool = BedTool
    sys.modules["pybedtools"] = pb
    return pysam, pb
