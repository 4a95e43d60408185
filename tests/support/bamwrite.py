"""Minimal pure-Python BAM writer (BGZF via zlib) for hand-authored fixtures and tiny examples.
The bulk synthetic BAMs come from liblongsom_io.so (lsio_synth_bam); this one favours clarity."""
import struct
import zlib
from typing import Dict, List, Sequence, Tuple

_CIGAR_OPS = "MIDNSHP=X"
_NT16 = "=ACMGRSVTWYHKDBN"


def _bgzf_block(data: bytes) -> bytes:
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    c = comp.compress(data) + comp.flush()
    bsize = len(c) + 25
    return (struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, bsize) + c +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def parse_cigar(s: str) -> List[Tuple[int, int]]:
    ops, num = [], ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            ops.append((_CIGAR_OPS.index(ch), int(num))); num = ""
    return ops


def encode_record(tid: int, pos: int, name: str, flag: int, mapq: int, cigar: str, seq: str, qual: Sequence[int], tags: Dict[str, str]) -> bytes:
    ops = parse_cigar(cigar) if cigar and cigar != "*" else []
    nm = name.encode() + b"\0"
    if seq == "*":
        seq = ""
    pk = bytearray()
    for i in range(0, len(seq), 2):
        hi = _NT16.index(seq[i]); lo = _NT16.index(seq[i + 1]) if i + 1 < len(seq) else 0
        pk.append((hi << 4) | lo)
    q = bytes(qual) if len(qual) == len(seq) else bytes([0xFF] * len(seq))
    aux = b""
    for k, v in tags.items():
        if isinstance(v, int):
            aux += k.encode() + b"i" + struct.pack("<i", v)
        else:
            aux += k.encode() + b"Z" + str(v).encode() + b"\0"
    body = (struct.pack("<iiBBHHHIiii", tid, pos, len(nm), mapq, 4680, len(ops), flag, len(seq), -1, -1, 0) + nm +
            b"".join(struct.pack("<I", (l << 4) | op) for op, l in ops) + bytes(pk) + q + aux)
    return struct.pack("<I", len(body)) + body


def write_bam(path: str, contigs: Sequence[Tuple[str, int]], records: Sequence[dict]) -> None:
    """records: dicts with tid, pos (0-based), name, flag, mapq, cigar, seq, qual (list of ints), tags; must already be
    coordinate sorted."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs)
    out = b"BAM\1" + struct.pack("<I", len(text)) + text.encode() + struct.pack("<I", len(contigs))
    for name, ln in contigs:
        out += struct.pack("<I", len(name) + 1) + name.encode() + b"\0" + struct.pack("<I", ln)
    for r in records:
        out += encode_record(r["tid"], r["pos"], r.get("name", "r"), r.get("flag", 0), r.get("mapq", 60), r["cigar"], r["seq"],
                             r.get("qual", []), r.get("tags", {}))
    with open(path, "wb") as f:
        for i in range(0, len(out), 0xFF00):
            f.write(_bgzf_block(out[i:i + 0xFF00]))
        f.write(_bgzf_block(b""))
