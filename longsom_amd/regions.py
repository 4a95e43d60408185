"""Genomic regions as the unit of parallelism of the SNV chain: over GPUs (one rank per GPU) and over time (windows).

The reference fans 50 kb windows out to a process pool, every worker writes `<chrom>__<start>__<end>.BaseCellCounts.temp`
and the parent concatenates the temp files in (chrom string, start) order (workflow/scripts/SNVCalling/BaseCellCounter.py:
12-19 collect_result, :22-79 concatenate_sort_temp_files_and_write, :392-402 the pool).  Here a region is a contiguous range
of 64-position tiles in (contig, position) order, much larger than 50 kb:
  * N GPUs: the decoded reads are cut into N regions of about equal EVENT count; rank r loads the reads that overlap its
    region (reads crossing a boundary are loaded on both sides), lsg_set_region makes every column belong to exactly one
    rank (the analogue of `POS >= START and POS < END`, :200), the rank writes its rows as per-contig piece files and the
    ranks all-gather (RCCL) the candidate rows step 2 and step 3 need; rank 0 concatenates the pieces and runs steps 2-3.
  * one GPU, a BAM larger than HBM: the BAM is streamed in batches (hostio.stream_bam); a window's region ends where the
    next batch's first read starts, reads that reach past that point are carried into the next window.
"""
import os
import re
import shutil
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .engine import ReadRecords

TILE = 64
Pos = Tuple[int, int]          # (tid, 0-based position)


def read_ends(rec: ReadRecords) -> np.ndarray:
    """first reference position after the last pileup column of every read (pos + 1 for a read without columns)"""
    end = rec.read_pos.astype(np.int64) + 1
    if rec.n_segs:
        # a read's segments are adjacent and ascending: its last segment ends the read
        last = np.flatnonzero(np.diff(rec.seg_read.astype(np.int64), append=-1) != 0)
        end[rec.seg_read[last]] = rec.seg_start[last].astype(np.int64) + rec.seg_len[last]
    return end


def read_events(rec: ReadRecords) -> np.ndarray:
    if not rec.n_segs:
        return np.zeros(rec.n_reads, np.int64)
    return np.bincount(rec.seg_read, weights=rec.seg_len, minlength=rec.n_reads).astype(np.int64)


def balanced_boundaries(rec: ReadRecords, n_contigs: int, world: int) -> List[Pos]:
    """world + 1 boundaries (tid, pos) with pos a multiple of 64: region r = [b[r], b[r+1]).  The reads arrive in coordinate
    order; boundary r sits at the start tile of the read at which the running EVENT count passes r/world of the total, so every
    rank walks about the same number of events whatever the expression profile (a chromosome-wise split would leave chr1 and
    chrM on one rank each).  Deterministic in the read arrays: every rank computes the same list."""
    bounds: List[Pos] = [(0, 0)]
    if rec.n_reads and world > 1:
        cum = np.cumsum(read_events(rec) + 24)                       # + the per-read header cost, so that empty reads still count
        tot = int(cum[-1])
        for r in range(1, world):
            i = int(np.searchsorted(cum, tot * r // world, side="left"))
            i = min(i, rec.n_reads - 1)
            b = (int(rec.read_tid[i]), (int(rec.read_pos[i]) // TILE) * TILE)
            bounds.append(max(b, bounds[-1]))
    else:
        bounds += [(n_contigs, 0)] * (world - 1)
    bounds.append((n_contigs, 0))
    return bounds


# ---- a coordinate-sorted BAM cut by its .bai ------------------------------------------------------------------------------------------
BAI_WINDOW = 16384


class BaiPlan:
    """The ranks' regions and file slices from the linear index of a .bai (hostio.read_bai): ioffset[tid][w] = smallest virtual offset
    (BGZF block start << 16 | offset inside the block) of the alignments overlapping the 16 kb window w.  Because the file is sorted by
    start, that offset is also at or before every alignment that overlaps any LATER window: a slice that starts there misses nothing a
    region starting at window w needs.  Regions are balanced by compressed bytes (the offsets themselves), boundaries at window starts
    (multiples of 64, as lsg_set_region wants).  What replaces: the reference's workers fetching their 50 kb window through the index
    (BaseCellCounter.py:190-191)."""

    def __init__(self, lin, n_contigs: int, world: int, file_size: int):
        tid, w, voff = [], [], []
        for t, v in enumerate(lin[:n_contigs]):
            nz = np.nonzero(v)[0]
            tid.append(np.full(len(nz), t, np.int64)); w.append(nz.astype(np.int64)); voff.append(np.asarray(v)[nz].astype(np.uint64))
        self.tid = np.concatenate(tid) if tid else np.zeros(0, np.int64)
        self.w = np.concatenate(w) if w else np.zeros(0, np.int64)
        self.voff = np.concatenate(voff) if voff else np.zeros(0, np.uint64)
        self.key = (self.tid << 32) | self.w
        self.n_contigs, self.world, self.file_size = n_contigs, world, int(file_size)
        coff = (self.voff >> np.uint64(16)).astype(np.int64)
        bounds: List[Pos] = [(0, 0)]
        for r in range(1, world):
            i = int(np.searchsorted(coff, self.file_size * r // world, side="left")) if len(coff) else 0
            b = (int(self.tid[i]), int(self.w[i]) * BAI_WINDOW) if i < len(coff) else (n_contigs, 0)
            bounds.append(max(b, bounds[-1]))
        bounds.append((n_contigs, 0))
        self.bounds = bounds

    def _at_or_after(self, tid: int, w: int) -> Optional[int]:
        i = int(np.searchsorted(self.key, (int(tid) << 32) | int(w), side="left"))
        return i if i < len(self.key) else None

    def start(self, lo: Pos) -> Optional[int]:
        """virtual offset at which the slice of a region starting at lo begins (None: no alignment at or after lo)"""
        i = self._at_or_after(lo[0], lo[1] // BAI_WINDOW)
        return None if i is None else int(self.voff[i])

    def end(self, hi: Pos, attempt: int) -> Optional[int]:
        """a virtual offset expected to lie past the first alignment starting at or after hi (None: go to the end of the file); the
        caller checks (lsg_bam_info.last_key) and comes back with the next attempt when a long alignment made it too short"""
        if hi[0] >= self.n_contigs:
            return None
        i = self._at_or_after(hi[0], hi[1] // BAI_WINDOW + (4 << attempt))
        return None if i is None else int(self.voff[i])


def _bgzf_block_end(mm, coff: int) -> int:
    """end (file offset) of the BGZF block that starts at coff"""
    h = bytes(mm[coff:coff + 18 + 64])
    if len(h) < 18 or h[0] != 31 or h[1] != 139 or not (h[3] & 4):
        raise ValueError("the .bai points at offset %d, which is not a BGZF block: index and BAM do not belong together" % coff)
    xlen = h[10] | (h[11] << 8)
    q = 0
    while q + 4 <= xlen and 12 + q + 4 <= len(h):
        sf = h[12 + q:12 + q + 4]
        slen = sf[2] | (sf[3] << 8)
        if sf[0:2] == b"BC" and slen == 2:
            return coff + (h[12 + q + 4] | (h[12 + q + 5] << 8)) + 1
        q += 4 + slen
    raise ValueError("BGZF block at offset %d has no BC field" % coff)


def _last_key_of_block(mm, v: int) -> Optional[int]:
    """(tid << 32 | pos) of the last record that STARTS in the BGZF block the virtual offset v points into, walking the records from v's
    place in the block (an index entry names a record's start); None when the block cannot be read that way (the device ingest then
    decides, as it always does in the end).  One block inflated with zlib on the host: microseconds, against an ingest repeated on the
    device when the guessed end of a slice stops short of the region's end (a spliced read's intron longer than the guess's margin)."""
    import struct
    import zlib
    try:
        coff, u = v >> 16, v & 0xFFFF
        end = _bgzf_block_end(mm, coff)
        h = bytes(mm[coff:coff + 12])
        xlen = h[10] | (h[11] << 8)
        data = zlib.decompressobj(-15).decompress(bytes(mm[coff + 12 + xlen:end - 8]))
        key = None
        while u + 12 <= len(data):
            bs, tid, pos = struct.unpack_from("<Iii", data, u)
            if bs < 32:
                return None
            key = (1 << 62) if tid < 0 else (tid << 32) | max(pos, 0)
            u += 4 + bs
        return key
    except Exception:                                   # noqa: BLE001 - a guess is all this is
        return None


def ingest_slice(engine, bam: str, plan: BaiPlan, lo: Pos, hi: Pos, barcodes, min_mapq: int):
    """the reads a rank needs for the region [lo, hi), from the slice of the BAM the index points at, ingested on the rank's GPU
    (lsg_load_bam_range).  Returns (info, cb_pass, cb_low) as Engine.load_bam; info["slice_bytes"] = bytes of the file that were read."""
    import mmap
    lo_key, hi_key = (lo[0] << 32) | lo[1], (hi[0] << 32) | hi[1]
    v0 = plan.start(lo) if lo < hi else None
    if v0 is None:
        return None
    with open(bam, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
        c0, u0 = v0 >> 16, v0 & 0xFFFF
        attempt = 0
        while True:
            v1 = plan.end(hi, attempt)
            if v1 is not None:
                # the guessed end is checked on the host before anything goes to the device: the slice's last block must hold a record at or
                # past the region's end (the device ingest verifies the same through info["last_key"] afterwards)
                k_last = _last_key_of_block(mm, v1)
                if k_last is not None and k_last < hi_key:
                    attempt += 1
                    continue
            c1 = len(mm) if v1 is None else _bgzf_block_end(mm, v1 >> 16)
            buf = np.frombuffer(mm, dtype=np.uint8, count=c1 - c0, offset=c0)
            try:
                info, cb_pass, cb_low = engine.load_bam_range(buf, u0, barcodes, min_mapq, lo_key, hi_key)
            except Exception as e:                      # noqa: BLE001 - the same error, with where in the file it happened
                raise type(e)("%s [slice of %s at file offsets %d..%d (first record at +%d of its first block), region %s..%s, attempt %d]"
                              % (e, bam, c0, c1, u0, lo, hi, attempt + 1)) from e
            finally:
                del buf
            if v1 is None or info["last_key"] >= hi_key:
                info["slice_bytes"] = c1 - c0
                info["attempts"] = attempt + 1
                return info, cb_pass, cb_low
            attempt += 1


def reads_overlapping(rec: ReadRecords, lo: Pos, hi: Pos, ends: Optional[np.ndarray] = None) -> np.ndarray:
    """mask of the reads with a pileup column in [lo, hi) — a superset is fine (the device counts only the region's columns)"""
    ends = read_ends(rec) if ends is None else ends
    tid = rec.read_tid.astype(np.int64)
    start_key = (tid << 32) | rec.read_pos.astype(np.int64).clip(0)
    end_key = (tid << 32) | ends
    return (start_key < ((hi[0] << 32) | hi[1])) & (end_key > ((lo[0] << 32) | lo[1]))


# ---- piece files -----------------------------------------------------------------------------------------------------------------
_PIECE = re.compile(r"^(?P<chrom>.+)__(?P<start>\d+)\.(?P<table>[^.]+(?:\.[^.]+)*)\.temp$")


def piece_path(tmp_dir: str, chrom: str, start1: int, table: str) -> str:
    """<tmp>/<chrom>__<start>.<table>.temp — the reference's temp-file naming (BaseCellCounter.py:318-319) minus the end coordinate"""
    return os.path.join(tmp_dir, "%s__%d.%s.temp" % (chrom, start1, table))


def concatenate_pieces(tmp_dir: str, table: str, header: str, out_path: str) -> int:
    """header + the table's pieces in (chrom string, start) order (BaseCellCounter.py:64-70); pieces hold rows only, so this is a
    byte copy.  Returns the number of pieces."""
    found = []
    for name in os.listdir(tmp_dir):
        m = _PIECE.match(name)
        if m and m.group("table") == table:
            found.append((m.group("chrom"), int(m.group("start")), os.path.join(tmp_dir, name)))
    found.sort()
    with open(out_path, "wb") as out:
        out.write(header.encode())
        out.flush()
        for _, _, p in found:
            with open(p, "rb") as f:
                left = os.fstat(f.fileno()).st_size
                try:                                      # the kernel copies file to file (no trip through this process's memory)
                    while left > 0:
                        n = os.sendfile(out.fileno(), f.fileno(), None, min(left, 1 << 30))
                        if n <= 0:
                            raise OSError("sendfile made no progress")
                        left -= n
                except OSError:
                    f.seek(os.fstat(f.fileno()).st_size - left)
                    out.seek(0, os.SEEK_END)
                    shutil.copyfileobj(f, out, 1 << 24)
    return len(found)


# ---- exchange --------------------------------------------------------------------------------------------------------------------
@dataclass
class Comm:
    """The process group of one run (torch.distributed; backend nccl = RCCL over xGMI on the GPU node, gloo in CPU tests)."""
    world: int = 1
    rank: int = 0
    device: Optional[object] = None        # torch.device the collectives' tensors live on (cuda:<local rank> for nccl)
    grouped: bool = False                  # a process group is up (world > 1, or a forced one-rank group)

    @classmethod
    def from_env(cls, device_index: Optional[int] = None) -> "Comm":
        """WORLD_SIZE / RANK / LOCAL_RANK as torch.distributed.run sets them.  Must run BEFORE the first HIP call of the process
        when RCCL is the backend.  LSG_DIST_BACKEND=gloo (+ LSG_DIST_DEVICE) rehearses several ranks on one GPU."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world <= 1 and os.environ.get("LSG_DIST_FORCE") != "1":       # LSG_DIST_FORCE=1: a one-rank group (RCCL on a one-GPU box)
            return cls()
        import torch
        import torch.distributed as dist
        rank = int(os.environ.get("RANK", "0"))
        backend = os.environ.get("LSG_DIST_BACKEND", "nccl")
        local = int(os.environ.get("LSG_DIST_DEVICE", os.environ.get("LOCAL_RANK", "0"))) if device_index is None else device_index
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if backend == "nccl":
            torch.cuda.set_device(local)
        if not dist.is_initialized():
            # a finite time limit on every collective: a rank that dies between two of them (an exception nobody voted on) makes its
            # peers fail after this long instead of waiting for ever (LONGSOM_COLLECTIVE_TIMEOUT_MIN, default 30)
            import datetime
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    timeout=datetime.timedelta(minutes=float(os.environ.get("LONGSOM_COLLECTIVE_TIMEOUT_MIN", "30"))))
        c = cls(world, rank, torch.device("cuda", local) if backend == "nccl" else torch.device("cpu"))
        c.grouped = True
        return c

    @property
    def local_device_index(self) -> int:
        return int(os.environ.get("LSG_DIST_DEVICE", os.environ.get("LOCAL_RANK", "0"))) if self.world > 1 else 0

    def barrier(self) -> None:
        if self.grouped:
            import torch.distributed as dist
            dist.barrier()

    def agree(self, error: Optional[BaseException], what: str) -> None:
        """A vote before the ranks go on: every rank calls it with the exception its own part of `what` raised (or None); when any rank
        failed, EVERY rank raises - the failing ones their own error, the others a RuntimeError naming how many failed - instead of
        the healthy ranks blocking in the next collective for a peer that has left."""
        n_bad = int(self.allreduce_sum(np.asarray([0 if error is None else 1], np.int64))[0])
        if error is not None:
            raise error
        if n_bad:
            raise RuntimeError("%s failed on %d of the %d ranks (their own errors say why)" % (what, n_bad, self.world))

    def allreduce_sum(self, values: np.ndarray) -> np.ndarray:
        """element-wise sum over the ranks of an int64 array (SplitBam's counters of a sharded ingest); every rank gets the result"""
        values = np.ascontiguousarray(values, np.int64)
        if not self.grouped:
            return values
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(values.copy()).to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def allgather_bytes(self, payload: bytes) -> List[bytes]:
        """every rank's payload on every rank: one all-gather of the lengths, one of the padded bytes"""
        if not self.grouped:
            return [payload]
        import torch
        import torch.distributed as dist
        n = torch.tensor([len(payload)], dtype=torch.int64, device=self.device)
        sizes = torch.zeros(self.world, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(sizes, n)
        sizes = sizes.cpu().tolist()
        cap = max(1, max(sizes))
        send = torch.zeros(cap, dtype=torch.uint8, device=self.device)
        if payload:
            send[: len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(self.device)
        recv = torch.zeros(self.world * cap, dtype=torch.uint8, device=self.device)
        dist.all_gather_into_tensor(recv, send)
        recv = recv.cpu().numpy()
        return [recv[r * cap: r * cap + sizes[r]].tobytes() for r in range(self.world)]

    def close(self) -> None:
        if self.grouped:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()


def pack_rows(pieces) -> bytes:
    """{(chrom, start): text (str or bytes) of the rows step 2 keeps} -> bytes for allgather_bytes: a count, per piece
    (len(chrom), start, len(text)) as three little-endian int64, the chrom names, then the texts as they are (no escaping: at
    C2 a rank's kept rows are hundreds of MB)"""
    import struct
    items = sorted(pieces.items())
    enc = [(c.encode(), int(s), t.encode() if isinstance(t, str) else bytes(t)) for (c, s), t in items]
    head = struct.pack("<q", len(enc)) + b"".join(struct.pack("<qqq", len(c), s, len(t)) for c, s, t in enc)
    return head + b"".join(c for c, _, _ in enc) + b"".join(t for _, _, t in enc)


def unpack_rows_bytes(payloads: Sequence[bytes]) -> bytes:
    """all ranks' kept rows in (chrom string, start) order — file order of the step-1 table they were cut from"""
    import struct
    allp = []
    for b in payloads:
        if not b:
            continue
        (n,) = struct.unpack_from("<q", b, 0)
        meta = [struct.unpack_from("<qqq", b, 8 + 24 * i) for i in range(n)]
        at = 8 + 24 * n
        names = []
        for lc, _, _ in meta:
            names.append(b[at:at + lc].decode()); at += lc
        for name, (_, start, lt) in zip(names, meta):
            allp.append((name, start, b[at:at + lt])); at += lt
    allp.sort(key=lambda x: (x[0], x[1]))
    return b"".join(t for _, _, t in allp)


def unpack_rows(payloads: Sequence[bytes]) -> str:
    return unpack_rows_bytes(payloads).decode()
