"""The SNV chain as one fused run: BAM + barcodes.tsv + reference in, the rule outputs of
SplitBam -> BaseCellCounter -> MergeCounts -> BaseCellCalling_step1/2/3 out (R:SNVCalling.smk:4-221).

The reads are decoded once and stay in HBM; the per-cell-type split is a table lookup on the device,
the merge is fused into the call kernel; only text formatting and steps 2/3 (candidate rows only) run
on the host.  Every output file has the name and the bytes the reference's scripts give it (except
the wall-clock ##fileDate line).
"""
import json
import os
import sys
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import pandas as pd

from . import calling, hostio, pon, regions, tsvio
from ._lib import CallParams, CountParams
from . import _lib
from .engine import Engine


@dataclass
class SnvParams:
    """config/config.yaml:73-90 (SNVCalling block) + the script defaults the rules do not override."""
    min_mapping_quality: int = 60
    min_bq: int = 20
    min_dp: int = 5
    min_cc: int = 5
    max_depth: int = 200000                # bam.pileup(..., max_depth = 200000), BaseCellCounter.py:191 (hard-coded there)
    min_cell_types: int = 2
    min_cells: int = 5                     # BaseCellCalling.step1.py --min_cells (the PoN rules set 1)
    min_distance: int = 0
    max_gnomad_vaf: float = 0.01
    delta_vaf: float = 0.05
    delta_mcf: float = 0.3
    min_ac_reads: int = 3
    min_ac_cells: int = 2
    clust_dist: int = 10000
    alpha1: float = 0.21356677091082193
    beta1: float = 104.95163748636298
    alpha2: float = 0.2474528917555431
    beta2: float = 162.03696139428595
    reference_gz_compat: bool = False      # True reproduces SURVEY quirk Q1 (.gz position sets read as empty)
    row_digests: bool = False              # also hash the count rows as they come back from the device (SnvOutputs.row_digests; bench.py checks them against the CPU oracle's)

    def count(self) -> CountParams:
        return CountParams.longsom_defaults(min_bq=self.min_bq, min_mq=self.min_mapping_quality, min_dp=self.min_dp, min_cc=self.min_cc,
                                            max_depth=self.max_depth)

    def call(self) -> CallParams:
        return CallParams.longsom_defaults(alpha1=self.alpha1, beta1=self.beta1, alpha2=self.alpha2, beta2=self.beta2,
                                           min_ac_cells=self.min_ac_cells, min_ac_reads=self.min_ac_reads,
                                           min_cell_types=self.min_cell_types, min_cells=self.min_cells)


@dataclass
class SnvOutputs:
    report: str
    counts: Dict[str, str]
    merged: str
    step1: str
    step2: str
    step3: str
    step3_unfiltered: str
    timings: Dict[str, float] = field(default_factory=dict)
    row_digests: Optional[dict] = None     # SnvParams.row_digests: rows, columns and xxhash of (keys, reference bases, counters) per cell type, as tools/oracle_hashes.py writes them
    _pending: list = field(default_factory=list, repr=False, compare=False)
    resident: Optional[dict] = field(default=None, repr=False, compare=False)      # a rank's region and decode summary (sharded two-pass loop)

    def start_background(self, fn) -> None:
        """run fn() (table writers) on a thread of its own; wait_for_tables joins it and re-raises what it raised"""
        import threading
        box = {}

        def work():
            try:
                fn()
            except BaseException as e:                   # noqa: BLE001 — handed to the joining thread
                box["error"] = e
        th = threading.Thread(target=work, name="longsom-tables")
        th.start()
        self._pending.append((th, box))

    def wait_for_tables(self) -> float:
        t0 = time.time()
        pending, self._pending = self._pending, []
        for th, box in pending:
            th.join()
        for th, box in pending:
            if "error" in box:
                raise box["error"]
        return time.time() - t0


def write_report(path: str, report: Dict[str, int], seconds: float) -> None:
    """{id}.report.txt of SplitBamCellTypes (:181-187): one-row tab-separated table."""
    d = dict(report); d["Total_time"] = round(seconds, 2)
    pd.DataFrame([d]).to_csv(path, index=False, sep="\t")


@dataclass
class Resident:
    """One sample decoded once and resident on the GPU: contigs, reference, reads; the barcode table is swapped per pass."""
    engine: Engine
    dec: "hostio.DecodedBam"
    table: "hostio.BarcodeTable"           # the table the BAM was decoded against (dense barcode ids)
    contig_names: List[str]
    seconds: Dict[str, float]


PILEUP_MAX_DEPTH = 200000   # bam.pileup(..., max_depth = 200000), BaseCellCounter.py:191 / HCCVSingleCellGenotype.py:122


def load_sample(bam: str, barcodes_tsv: str, ref_fasta: str, engine: Engine, min_mapq: int, allow_depth_overflow: Optional[bool] = None,
                ingest: Optional[str] = None, count_params: Optional[CountParams] = None, keep_store: bool = True, keep_unlisted: bool = False) -> Resident:
    """count_params: the parameters of the count that follows, when the caller knows them (every fused rule does): the load then makes
    that count in the pass that builds the store (Engine.set_count_at_load) and the first pileup_count under them costs nothing.
    keep_store=False (with count_params): that count is the only one the caller will ask for - the load writes no tile store
    (Engine.set_store_policy); another count or a genotyping pass on these reads then raises.
    keep_unlisted: the reads without a listed barcode stay resident (never counted): the per-cell genotyping's pileup of the unsplit BAM
    holds them in its max_depth buffer (HCCVSingleCellGenotype.py:121-122); the chains that genotype pass True.
    ingest: "device" = the BAM's bytes go to the GPU and are inflated, decoded and laid out there (lsg_load_bam); "host" = the host
    decoder (liblongsom_io) + lsg_load_reads; "auto" (default, or LONGSOM_INGEST) = device, and host for a BAM whose records are not
    aligned to its BGZF blocks (not written by htslib).  Same store, same report either way (tests/test_ingest_gpu.py)."""
    ingest = ingest or os.environ.get("LONGSOM_INGEST", "auto")
    saved = engine.load_settings()                 # (a caller's own engine keeps the load filter / store policy it came with)
    engine.set_count_at_load(count_params)
    engine.set_store_policy(engine.STORE_KEEP if keep_store or count_params is None else engine.STORE_SKIP_WHEN_COUNTED)
    engine.set_keep_unlisted(keep_unlisted)
    old_keep = hostio.set_keep_unlisted(keep_unlisted)
    # a BAM that is counted once under known parameters: the reads that count's own filters would refuse are not stored at all (what
    # SplitBamCellTypes.py:110-113 does to the BAM a rule counts) - every stored read is then admitted, and the load sorts keys alone
    once = count_params is not None and not keep_store and not keep_unlisted
    if once:
        engine.set_load_filter(count_params.min_mq, count_params.flag_exclude, count_params.ignore_orphans)
    try:
        return _load_sample(bam, barcodes_tsv, ref_fasta, engine, min_mapq, ingest)
    finally:
        engine.restore_load_settings(saved)
        hostio.set_keep_unlisted(old_keep)


def _load_sample(bam: str, barcodes_tsv: str, ref_fasta: str, engine: Engine, min_mapq: int, ingest: str) -> Resident:
    t = {}
    t0 = time.time()
    bc = hostio.read_barcodes(barcodes_tsv)
    names_fa, seqs = tsvio.read_fasta(ref_fasta)
    seq_of = dict(zip(names_fa, seqs))
    if ingest in ("device", "auto"):
        names, lens, first = hostio.bam_header(bam)
        for n, l in zip(names, lens):
            if n not in seq_of or len(seq_of[n]) != int(l):
                raise ValueError("contig %s of the BAM header is missing from %s or has another length" % (n, ref_fasta))
        t["header_fasta"] = time.time() - t0
        t0 = time.time()
        engine.set_contigs(lens)
        for tid, n in enumerate(names):
            engine.load_reference(tid, seq_of[n])
        engine.set_barcodes(bc.celltype_of, len(bc.celltype_names))
        engine.set_region()
        t["reference"] = time.time() - t0
        t0 = time.time()
        try:
            info, cb_pass, cb_low = engine.load_bam(bam, bc.barcodes, min_mapq=min_mapq, first_record_offset=first)
        except _lib.LsgError as e:
            if ingest == "device" or "straddle" not in str(e):
                raise
            info = None
        if info is not None:
            rep = {"Total_reads": int(info["total_reads"]), "Pass_reads": int(info["pass_reads"]), "CB_not_found": int(info["cb_not_found"]),
                   "CB_not_matched": int(info["cb_not_matched"])}
            if info["mapq_filtered"]:
                rep["MAPQ"] = int(info["mapq_filtered"])
            dec = hostio.DecodedBam(None, names, np.asarray(lens, np.int64), rep, None, cb_pass, cb_low)
            t["decode"] = time.time() - t0                 # device ingest: H2D + inflate + decode + store build (info has the phases)
            t["load"] = 0.0
            for k, v in info.items():                      # the phases of the device ingest, in seconds like everything else here
                if k.startswith("ms_") and k != "ms_total":
                    t["ingest_" + k[3:]] = float(v) / 1e3
            return Resident(engine, dec, bc, names, t)
        t0 = time.time()
    dec = hostio.decode_bam(bam, bc.barcodes, min_mapq=min_mapq)
    # the pileup is driven by the FASTA's contigs (MakeWindows, BaseCellCounter.py:84-86); BAM tids index dec.contig_names
    contig_names = dec.contig_names
    for n, l in zip(contig_names, dec.contig_len):
        if n not in seq_of or len(seq_of[n]) != int(l):
            raise ValueError("contig %s of the BAM header is missing from %s or has another length" % (n, ref_fasta))
    t["decode"] = time.time() - t0
    t0 = time.time()
    engine.set_contigs(dec.contig_len)
    for tid, n in enumerate(contig_names):
        engine.load_reference(tid, seq_of[n])
    engine.set_barcodes(bc.celltype_of, len(bc.celltype_names))
    engine.set_region()
    engine.load_reads(dec.records)
    t["load"] = time.time() - t0
    return Resident(engine, dec, bc, contig_names, t)


def chain_step1(res: Resident, celltype_of: np.ndarray, celltype_names: List[str], report: Dict[str, int], out_dir: str, sample_id: str,
                params: SnvParams, write_tables: bool = True, background_tables: bool = False, kept_rows: bool = True):
    """SplitBam report -> BaseCellCounter -> MergeCounts -> BaseCellCalling step 1 over the resident reads.  Returns (outputs, text of
    the rows step 2 keeps, call records, timings); write_tables=False keeps everything off the disk except the report."""
    eng, contig_names = res.engine, res.contig_names
    t = dict(res.seconds)
    t0 = time.time()
    eng.set_barcodes(celltype_of, len(celltype_names))
    n_rows, n_cols = eng.pileup_count(params.count())
    n_sites, n_cand = eng.call_step1(params.call())
    t["gpu_count_call"] = time.time() - t0
    t0 = time.time()
    d = {k: os.path.join(out_dir, k) for k in ("SplitBam", "BaseCellCounter/" + sample_id, "MergeCounts", "BaseCellCalling")}
    for p in d.values():
        os.makedirs(p, exist_ok=True)
    out = SnvOutputs(report=os.path.join(d["SplitBam"], sample_id + ".report.txt"), counts={}, merged="", step1="", step2="", step3="",
                     step3_unfiltered="")
    write_report(out.report, report, t["decode"])
    if not write_tables:
        calls = eng.fetch_calls(candidates_only=True)
        t["fetch"] = time.time() - t0
        return out, None, calls, t
    date = tsvio.file_date()
    for ct, name in enumerate(celltype_names):
        out.counts[name] = os.path.join(d["BaseCellCounter/" + sample_id], "%s.%s.tsv" % (sample_id, name))
    out.merged = os.path.join(d["MergeCounts"], sample_id + ".BaseCellCounts.AllCellTypes.tsv")
    out.step1 = os.path.join(d["BaseCellCalling"], sample_id + ".calling.step1.tsv")
    if os.environ.get("LONGSOM_HOST_TABLES", "0") != "1":
        return _device_tables(eng, out, contig_names, celltype_names, sample_id, params, date, background_tables, t, t0, (n_rows, n_cols, n_sites, n_cand), kept_rows)
    import threading
    per_ct: List = [None] * len(celltype_names)
    arrived = [threading.Event() for _ in celltype_names]          # a cell type's count rows are on the host
    failed: List[BaseException] = []                               # ... or will never be

    def count_tables():
        for ct, name in enumerate(celltype_names):
            arrived[ct].wait()
            if failed:
                return
            t1 = time.time()
            tsvio.write_counts_tsv(out.counts[name], *per_ct[ct], contig_names, "%s.%s" % (sample_id, name), date)
            t["table_counts_" + name] = time.time() - t1        # (seconds of the writer itself, in the background when background_tables)

    def merged_table():
        if failed:
            return
        t1 = time.time()
        tsvio.write_merged_tsv(out.merged, per_ct, contig_names, celltype_names, date)
        t["table_merged"] = time.time() - t1
    try:
        for ct in range(len(celltype_names)):
            per_ct[ct] = eng.fetch_counts(ct)
            arrived[ct].set()
            if ct == 0 and background_tables:
                # the per-cell-type and the merged table need the count rows only: their writer starts with the first cell type's rows, while
                # the others' and the call records are still on their way from the device (the merged table behind the count tables on ONE
                # thread: a third writer beside steps 2 and 3 made every one of them slower - the host's threads are all busy - and the run
                # no shorter)
                out.start_background(lambda: (count_tables(), merged_table()))
    except BaseException as e:                                   # (the writer must not wait for rows that will never come)
        failed.append(e)
        for ev in arrived:
            ev.set()
        raise
    calls = eng.fetch_calls()
    t["fetch"] = time.time() - t0
    if params.row_digests:
        t0 = time.time()
        out.row_digests = _row_digests(eng, len(celltype_names), (n_rows, n_cols, n_sites, n_cand), per_ct)
        t["row_digests"] = time.time() - t0
    t0 = time.time()
    header = [l + "\n" for l in tsvio.merged_header(celltype_names, date).split("\n") if l.startswith("##")]
    if background_tables:
        # Steps 2 and 3 only need the rows step 2 keeps: those are formatted first (a third of the step-1 table's rows, nothing written);
        # the per-cell-type and merged tables and the step-1 table itself are written by threads of their own (native writers, no GIL)
        # beside steps 2 and 3 - the box's disk takes several files at once faster than one: the caller joins them (SnvOutputs.wait_for_tables)
        s1 = tsvio.step1_kept_rows(calls, per_ct, contig_names, celltype_names, header, as_bytes=True)
        def step1_table():
            t1 = time.time()
            tsvio.write_step1_tsv(out.step1, calls, per_ct, contig_names, celltype_names, header, collect=False)
            t["table_step1"] = time.time() - t1
        out.start_background(step1_table)
        t["write_tables"] = time.time() - t0          # (what the chain waited for: the kept rows; tables_wait is what was left of the writers at the end)
        return out, s1, calls, t
    count_tables()
    merged_table()
    s1 = tsvio.write_step1_tsv(out.step1, calls, per_ct, contig_names, celltype_names, header, as_bytes=True)      # s1 = header + the rows step 2 keeps
    t["write_tables"] = time.time() - t0
    return out, s1, calls, t


def _row_digests(eng, n_ct: int, shape, per_ct=None) -> dict:
    """SnvParams.row_digests: rows, columns and xxhash of (keys, reference bases, counters) per cell type, as tools/oracle_hashes.py writes them"""
    import xxhash
    n_rows, n_cols, n_sites, n_cand = shape
    out = {"rows": [int(x) for x in n_rows], "columns": int(n_cols), "merged_sites": int(n_sites), "candidate_rows": int(n_cand)}
    for ct in range(n_ct):
        rows = per_ct[ct] if per_ct is not None else eng.fetch_counts(ct)
        out["ct%d" % ct] = [xxhash.xxh64(np.ascontiguousarray(x).tobytes() if x.size < (1 << 20) else memoryview(np.ascontiguousarray(x)).cast("B")).hexdigest() for x in rows]
    return out


def _device_tables(eng, out: "SnvOutputs", contig_names, celltype_names, sample_id, params, date, background: bool, t, t0, shape, kept_rows: bool = True):
    """The tables of chain_step1 printed on the device (Engine.format_table: csrc/tables.hip, byte for byte the host writers' text) and
    streamed into their files (Engine.append_table), one thread per file; the count rows never come to the host.  The rows step 2 keeps
    come first - steps 2 and 3 wait for nothing else.  LONGSOM_HOST_TABLES=1 keeps the host writers (csrc/hostio/tsvwrite.cpp)."""
    import threading
    eng.set_table_names(contig_names, celltype_names)
    header = [l + "\n" for l in tsvio.merged_header(celltype_names, date).split("\n") if l.startswith("##")]
    head1 = tsvio.step1_header(header, celltype_names)
    s1 = head1.encode()
    if kept_rows:                                    # (kept_rows=False: the caller runs step 2 on the device too and wants the header alone)
        n = eng.format_table(eng.TABLE_STEP1_KEPT)
        s1 = eng.table_bytes(eng.TABLE_STEP1_KEPT, n, prefix=s1)
        eng.free_table(eng.TABLE_STEP1_KEPT)
    t["fetch"] = time.time() - t0                    # (what came to the host: the kept rows' text)
    t0 = time.time()
    files = [(ct, out.counts[name], tsvio.counts_header("%s.%s" % (sample_id, name), date)) for ct, name in enumerate(celltype_names)]
    files += [(eng.TABLE_MERGED, out.merged, tsvio.merged_header(celltype_names, date)), (eng.TABLE_STEP1, out.step1, head1)]
    sizes = {}
    for table, path, head in files:
        with open(path, "w") as f:
            f.write(head)
        sizes[table] = eng.format_table(table)
    t["format_tables"] = time.time() - t0
    t0 = time.time()

    def stream(table, path):
        t1 = time.time()
        eng.append_table(table, path)
        eng.free_table(table)
        t["table_%d" % table] = time.time() - t1
    if background:
        for table, path, _ in files:                  # (the box's file system takes several files at once faster than one)
            out.start_background(lambda table=table, path=path: stream(table, path))
    else:
        for table, path, _ in files:
            stream(table, path)
    t["write_tables"] = time.time() - t0
    if params.row_digests:
        t1 = time.time()
        out.row_digests = _row_digests(eng, len(celltype_names), shape)
        t["row_digests"] = time.time() - t1
    calls = None if background else eng.fetch_calls()
    return out, s1, calls, t


def run_chain(res: Resident, celltype_of: np.ndarray, celltype_names: List[str], report: Dict[str, int], out_dir: str, sample_id: str,
              params: SnvParams, editing: Optional[str] = None, pon_sr: Optional[str] = None, pon_lr: Optional[str] = None,
              gnomad_af_json: Optional[str] = None, step3: bool = True) -> SnvOutputs:
    """SplitBam report -> BaseCellCounter -> MergeCounts -> BaseCellCalling step 1-3 for one barcode -> cell-type table over the
    resident reads (celltype_of[barcode id] = index into celltype_names, 255 = barcode not listed).  step3=False stops after
    step 2, as pass 1 of the reference does (rules/CellTypeReannotation.smk has no step-3 rule: HCCV reads calling.step2.tsv)."""
    eng, contig_names = res.engine, res.contig_names
    # Step 2 without a gnomAD source and with --min_distance 0 (LongSom's own setting) tags rows by position sets and blanks NA cells: the
    # device prints that table itself and tells step 3 what it needs of it (_device_steps23); any other step 2 runs on the host over the
    # text of the rows it keeps.  LONGSOM_HOST_TABLES=1 / LONGSOM_HOST_STEP2=1 keep the host paths.
    device23 = (os.environ.get("LONGSOM_HOST_TABLES", "0") != "1" and os.environ.get("LONGSOM_HOST_STEP2", "0") != "1" and not gnomad_af_json
                and int(params.min_distance) == 0 and len(set(contig_names)) == len(contig_names)
                and not any(ch in n for n in list(contig_names) + list(celltype_names) for ch in "\t\n"))
    out, s1, _, t = chain_step1(res, celltype_of, celltype_names, report, out_dir, sample_id, params, background_tables=True, kept_rows=not device23)
    try:
        if device23:
            return _device_steps23(out, s1, t, eng, contig_names, out_dir, sample_id, params, editing, pon_sr, pon_lr, step3)
        return _chain_steps23(out, s1, t, eng, contig_names, out_dir, sample_id, params, editing, pon_sr, pon_lr, gnomad_af_json, step3)
    except BaseException:
        try:                                   # (the writers stream from the engine: nobody may close it under them)
            out.wait_for_tables()
        except BaseException:                  # noqa: BLE001 - the first error is the one to report
            pass
        raise


def _device_steps23(out, head1: bytes, t, eng, contig_names, out_dir, sample_id, params, editing, pon_sr, pon_lr, step3):
    """Steps 2 and 3 of run_chain with the step-2 table printed on the device (Engine.TABLE_STEP2: csrc/tables.hip): its text goes from
    the device straight into its file; the host sees the kinds of cell of its columns and the rows step 3 can keep
    (Engine.step2_summary), which is all calling.step3 reads of a table this large."""
    t0 = time.time()
    keys = [calling.read_posset_keys(p, contig_names, params.reference_gz_compat) for p in (editing, pon_sr, pon_lr)]
    for kind, k in zip((calling.KIND_EDITING, calling.KIND_PON_SR, calling.KIND_PON_LR), keys):
        eng.load_posset(kind, k)
    # the table's head as step 2 writes it: the comment lines, then the column header (calling._step2_scanned)
    lines = [l for l in head1.split(b"\n") if l]
    hdr = b"\n".join([l for l in lines if b"#CHROM" not in l] + [[l for l in lines if b"#CHROM" in l][-1]]) + b"\n"
    cols = hdr.decode().split("\n")[-2].split("\t")
    out.step2 = os.path.join(out_dir, "BaseCellCalling", sample_id + ".calling.step2.tsv")
    with open(out.step2, "wb") as f:
        f.write(hdr)
    n2 = eng.format_table(eng.TABLE_STEP2)
    kinds, n_surv = eng.step2_summary(len(cols))
    survivors = eng.table_bytes(eng.TABLE_STEP3_ROWS, n_surv, prefix=hdr)
    eng.free_table(eng.TABLE_STEP3_ROWS)
    out.start_background(lambda: eng.append_table(eng.TABLE_STEP2, out.step2))
    t["step2"] = time.time() - t0
    try:
        if step3:
            t0 = time.time()
            final, unfiltered = calling.step3_bytes(survivors, params.delta_vaf, params.delta_mcf, params.min_ac_reads, params.min_ac_cells, params.clust_dist,
                                                    all_kinds=kinds, full_text=lambda: eng.table_bytes(eng.TABLE_STEP2, n2, prefix=hdr), survivors_only=True)
            out.step3 = os.path.join(out_dir, "BaseCellCalling", sample_id + ".calling.step3.tsv")
            out.step3_unfiltered = os.path.join(out_dir, "BaseCellCalling", sample_id + ".calling.step3.unfiltered.tsv")
            tsvio.write_bytes(out.step3, final)
            tsvio.write_bytes(out.step3_unfiltered, unfiltered)
            t["step3"] = time.time() - t0
        t["tables_wait"] = out.wait_for_tables()
    finally:
        try:
            out.wait_for_tables()
        finally:
            eng.free_table(eng.TABLE_STEP2)
    out.timings = t
    return out


def _chain_steps23(out, s1, t, eng, contig_names, out_dir, sample_id, params, editing, pon_sr, pon_lr, gnomad_af_json, step3):
    d = {"BaseCellCalling": os.path.join(out_dir, "BaseCellCalling")}
    t0 = time.time()
    keys = [calling.read_posset_keys(p, contig_names, params.reference_gz_compat) for p in (editing, pon_sr, pon_lr)]
    af = calling.open_gnomad(gnomad_af_json)              # a JSON table or a gnomad_db directory / sqlite file
    s2 = calling.step2_bytes(s1, eng, contig_names, keys[0], keys[1], keys[2], params.min_distance, af, params.max_gnomad_vaf)
    out.step2 = os.path.join(d["BaseCellCalling"], sample_id + ".calling.step2.tsv")
    tsvio.write_bytes(out.step2, s2)
    t["step2"] = time.time() - t0
    if not step3:
        t["tables_wait"] = out.wait_for_tables()
        out.timings = t
        return out
    t0 = time.time()
    final, unfiltered = calling.step3_bytes(s2, params.delta_vaf, params.delta_mcf, params.min_ac_reads, params.min_ac_cells, params.clust_dist)
    out.step3 = os.path.join(d["BaseCellCalling"], sample_id + ".calling.step3.tsv")
    out.step3_unfiltered = os.path.join(d["BaseCellCalling"], sample_id + ".calling.step3.unfiltered.tsv")
    tsvio.write_bytes(out.step3, final)
    tsvio.write_bytes(out.step3_unfiltered, unfiltered)
    t["step3"] = time.time() - t0
    t["tables_wait"] = out.wait_for_tables()              # (what of the background writers' time steps 2 and 3 did not cover)
    out.timings = t
    return out


def run_snv(bam: str, barcodes_tsv: str, ref_fasta: str, out_dir: str, sample_id: str, params: Optional[SnvParams] = None,
            editing: Optional[str] = None, pon_sr: Optional[str] = None, pon_lr: Optional[str] = None,
            gnomad_af_json: Optional[str] = None, device: int = 0, engine: Optional[Engine] = None,
            comm: Optional["regions.Comm"] = None, window_bytes: Optional[int] = None) -> SnvOutputs:
    """One sample through the whole chain.  comm (world > 1): one rank per GPU, genomic regions sharded over the ranks, call
    `regions.Comm.from_env()` before anything touches the GPU.  window_bytes: stream the BAM in batches of about that many
    uncompressed bytes and count window by window (a BAM whose reads do not fit in HBM; decode overlaps the GPU work)."""
    params = params or SnvParams()
    comm = comm or regions.Comm()
    own = engine is None
    eng = engine or Engine(comm.local_device_index if comm.world > 1 else device)
    try:
        if comm.world > 1 and window_bytes:
            raise ValueError("window_bytes (--window_gb) streams one process's BAM window by window; with %d ranks every rank holds its region's reads at once: "
                             "use one or the other" % comm.world)
        if comm.world > 1 or window_bytes:
            return _run_snv_regions(bam, barcodes_tsv, ref_fasta, out_dir, sample_id, params, editing, pon_sr, pon_lr, gnomad_af_json, eng, comm, window_bytes, keep_store=not own)
        # (the chain counts its sample once: an engine of our own keeps no store for counts nobody will ask for)
        res = load_sample(bam, barcodes_tsv, ref_fasta, eng, params.min_mapping_quality, count_params=params.count(), keep_store=not own)
        return run_chain(res, res.table.celltype_of, res.table.celltype_names, res.dec.report, out_dir, sample_id, params, editing, pon_sr, pon_lr,
                         gnomad_af_json)
    finally:
        if own:
            eng.close()


def _key(p) -> int:
    return (int(p[0]) << 32) | int(p[1])


def _windows(batches, n_contigs: int):
    """(lo, hi, records) per window from a stream of decoded batches in file (= coordinate) order: a window's region ends at the
    start tile of the next batch's first read — every read that can cover a column below that point has been seen — and the reads
    that reach past it are carried into the next window."""
    carry = None
    lo = (0, 0)
    prev = next(batches, None)
    while prev is not None:
        nxt = next(batches, None)
        while nxt is not None and nxt.records.n_reads == 0:          # a batch whose records were all dropped at decode: merge its counters forward
            for k, v in nxt.report.items():
                prev.report[k] = prev.report.get(k, 0) + v
            nxt = next(batches, None)
        rec = prev.records if carry is None else hostio.concat_records([carry, prev.records])
        hi = (n_contigs, 0)
        if nxt is not None:
            hi = (int(nxt.records.read_tid[0]), (int(nxt.records.read_pos[0]) // regions.TILE) * regions.TILE)
        if _key(hi) < _key(lo):
            raise ValueError("the BAM is not coordinate sorted: a batch starts at %s, before %s" % (hi, lo))
        yield lo, hi, rec, prev
        carry = None
        if nxt is not None and rec.n_reads:
            ends = regions.read_ends(rec)
            mask = ((rec.read_tid.astype(np.int64) << 32) | ends) > _key(hi)
            if mask.any():
                carry = rec.subset(mask)
        lo, prev = hi, nxt


def _prefetch(gen, depth: int = 1):
    """run a generator in a background thread, `depth` items ahead (the decode of the next batch overlaps the GPU and the writers)"""
    import queue
    import threading
    q: "queue.Queue" = queue.Queue(maxsize=depth)
    end = object()

    def work():
        try:
            for item in gen:
                q.put(item)
            q.put(end)
        except BaseException as e:            # noqa: BLE001 - handed to the consumer
            q.put(e)
    threading.Thread(target=work, daemon=True).start()
    while True:
        item = q.get()
        if item is end:
            return
        if isinstance(item, BaseException):
            raise item
        yield item


def _run_snv_regions(bam, barcodes_tsv, ref_fasta, out_dir, sample_id, params, editing, pon_sr, pon_lr, gnomad_af_json, eng, comm, window_bytes,
                     step3: bool = True, resident: Optional[dict] = None, table=None, keep_store: bool = True) -> SnvOutputs:
    """keep_store=False: a rank's slice is counted once and nothing else is asked of its reads (run_snv): the load keeps no tile store.
    resident / table / step3 serve the sharded two-pass loop (run_reannotation with several ranks): `resident` = SnvOutputs.resident of
    an earlier call on the same engine (the rank's reads stay in HBM, nothing is ingested again), `table` = (celltype_of per barcode
    id, cell-type names, SplitBam report) of the pass, step3=False stops after the step-2 table (pass 1 of the reference has no step 3)."""
    t: Dict[str, float] = {"decode": 0.0, "load": 0.0, "gpu_count_call": 0.0, "fetch": 0.0, "write_tables": 0.0}
    t_all = time.time()
    bc = hostio.read_barcodes(barcodes_tsv)
    names_fa, seqs = tsvio.read_fasta(ref_fasta)
    seq_of = dict(zip(names_fa, seqs))
    cts = list(table[1]) if table is not None else bc.celltype_names
    ct_of = np.asarray(table[0], np.uint8) if table is not None else bc.celltype_of
    d = {k: os.path.join(out_dir, k) for k in ("SplitBam", "BaseCellCounter/" + sample_id, "MergeCounts", "BaseCellCalling")}
    tmp = os.path.join(out_dir, "_pieces." + sample_id)
    if comm.rank == 0:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
        for p in list(d.values()) + [tmp]:
            os.makedirs(p, exist_ok=True)
    comm.barrier()
    report: Dict[str, int] = {}
    contig = {}

    def setup(dec):
        for n, l in zip(dec.contig_names, dec.contig_len):
            if n not in seq_of or len(seq_of[n]) != int(l):
                raise ValueError("contig %s of the BAM header is missing from %s or has another length" % (n, ref_fasta))
        eng.set_contigs(dec.contig_len)
        for tid, n in enumerate(dec.contig_names):
            eng.load_reference(tid, seq_of[n])
        eng.set_barcodes(ct_of, len(cts))
        contig["names"] = dec.contig_names

    if resident is not None:
        # the reads of this rank's region are in HBM already: the pass differs by its barcode -> cell-type table only
        if comm.world <= 1:
            raise ValueError("a resident pass belongs to a run with several ranks")
        lo, hi, dec = resident["lo"], resident["hi"], resident["dec"]
        contig["names"] = dec.contig_names
        eng.set_barcodes(ct_of, len(cts))
        work = iter([(lo, hi, None, dec)])
        report = dict(table[2]) if table is not None and table[2] is not None else dict(dec.report)
    elif comm.world > 1:
        t0 = time.time()
        dec = None
        if os.environ.get("LONGSOM_INGEST", "auto") in ("device", "auto"):
            # every rank ingests the file on its OWN GPU (inflate, decode, store: 0.5 s per GB, nothing on the host's threads, which the
            # ranks used to share); the regions are cut from the resident per-read / per-segment arrays, the same on every rank
            names_b, lens_b, first_rec = hostio.bam_header(bam)
            hdr = hostio.DecodedBam(None, names_b, np.asarray(lens_b, np.int64), {})
            setup(hdr)
            keys = ("total_reads", "pass_reads", "cb_not_found", "cb_not_matched", "mapq_filtered")

            def report_of(counts):
                rep = {"Total_reads": int(counts[0]), "Pass_reads": int(counts[1]), "CB_not_found": int(counts[2]), "CB_not_matched": int(counts[3])}
                if counts[4]:
                    rep["MAPQ"] = int(counts[4])
                return rep
            bai = hostio.find_bai(bam) if os.environ.get("LONGSOM_SHARD_INGEST", "1") != "0" else None
            if bai is not None:
                # every rank ingests the SLICE of the file its region needs, found through the .bai's linear index (regions.BaiPlan), as the
                # reference's workers fetch their window through the index; SplitBam's counters are summed over the ranks (each record is
                # counted by the rank whose region holds its start).  Any rank that cannot (records not aligned to the blocks of its
                # slice) takes every rank to the whole-file path below: the ranks agree first.
                plan = regions.BaiPlan(hostio.read_bai(bai), len(names_b), comm.world, os.path.getsize(bam))
                bounds = plan.bounds
                lo, hi = bounds[comm.rank], bounds[comm.rank + 1]
                got, ok, fatal = None, 1, None
                try:
                    # the rank's region and the count's parameters are known before its slice is loaded: the load counts in the same pass
                    # (for run_snv it is the slice's only count: under the count's own read filters, no tile store kept - what load_sample does for one GPU)
                    eng.set_region(lo[0], lo[1], hi[0], hi[1])
                    saved = eng.load_settings()
                    cp_slice = params.count()
                    eng.set_count_at_load(cp_slice)
                    if not keep_store:
                        eng.set_load_filter(cp_slice.min_mq, cp_slice.flag_exclude, cp_slice.ignore_orphans)
                        eng.set_store_policy(eng.STORE_SKIP_WHEN_COUNTED)
                    try:
                        got = regions.ingest_slice(eng, bam, plan, lo, hi, bc.barcodes, params.min_mapping_quality)
                    finally:
                        eng.restore_load_settings(saved)
                except _lib.LsgError as e:
                    if "straddle" in str(e):
                        ok = 0
                    else:
                        fatal = e
                except BaseException as e:                     # noqa: BLE001 - an index that is not this BAM's, a corrupt block, no memory ...
                    fatal = e
                comm.agree(fatal, "the ingest of a rank's slice of %s" % bam)      # (every rank raises when one did: nobody waits in the all-reduce below)
                n_cb = len(bc.barcodes)
                if int(comm.allreduce_sum(np.asarray([ok], np.int64))[0]) == comm.world:
                    counts = np.zeros(5 + 2 * n_cb + 2 * comm.world, np.int64)      # SplitBam's counters, the tallies, and per rank: records and bytes of its slice
                    if got is not None:
                        info, cb_pass, cb_low = got
                        counts[:5] = [info[k] for k in keys]; counts[5:5 + n_cb] = cb_pass; counts[5 + n_cb:5 + 2 * n_cb] = cb_low
                        counts[5 + 2 * n_cb + comm.rank] = info["n_records"]; counts[5 + 2 * n_cb + comm.world + comm.rank] = info["slice_bytes"]
                    else:
                        eng.load_reads(hostio.ReadRecords.empty())      # (no alignment in this rank's region)
                    counts = comm.allreduce_sum(counts)
                    t["ingest_records_by_rank"] = counts[5 + 2 * n_cb:5 + 2 * n_cb + comm.world].tolist()
                    t["ingest_slice_MB_by_rank"] = [round(x / 1e6, 3) for x in counts[5 + 2 * n_cb + comm.world:].tolist()]
                    dec = hostio.DecodedBam(None, names_b, np.asarray(lens_b, np.int64), report_of(counts), None, counts[5:5 + n_cb].copy(), counts[5 + n_cb:5 + 2 * n_cb].copy())
                    mine = None
            if dec is None:
                try:
                    info, cb_pass, cb_low = eng.load_bam(bam, bc.barcodes, min_mapq=params.min_mapping_quality, first_record_offset=first_rec)
                    dec = hostio.DecodedBam(None, names_b, np.asarray(lens_b, np.int64), report_of([info[k] for k in keys]), None, cb_pass, cb_low)
                    bounds = regions.balanced_boundaries(eng.reads_to_host(events=False), len(names_b), comm.world)
                    lo, hi = bounds[comm.rank], bounds[comm.rank + 1]
                    mine = None                                   # the rank's store holds the whole file; lsg_set_region makes the columns its own
                except _lib.LsgError as e:
                    if os.environ.get("LONGSOM_INGEST", "auto") == "device" or "straddle" not in str(e):
                        raise
                    dec = None
        if dec is None:
            dec = hostio.decode_bam(bam, bc.barcodes, min_mapq=params.min_mapping_quality, threads=max(1, (os.cpu_count() or 1) // comm.world))
            bounds = regions.balanced_boundaries(dec.records, len(dec.contig_names), comm.world)
            lo, hi = bounds[comm.rank], bounds[comm.rank + 1]
            mine = dec.records.subset(regions.reads_overlapping(dec.records, lo, hi))
        t["decode"] = time.time() - t0
        work = iter([(lo, hi, mine, dec)])
        report = dict(dec.report)
    elif (os.environ.get("LONGSOM_INGEST", "auto") in ("device", "auto") and os.environ.get("LONGSOM_SHARD_INGEST", "1") != "0"
          and hostio.find_bai(bam) is not None):
        # windows of an INDEXED BAM: the regions a sharded run would give its ranks, taken one after the other by this one GPU — every
        # window's slice of the file goes to the device as it is (regions.ingest_slice: no host decode, no carried reads: a slice
        # holds every read that reaches into its region)
        names_b, lens_b, first_rec = hostio.bam_header(bam)
        first = hostio.DecodedBam(None, names_b, np.asarray(lens_b, np.int64), {})
        n_win = max(1, -(-os.path.getsize(bam) * 3 // int(window_bytes)))          # (window_bytes counts uncompressed bytes: about a third as many in the file)
        plan = regions.BaiPlan(hostio.read_bai(hostio.find_bai(bam)), len(names_b), n_win, os.path.getsize(bam))

        def device_windows():
            setup(first)
            for r in range(n_win):
                lo, hi = plan.bounds[r], plan.bounds[r + 1]
                t0 = time.time()
                got = regions.ingest_slice(eng, bam, plan, lo, hi, bc.barcodes, params.min_mapping_quality)
                t["decode"] += time.time() - t0
                if got is None:
                    continue
                info = got[0]
                rep = {"Total_reads": int(info["total_reads"]), "Pass_reads": int(info["pass_reads"]), "CB_not_found": int(info["cb_not_found"]),
                       "CB_not_matched": int(info["cb_not_matched"])}
                if info["mapq_filtered"]:
                    rep["MAPQ"] = int(info["mapq_filtered"])
                yield lo, hi, None, hostio.DecodedBam(None, names_b, np.asarray(lens_b, np.int64), rep)
        work = device_windows()
    else:
        batches = hostio.stream_bam(bam, bc.barcodes, min_mapq=params.min_mapping_quality, batch_bytes=int(window_bytes))
        first = next(batches, None)
        if first is None:
            raise ValueError("%s holds no BAM records" % bam)

        def chained():
            yield first
            yield from batches
        work = _prefetch(_windows(chained(), len(first.contig_names)))
    kept: Dict[tuple, str] = {}
    # Step 2 is row-local when its distance filter is off (LongSom's setting: min_distance 0) — position-set probes, the gnomAD lookup, the
    # FILTER tags and the blanked NA fields all look at one row — so every rank (every window) runs it over its own rows on its own GPU,
    # writes its pieces of the step-2 table, and only the rows that survive step 3's FILTER patterns travel to rank 0, whose step 3 needs
    # the candidates of every region for its cluster filter (step3.py:283-306).  With a distance filter the rows' neighbours matter and
    # rank 0 runs step 2 over all kept rows, as before.
    local_step2 = int(params.min_distance) == 0 and os.environ.get("LONGSOM_LOCAL_STEP2", "1") != "0"
    s2_state: Dict[str, object] = {}

    def step2_piece(rows: bytes) -> bytes:
        """the step-2 rows of a piece of the step-1 table (rows only in, rows only out)"""
        if "keys" not in s2_state:
            s2_state["keys"] = [calling.read_posset_keys(p, contig["names"], params.reference_gz_compat) for p in (editing, pon_sr, pon_lr)]
            s2_state["af"] = calling.open_gnomad(gnomad_af_json)
            mh_ = tsvio.merged_header(cts, "##fileDate=x\n")
            s2_state["head"] = tsvio.step1_header([l + "\n" for l in mh_.split("\n") if l.startswith("##")], cts).encode()
            k = s2_state["keys"]
            s2_state["head_out"] = len(calling.step2_bytes(s2_state["head"], eng, contig["names"], k[0], k[1], k[2], 0, s2_state["af"], params.max_gnomad_vaf))
        k = s2_state["keys"]
        out2 = calling.step2_bytes(s2_state["head"] + rows, eng, contig["names"], k[0], k[1], k[2], 0, s2_state["af"], params.max_gnomad_vaf)
        return out2[s2_state["head_out"]:]
    n_windows = 0
    region_error = None
    try:
      for lo, hi, rec, dec in work:
          if not contig:
              setup(dec)
          names = contig["names"]
          if comm.world == 1:
              for k, v in dec.report.items():
                  report[k] = report.get(k, 0) + v
          n_windows += 1
          t0 = time.time()
          if rec is not None:                                # (None: the device ingest loaded the reads already)
              eng.load_reads(rec)
          eng.set_region(lo[0], lo[1], hi[0], hi[1])
          t["load"] += time.time() - t0
          t0 = time.time()
          eng.pileup_count(params.count())
          eng.call_step1(params.call())
          t["gpu_count_call"] += time.time() - t0
          t0 = time.time()
          per_ct = [eng.fetch_counts(ct) for ct in range(len(cts))]
          calls = eng.fetch_calls()
          t["fetch"] += time.time() - t0
          t0 = time.time()
          ckeys = calls["key"] if len(calls) else np.zeros(0, np.int64)
          for tid in np.unique(ckeys >> 32).tolist():
              k_lo, k_hi = tid << 32, (tid + 1) << 32
              sl = [slice(int(np.searchsorted(k, k_lo)), int(np.searchsorted(k, k_hi))) for k, _, _ in per_ct]
              sub = [(k[s_], r[s_], c[s_]) for (k, r, c), s_ in zip(per_ct, sl)]
              c0, c1 = int(np.searchsorted(ckeys, k_lo)), int(np.searchsorted(ckeys, k_hi))
              start1 = int(ckeys[c0] & 0xFFFFFFFF) + 1
              chrom = names[tid]
              for ct, name in enumerate(cts):
                  if len(sub[ct][0]):
                      tsvio.write_counts_tsv(regions.piece_path(tmp, chrom, start1, "counts." + name), *sub[ct], names, "", header=False)
              tsvio.write_merged_tsv(regions.piece_path(tmp, chrom, start1, "merged"), sub, names, cts, header=False)
              rows1 = tsvio.write_step1_tsv(regions.piece_path(tmp, chrom, start1, "step1"), calls[c0:c1], sub, names, cts, [], header=False, as_bytes=True)
              if local_step2:
                  t1 = time.time()
                  rows2 = step2_piece(rows1) if rows1 else b""
                  with open(regions.piece_path(tmp, chrom, start1, "step2"), "wb") as f:
                      f.write(rows2)
                  if rows2:      # what pandas' dtype inference over the WHOLE step-2 table will see in this piece's rows (calling.step3, all_kinds)
                      kk = tsvio.column_kinds(rows2, rows2[:rows2.index(b"\n")].count(b"\t") + 1)
                      s2_state["kinds"] = kk if "kinds" not in s2_state or len(s2_state["kinds"]) != len(kk) else (s2_state["kinds"] | kk)
                  surv = calling._step3_survivors(rows2, 6) if rows2 and os.environ.get("LONGSOM_STEP3_FULL_PARSE", "0") != "1" else None
                  kept[(chrom, start1)] = rows2 if surv is None else surv      # (Cell_types is column 6 of the step-1 / step-2 tables)
                  t["step2"] = t.get("step2", 0.0) + time.time() - t1
              else:
                  kept[(chrom, start1)] = rows1
          t["write_tables"] += time.time() - t0
    except BaseException as e:                              # noqa: BLE001 - raised again by the vote below, on every rank
        region_error = e
    # (no collective inside the loop above: a rank that failed there is heard of here, before anybody waits for its rows)
    comm.agree(region_error, "counting / writing a rank's region of %s" % bam)
    if not contig:                                     # a rank (or a file) without reads: still needs the contig names for the headers
        setup(dec if comm.world > 1 else first)
    names = contig["names"]
    # the candidate rows of every region on every rank (RCCL all-gather over xGMI when world > 1); then the pieces are complete
    payloads = comm.allgather_bytes(regions.pack_rows(kept))
    kinds_all = None
    if local_step2:                                    # the kinds of cell every rank saw in its rows of the step-2 table, OR-ed
        for blob in comm.allgather_bytes(s2_state["kinds"].tobytes() if "kinds" in s2_state else b""):
            if blob:
                kb = np.frombuffer(blob, np.uint8)
                kinds_all = kb.copy() if kinds_all is None or len(kinds_all) != len(kb) else (kinds_all | kb)
    comm.barrier()
    out = SnvOutputs(report=os.path.join(d["SplitBam"], sample_id + ".report.txt"), counts={}, merged="", step1="", step2="", step3="", step3_unfiltered="")
    for name in cts:
        out.counts[name] = os.path.join(d["BaseCellCounter/" + sample_id], "%s.%s.tsv" % (sample_id, name))
    out.merged = os.path.join(d["MergeCounts"], sample_id + ".BaseCellCounts.AllCellTypes.tsv")
    out.step1 = os.path.join(d["BaseCellCalling"], sample_id + ".calling.step1.tsv")
    out.step2 = os.path.join(d["BaseCellCalling"], sample_id + ".calling.step2.tsv")
    out.step3 = os.path.join(d["BaseCellCalling"], sample_id + ".calling.step3.tsv")
    out.step3_unfiltered = os.path.join(d["BaseCellCalling"], sample_id + ".calling.step3.unfiltered.tsv")
    t["windows"] = n_windows
    if comm.rank == 0:
        t0 = time.time()
        date = tsvio.file_date()
        write_report(out.report, report, t["decode"] if comm.world > 1 else time.time() - t_all)
        for name in cts:
            regions.concatenate_pieces(tmp, "counts." + name, tsvio.counts_header("%s.%s" % (sample_id, name), date), out.counts[name])
        mh = tsvio.merged_header(cts, date)
        regions.concatenate_pieces(tmp, "merged", mh, out.merged)
        s1h = tsvio.step1_header([l + "\n" for l in mh.split("\n") if l.startswith("##")], cts)
        regions.concatenate_pieces(tmp, "step1", s1h, out.step1)
        t["concatenate"] = time.time() - t0
        t0 = time.time()
        if local_step2:
            # the pieces of the step-2 table are on disk; what came over the wire is the rows step 3 still looks at
            keys = [calling.read_posset_keys(p, names, params.reference_gz_compat) for p in (editing, pon_sr, pon_lr)]
            s2h = calling.step2_bytes(s1h.encode(), eng, names, keys[0], keys[1], keys[2], 0, calling.open_gnomad(gnomad_af_json), params.max_gnomad_vaf)
            regions.concatenate_pieces(tmp, "step2", s2h.decode(), out.step2)
            s2 = s2h + regions.unpack_rows_bytes(payloads)
            t["step2"] = t.get("step2", 0.0) + time.time() - t0
        else:
            s1 = s1h.encode() + regions.unpack_rows_bytes(payloads)
            keys = [calling.read_posset_keys(p, names, params.reference_gz_compat) for p in (editing, pon_sr, pon_lr)]
            s2 = calling.step2_bytes(s1, eng, names, keys[0], keys[1], keys[2], params.min_distance, calling.open_gnomad(gnomad_af_json), params.max_gnomad_vaf)
            open(out.step2, "wb").write(s2)
            t["step2"] = time.time() - t0
        if step3:
            t0 = time.time()
            # (s2 holds the survivors of every region when step 2 ran where the rows are: the whole table's dtypes come with kinds_all, the
            # whole table itself - should a foreign cell make its printed form depend on them - from the file rank 0 has just assembled)
            final, unfiltered = calling.step3_bytes(s2, params.delta_vaf, params.delta_mcf, params.min_ac_reads, params.min_ac_cells, params.clust_dist,
                                                    all_kinds=kinds_all if local_step2 else None,
                                                    full_text=(lambda: open(out.step2, "rb").read()) if local_step2 else None)
            open(out.step3, "wb").write(final)
            open(out.step3_unfiltered, "wb").write(unfiltered)
            t["step3"] = time.time() - t0
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    comm.barrier()
    out.timings = t
    if not step3:
        out.step3 = out.step3_unfiltered = ""
    if comm.world > 1:
        out.resident = {"lo": lo, "hi": hi, "dec": dec, "table": bc}
    return out


@dataclass
class ReannoParams:
    """config/config.yaml:40-68 (Reanno block): pass-1 chain parameters + HCCV + re-annotation."""
    # Reanno.BaseCellCalling.  config.yaml:50-51 lists min_ac_cells 5 / min_ac_reads 20, but the reference's pass-1 step-1 rule never
    # forwards them (rules/CellTypeReannotation.smk:208-238) and pass 1 has no step 3, so the script defaults 2 / 3
    # (BaseCellCalling.step1.py:594-595) are what runs; the fused pass 1 does the same
    chain: SnvParams = field(default_factory=SnvParams)
    hccv_min_depth: float = 50
    hccv_delta_vaf: float = 0.2
    hccv_delta_mcf: float = 0.25
    hccv_clust_dist: int = 10000
    chrm_contaminant: str = "False"
    alt_flag: str = "All"
    pvalue: float = 0.01
    genotype_min_bq: int = 30
    min_variants: int = 3
    min_fraction: float = 0.25


@dataclass
class ReannoOutputs:
    pass1: SnvOutputs
    hccv: str
    genotype: str
    barcodes: str
    pass2: Optional[SnvOutputs]
    n_cells_kept: int = 0
    n_cancer: int = 0
    timings: Dict[str, float] = field(default_factory=dict)


def run_reannotation(bam: str, barcodes_tsv: str, ref_fasta: str, out_dir: str, sample_id: str, reanno_params: Optional[ReannoParams] = None,
                     snv_params: Optional[SnvParams] = None, fusions_tsv: Optional[str] = None, editing: Optional[str] = None,
                     pon_sr: Optional[str] = None, pon_lr: Optional[str] = None, gnomad_af_json: Optional[str] = None, device: int = 0,
                     engine: Optional[Engine] = None, pass1_step3: bool = False, comm: Optional["regions.Comm"] = None) -> ReannoOutputs:
    """The two-pass loop of the workflow (rules/CellTypeReannotation.smk + rules/SNVCalling.smk) in one process: the BAM is
    decoded and loaded ONCE; pass 1 calls with the automated annotation, the HCCV sites are genotyped per cell on the resident
    reads, the cells are re-annotated, and pass 2 re-counts the same resident reads under the new barcode table.
    Files: <out>/CellTypeReannotation/{SplitBam,BaseCellCounter,MergeCounts,BaseCellCalling,HCCV,ReannotatedCellTypes}/... and
    <out>/SNVCalling/{SplitBam,BaseCellCounter,MergeCounts,BaseCellCalling}/...
    comm (world > 1, BASELINE config 5 on several GPUs): one rank per GPU, every rank keeps the reads of its region resident across
    both passes (_run_reannotation_ranks)."""
    from . import reanno
    rp = reanno_params or ReannoParams()
    sp = snv_params or SnvParams()
    if comm is not None and comm.world > 1:
        own = engine is None
        eng = engine or Engine(comm.local_device_index)
        try:
            return _run_reannotation_ranks(bam, barcodes_tsv, ref_fasta, out_dir, sample_id, rp, sp, fusions_tsv, editing, pon_sr, pon_lr, gnomad_af_json, eng, comm,
                                           pass1_step3)
        finally:
            if own:
                eng.close()
    own = engine is None
    eng = engine or Engine(device)
    t = {}
    try:
        res = load_sample(bam, barcodes_tsv, ref_fasta, eng, rp.chain.min_mapping_quality, keep_unlisted=True)      # (the genotyping pileup's buffer holds every read)
        d1 = os.path.join(out_dir, "CellTypeReannotation")
        p1 = run_chain(res, res.table.celltype_of, res.table.celltype_names, res.dec.report, d1, sample_id, rp.chain, editing, pon_sr, pon_lr, gnomad_af_json,
                       step3=pass1_step3)
        t0 = time.time()
        os.makedirs(os.path.join(d1, "HCCV"), exist_ok=True)
        hccv = reanno.hccv_filter(p1.step2, os.path.join(d1, "HCCV", sample_id), rp.hccv_min_depth, rp.hccv_delta_vaf, rp.hccv_delta_mcf, rp.hccv_clust_dist)
        t["hccv"] = time.time() - t0
        t0 = time.time()
        eng.set_barcodes(res.table.celltype_of, len(res.table.celltype_names))
        geno = os.path.join(d1, "HCCV", sample_id + ".SNVs.SingleCellGenotype.tsv")
        gstats: Dict[str, dict] = {}
        n_rows = reanno.single_cell_genotype(eng, hccv, res.table, res.contig_names, geno, alt_flag=rp.alt_flag, min_bq=rp.genotype_min_bq,
                                             min_mq=rp.chain.min_mapping_quality, alpha2=rp.chain.alpha2, beta2=rp.chain.beta2, pvalue=rp.pvalue,
                                             chrm_contaminant=rp.chrm_contaminant, stats=gstats)
        t["genotype"] = time.time() - t0
        out = ReannoOutputs(p1, hccv, geno, "", None, timings=t)
        if n_rows == 0:                                   # no HCCV: the reference writes no genotype table and the workflow stops here
            return out
        t0 = time.time()
        os.makedirs(os.path.join(d1, "ReannotatedCellTypes"), exist_ok=True)
        out.barcodes = os.path.join(d1, "ReannotatedCellTypes", sample_id + ".tsv")
        out.n_cells_kept, out.n_cancer = reanno.celltype_reannotation(geno, fusions_tsv or "", barcodes_tsv, out.barcodes, rp.min_variants, rp.min_fraction,
                                                                      stats=gstats if os.environ.get("LONGSOM_REANNO_FROM_FILE", "0") != "1" else None)
        t["reannotation"] = time.time() - t0
        # pass 2: same resident reads, new barcode -> cell-type table (cells below coverage are no longer listed)
        if sp.min_mapping_quality != rp.chain.min_mapping_quality:
            raise ValueError("the two passes must share min_mapping_quality to share one decode (SplitBam report)")
        if out.n_cells_kept == 0:
            return out
        new = hostio.read_barcodes(out.barcodes)
        idx = {b: i for i, b in enumerate(res.table.barcodes)}
        ct2 = np.full(len(res.table.barcodes), 255, np.uint8)
        for b, c in zip(new.barcodes, new.celltype_of):
            ct2[idx[b]] = c
        out.pass2 = run_chain(res, ct2, new.celltype_names, res.dec.report_for(ct2 != 255), os.path.join(out_dir, "SNVCalling"), sample_id, sp, editing,
                              pon_sr, pon_lr, gnomad_af_json)
        return out
    finally:
        if own:
            eng.close()


def _run_reannotation_ranks(bam, barcodes_tsv, ref_fasta, out_dir, sample_id, rp, sp, fusions_tsv, editing, pon_sr, pon_lr, gnomad_af_json, eng, comm,
                            pass1_step3) -> ReannoOutputs:
    """The two-pass loop over several ranks: pass 1 is the sharded SNV run (every rank ingests and keeps its
    region's slice of the BAM, writes its pieces of the tables, rank 0 assembles them); rank 0 filters the HCCVs (a small host step
    over the step-2 table); every rank genotypes the HCCV sites of its own region on its resident reads and the per-(site, barcode)
    tables are summed over the ranks (one all-reduce: sites x barcodes x 2 integers); rank 0 re-annotates the cells; the new barcode ->
    cell-type table reaches the ranks as the file the rule graph writes anyway; pass 2 re-counts the SAME resident reads on every rank.
    No BAM byte is read twice, no read crosses a link: what travels is the candidate rows of both passes and the genotype tables."""
    from . import reanno
    if sp.min_mapping_quality != rp.chain.min_mapping_quality:
        raise ValueError("the two passes must share min_mapping_quality to share one decode (SplitBam report)")
    t: Dict[str, float] = {}
    d1 = os.path.join(out_dir, "CellTypeReannotation")
    eng.set_keep_unlisted(True)                             # (the genotyping pileup's buffer holds every read, listed or not: HCCVSingleCellGenotype.py:121-122)
    old_keep = hostio.set_keep_unlisted(True)
    try:
        p1 = _run_snv_regions(bam, barcodes_tsv, ref_fasta, d1, sample_id, rp.chain, editing, pon_sr, pon_lr, gnomad_af_json, eng, comm, None, step3=pass1_step3)
    finally:
        eng.set_keep_unlisted(False)
        hostio.set_keep_unlisted(old_keep)
    state = p1.resident
    table, dec = state["table"], state["dec"]
    hccv = os.path.join(d1, "HCCV", sample_id + ".HCCV.tsv")
    geno = os.path.join(d1, "HCCV", sample_id + ".SNVs.SingleCellGenotype.tsv")
    t0 = time.time()
    if comm.rank == 0:
        os.makedirs(os.path.join(d1, "HCCV"), exist_ok=True)
        made = reanno.hccv_filter(p1.step2, os.path.join(d1, "HCCV", sample_id), rp.hccv_min_depth, rp.hccv_delta_vaf, rp.hccv_delta_mcf, rp.hccv_clust_dist)
        assert os.path.abspath(made) == os.path.abspath(hccv), (made, hccv)
    comm.barrier()                                          # the HCCV file is there for every rank
    t["hccv"] = time.time() - t0
    t0 = time.time()
    eng.set_barcodes(table.celltype_of, len(table.celltype_names))
    gstats: Dict[str, dict] = {}
    n_rows = reanno.single_cell_genotype(eng, hccv, table, dec.contig_names, geno, alt_flag=rp.alt_flag, min_bq=rp.genotype_min_bq,
                                         min_mq=rp.chain.min_mapping_quality, alpha2=rp.chain.alpha2, beta2=rp.chain.beta2, pvalue=rp.pvalue,
                                         chrm_contaminant=rp.chrm_contaminant, comm=comm, region=(state["lo"], state["hi"]), stats=gstats)
    t["genotype"] = time.time() - t0
    out = ReannoOutputs(p1, hccv, geno, "", None, timings=t)
    if n_rows == 0:
        return out
    t0 = time.time()
    out.barcodes = os.path.join(d1, "ReannotatedCellTypes", sample_id + ".tsv")
    kept = np.zeros(2, np.int64)
    if comm.rank == 0:
        os.makedirs(os.path.join(d1, "ReannotatedCellTypes"), exist_ok=True)
        kept[:] = reanno.celltype_reannotation(geno, fusions_tsv or "", barcodes_tsv, out.barcodes, rp.min_variants, rp.min_fraction,
                                               stats=gstats if os.environ.get("LONGSOM_REANNO_FROM_FILE", "0") != "1" else None)
    kept = comm.allreduce_sum(kept)                          # (also the barrier behind which the new table is on disk)
    out.n_cells_kept, out.n_cancer = int(kept[0]), int(kept[1])
    t["reannotation"] = time.time() - t0
    if out.n_cells_kept == 0:
        return out
    new = hostio.read_barcodes(out.barcodes)
    idx = {b: i for i, b in enumerate(table.barcodes)}
    ct2 = np.full(len(table.barcodes), 255, np.uint8)
    for b, c in zip(new.barcodes, new.celltype_of):
        ct2[idx[b]] = c
    out.pass2 = _run_snv_regions(bam, barcodes_tsv, ref_fasta, os.path.join(out_dir, "SNVCalling"), sample_id, sp, editing, pon_sr, pon_lr, gnomad_af_json, eng, comm,
                                 None, resident=state, table=(ct2, new.celltype_names, dec.report_for(ct2 != 255)))
    return out


@dataclass
class PonOutputs:
    pon: str
    step1: Dict[str, str]
    n_sites: int = 0
    timings: Dict[str, Dict[str, float]] = field(default_factory=dict)


def pon_params(**kw) -> SnvParams:
    """config/config.yaml:33-37 (PoN block) over the BaseCellCalling step-1 defaults (rules/PoN.smk BaseCellCalling_step1_PoN)."""
    base = dict(min_ac_cells=1, min_ac_reads=1, min_cells=1, min_cell_types=1)
    base.update(kw)
    return SnvParams(**base)


def run_pon(normals, ref_fasta: str, out_dir: str, params: Optional[SnvParams] = None, min_samples: int = 1, rm_prefix: str = "No",
            write_tables: bool = True, out_name: str = "PoN_LR.tsv", device: int = 0, engine: Optional[Engine] = None,
            comm: Optional["regions.Comm"] = None) -> PonOutputs:
    """rules/PoN.smk from SplitBam_PoN to PoN in one process: for every normal (id, bam, barcodes.tsv) the count + step-1 call
    chain, then scripts/PoN/PoN.py over the sites with a filter status.  The beta-binomial parameters come in through `params`
    (BetaBinEstimation.py's VGAM fit stays outside).  write_tables=False skips the per-normal tables (the PoN needs only the call
    records); the PoN file is the same either way.  Output layout: <out_dir>/PoN/{SplitBam,BaseCellCounter,MergeCounts,
    BaseCellCalling,PoN}/ as in the rule file.  comm (world > 1): the normals are independent samples — rank r takes every world-th one
    on its own GPU, the sites of their call records are gathered (one all-gather of a few MB) and rank 0 writes the panel."""
    import json
    params = params or pon_params()
    comm = comm or regions.Comm()
    own = engine is None
    eng = engine or Engine(comm.local_device_index if comm.world > 1 else device)
    root = os.path.join(out_dir, "PoN")
    entries, step1, timings = [], {}, {}
    normals = list(normals)
    try:
        for sample_id, bam, barcodes_tsv in normals[comm.rank::comm.world]:
            res = load_sample(bam, barcodes_tsv, ref_fasta, eng, params.min_mapping_quality)
            out, _, calls, t = chain_step1(res, res.table.celltype_of, res.table.celltype_names, res.dec.report, root, sample_id, params, write_tables)
            label = sample_id + ".calling.step1.tsv"          # basename of the table, the sample id of PoN.py:55
            entries += pon.sites_of_calls(calls, res.contig_names, label)
            step1[sample_id] = out.step1
            timings[sample_id] = t
    finally:
        if own:
            eng.close()
    if comm.world > 1:
        # (plain JSON over the wire, the sites' byte fields as latin-1 text: nothing a peer sends is executed on arrival)
        mine = json.dumps({"entries": [[f.decode("latin-1") for f in e] for e in entries], "step1": step1, "timings": timings})
        entries, step1, timings = [], {}, {}
        for blob in comm.allgather_bytes(mine.encode("latin-1")):
            d = json.loads(blob.decode("latin-1"))
            entries += [tuple(f.encode("latin-1") for f in e) for e in d["entries"]]
            step1.update(d["step1"]); timings.update(d["timings"])
    path = os.path.join(root, "PoN", out_name)
    text = pon.pon_text(entries, min_samples, rm_prefix)          # (sorted inside: the panel does not depend on which rank took which normal)
    if comm.rank == 0:
        os.makedirs(os.path.join(root, "PoN"), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)
    comm.barrier()
    return PonOutputs(path, step1, sum(1 for l in text.split("\n") if l and not l.startswith("#")), timings)
