"""The SNV chain as one fused run: BAM + barcodes.tsv + reference in, the rule outputs of
SplitBam -> BaseCellCounter -> MergeCounts -> BaseCellCalling_step1/2/3 out (R:SNVCalling.smk:4-221).

The reads are decoded once and stay in HBM; the per-cell-type split is a table lookup on the device,
the merge is fused into the call kernel; only text formatting and steps 2/3 (candidate rows only) run
on the host.  Every output file has the name and the bytes the reference's scripts give it (except
the wall-clock ##fileDate line).
"""
import json
import os
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import pandas as pd

from . import calling, hostio, tsvio
from ._lib import CallParams, CountParams
from .engine import Engine


@dataclass
class SnvParams:
    """config/config.yaml:73-90 (SNVCalling block) + the script defaults the rules do not override."""
    min_mapping_quality: int = 60
    min_bq: int = 20
    min_dp: int = 5
    min_cc: int = 5
    min_cell_types: int = 2
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

    def count(self) -> CountParams:
        return CountParams.longsom_defaults(min_bq=self.min_bq, min_mq=self.min_mapping_quality, min_dp=self.min_dp, min_cc=self.min_cc)

    def call(self) -> CallParams:
        return CallParams.longsom_defaults(alpha1=self.alpha1, beta1=self.beta1, alpha2=self.alpha2, beta2=self.beta2,
                                           min_ac_cells=self.min_ac_cells, min_ac_reads=self.min_ac_reads,
                                           min_cell_types=self.min_cell_types)


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


def write_report(path: str, report: Dict[str, int], seconds: float) -> None:
    """{id}.report.txt of SplitBamCellTypes (:181-187): one-row tab-separated table."""
    d = dict(report); d["Total_time"] = round(seconds, 2)
    pd.DataFrame([d]).to_csv(path, index=False, sep="\t")


def run_snv(bam: str, barcodes_tsv: str, ref_fasta: str, out_dir: str, sample_id: str, params: Optional[SnvParams] = None,
            editing: Optional[str] = None, pon_sr: Optional[str] = None, pon_lr: Optional[str] = None,
            gnomad_af_json: Optional[str] = None, device: int = 0, engine: Optional[Engine] = None) -> SnvOutputs:
    params = params or SnvParams()
    t = {}
    t0 = time.time()
    bc = hostio.read_barcodes(barcodes_tsv)
    dec = hostio.decode_bam(bam, bc.barcodes, min_mapq=params.min_mapping_quality)
    names_fa, seqs = tsvio.read_fasta(ref_fasta)
    seq_of = dict(zip(names_fa, seqs))
    # the pileup is driven by the FASTA's contigs (MakeWindows, BaseCellCounter.py:84-86); BAM tids index dec.contig_names
    contig_names = dec.contig_names
    for n, l in zip(contig_names, dec.contig_len):
        if n not in seq_of or len(seq_of[n]) != int(l):
            raise ValueError("contig %s of the BAM header is missing from %s or has another length" % (n, ref_fasta))
    t["decode"] = time.time() - t0
    own = engine is None
    eng = engine or Engine(device)
    try:
        t0 = time.time()
        eng.set_contigs(dec.contig_len)
        for tid, n in enumerate(contig_names):
            eng.load_reference(tid, seq_of[n])
        eng.set_barcodes(bc.celltype_of, len(bc.celltype_names))
        eng.set_region()
        eng.load_reads(dec.records)
        t["load"] = time.time() - t0
        t0 = time.time()
        eng.pileup_count(params.count())
        n_sites, n_cand = eng.call_step1(params.call())
        t["gpu_count_call"] = time.time() - t0
        t0 = time.time()
        per_ct = [eng.fetch_counts(ct) for ct in range(len(bc.celltype_names))]
        calls = eng.fetch_calls()
        t["fetch"] = time.time() - t0
        t0 = time.time()
        d = {k: os.path.join(out_dir, k) for k in ("SplitBam", "BaseCellCounter/" + sample_id, "MergeCounts", "BaseCellCalling")}
        for p in d.values():
            os.makedirs(p, exist_ok=True)
        out = SnvOutputs(report=os.path.join(d["SplitBam"], sample_id + ".report.txt"), counts={}, merged="", step1="", step2="", step3="",
                         step3_unfiltered="")
        write_report(out.report, dec.report, t["decode"])
        date = tsvio.file_date()
        for ct, name in enumerate(bc.celltype_names):
            p = os.path.join(d["BaseCellCounter/" + sample_id], "%s.%s.tsv" % (sample_id, name))
            with open(p, "w") as f:
                f.write(tsvio.format_counts_tsv(*per_ct[ct], contig_names, "%s.%s" % (sample_id, name), date))
            out.counts[name] = p
        merged_text = tsvio.format_merged_tsv(per_ct, contig_names, bc.celltype_names, date)
        out.merged = os.path.join(d["MergeCounts"], sample_id + ".BaseCellCounts.AllCellTypes.tsv")
        open(out.merged, "w").write(merged_text)
        header = [l + "\n" for l in merged_text.split("\n") if l.startswith("##")]
        s1 = tsvio.format_step1_tsv(calls, per_ct, contig_names, bc.celltype_names, header)
        out.step1 = os.path.join(d["BaseCellCalling"], sample_id + ".calling.step1.tsv")
        open(out.step1, "w").write(s1)
        keys = [calling.read_posset_keys(p, contig_names, params.reference_gz_compat) for p in (editing, pon_sr, pon_lr)]
        af = json.load(open(gnomad_af_json)) if gnomad_af_json else None
        s2 = calling.step2(s1, eng, contig_names, keys[0], keys[1], keys[2], params.min_distance, af, params.max_gnomad_vaf)
        out.step2 = os.path.join(d["BaseCellCalling"], sample_id + ".calling.step2.tsv")
        open(out.step2, "w").write(s2)
        final, unfiltered = calling.step3(s2, params.delta_vaf, params.delta_mcf, params.min_ac_reads, params.min_ac_cells, params.clust_dist)
        out.step3 = os.path.join(d["BaseCellCalling"], sample_id + ".calling.step3.tsv")
        out.step3_unfiltered = os.path.join(d["BaseCellCalling"], sample_id + ".calling.step3.unfiltered.tsv")
        open(out.step3, "w").write(final)
        open(out.step3_unfiltered, "w").write(unfiltered)
        t["format_write"] = time.time() - t0
        out.timings = t
        return out
    finally:
        if own:
            eng.close()
