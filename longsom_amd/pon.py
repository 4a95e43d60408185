"""Panel of normals from the step-1 calls of normal samples.

The reference builds it with a shell pipeline (scripts/PoN/PoN.py:52-58): for every listed step-1 table keep the rows whose
FILTER column (6th) is not "." and print (CHROM, Start, basename of the file); optionally strip a leading "chr"; `sort -k1,1
-k2,2`; `datamash groupby 1,2 count 3 collapse 3`; keep groups with count >= min_samples; append to a four-line header.
Restated here on the host: the work per normal is the count + call chain (the GPU path), the aggregation is a sort of the few
sites with a filter status.  Sort order: the C locale's byte order (sort(1) compares field 1, then field 2 AS A STRING, then
the whole line), which is what `sort` does under LC_ALL=C / POSIX; other locales may order contig names with punctuation
differently, the set of rows is the same.
"""
import os
import time
from itertools import groupby
from typing import Iterable, List, Sequence, Tuple

import numpy as np

from .tsvio import SF_CANDIDATE

SF_NOISY_SITE = 16        # enum lsg_site_filter, include/longsom_hip.h

HEADER = ("##INFO=Num_samples,Description=Number of significant samples (beta-binomial test)\n"
          "##INFO=Sample_ids,Description=ID of the significant samples (beta-binomial test)\n"
          "#CHROM\tPOS\tNum_samples\tSample_ids\n")                  # PoN.py:37-45

Entry = Tuple[bytes, bytes, bytes]      # (chrom, position as printed, sample label)


def sites_of_step1_file(path: str) -> List[Entry]:
    """grep -v '^#' | awk '$6 != "."  {print $1, $2, basename}'  (PoN.py:55)"""
    label = os.path.basename(path).encode()
    out = []
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b"#"):
                continue
            el = line.rstrip(b"\n").split(b"\t", 6)
            if len(el) > 5 and el[5] != b".":
                out.append((el[0], el[1], label))
            elif len(el) <= 5:                       # awk: a missing 6th field is the empty string, which is != "."
                out.append((el[0] if el else b"", el[1] if len(el) > 1 else b"", label))
    return out


def sites_of_calls(calls: np.ndarray, contig_names: Sequence[str], label: str) -> List[Entry]:
    """The same rows taken from the step-1 call records (lsg_fetch_calls) instead of the table's text: FILTER is "." unless the
    site has an alt candidate or is a Noisy_site (BaseCellCalling.step1.py:333-372)."""
    sel = (calls["site_filter"] & np.uint32(SF_CANDIDATE | SF_NOISY_SITE)) != 0
    keys = calls["key"][sel]
    names = [n.encode() for n in contig_names]
    lab = label.encode()
    return [(names[int(k) >> 32], b"%d" % ((int(k) & 0xFFFFFFFF) + 1), lab) for k in keys]


def pon_text(entries: Iterable[Entry], min_samples: int = 2, rm_prefix: str = "Yes", date_line: str = None) -> str:
    rows = list(entries)
    if rm_prefix != "No":                            # sed 's/^chr//g'
        rows = [(c[3:] if c.startswith(b"chr") else c, p, s) for c, p, s in rows]
    rows.sort()                                      # sort -k1,1 -k2,2, ties by the whole line
    out = [date_line if date_line is not None else "##fileDate=%s\n" % time.strftime("%d/%m/%Y"), HEADER]
    for (c, p), g in groupby(rows, key=lambda r: (r[0], r[1])):
        ids = [r[2] for r in g]
        if len(ids) >= min_samples:                  # datamash groupby 1,2 count 3 collapse 3 | awk '$3 >= min_samples'
            out.append("%s\t%s\t%d\t%s\n" % (c.decode(), p.decode(), len(ids), b",".join(ids).decode()))
    return "".join(out)


def build_from_files(in_tsv: str, out_file: str, min_samples: int = 2, rm_prefix: str = "Yes") -> int:
    """PoN.py --in_tsv LIST --out_file OUT [--min_samples N] [--rm_prefix Yes|No]; returns the number of PoN sites."""
    entries: List[Entry] = []
    for path in open(in_tsv).read().split():         # for file in $(cat list)
        entries += sites_of_step1_file(path)
    text = pon_text(entries, min_samples, rm_prefix)
    with open(out_file, "w") as f:
        f.write(text)
    return sum(1 for l in text.split("\n") if l and not l.startswith("#"))
