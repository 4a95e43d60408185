"""Cell-type re-annotation pass of LongSom, host side:

  hccv_filter              <- HCCV_SNV                 scripts/CellTypeReannotation/HighConfidenceCancerVariants.py:8-88 (+ helpers :90-255)
  single_cell_genotype     <- run_interval + main      scripts/CellTypeReannotation/HCCVSingleCellGenotype.py:82-407
                                                        (the per-(site, barcode) counting itself is lsg_genotype_cells on the GPU,
                                                        the per-cell beta-binomial tail is lsg_betabinom_sf4)
  celltype_reannotation    <- main                     scripts/CellTypeReannotation/CellTypeReannotation.py:6-119

Files in, files out, byte for byte what the reference scripts leave behind (pandas does the text round trips there, so it
does them here).
"""
import math
import os
import re
from collections import Counter, OrderedDict
from typing import Dict, List, Optional, Sequence

import numpy as np
import pandas as pd

SYM_OF_BASE = {"A": 0, "C": 1, "T": 2, "G": 3, "I": 4, "D": 5, "N": 6}
GENOTYPE_HEADER = ["#CHROM", "Start", "End", "REF", "ALT_expected", "Cell_type_expected", "Num_cells_expected", "CB",
                   "Cell_type_observed", "Dp", "ALT", "VAF", "BetaBin", "MutationStatus"]
HCCV_INFO_LINE = "##INFO=HCCV_FILTER,Description=Filter status of the variant site for cell reannotation (high-confidence cancer variants)\n"


# ---------------------------------------------------------------------------------------------------------------------
# High-confidence cancer variants
# ---------------------------------------------------------------------------------------------------------------------
def _strip_multiallelic(flt: str) -> str:
    for pat in ("Multi-allelic,", ",Multi-allelic", "Multi-allelic"):        # :134-136, in this order
        flt = flt.replace(pat, "")
    return flt


def _dominant_alt(ref: str, info: str):
    """The cancer column's strongest non-reference A/C/T/G read count; None unless it is > 20x the runner-up (:108-117)."""
    counts = [int(v) for v in info.split("|")[3].split(":")[:4]]
    counts["ACTG".index(ref)] = 0
    top = counts.index(max(counts))                      # (np.argmax: the first of equal maxima)
    best = counts[top]
    counts[top] = 0
    second = max(counts)
    if not (second / best < 0.05):
        return None
    return top


def _resolve_multiallelic(row):
    """MultiAllelic_filtering (:90-161) on one row -> (ALT, FILTER, Cell_types, Bc, Cc, VAF, MCF, verdict)."""
    alt, flt, ctypes_s = row["ALT"], row["FILTER"], row["Cell_types"]
    same = (alt, flt, ctypes_s, row["Bc"], row["Cc"], row["VAF"], row["MCF"])
    if "Multi-allelic" not in flt and "|" not in alt:
        return same + ("KEEP",)
    ctypes = ctypes_s.split(",")
    cancer, normal = row["Cancer"], row["Non-Cancer"]
    if len(ctypes) > 1:
        i_c = 0 if ctypes[0] == "Cancer" else 1                                 # :95-100 (Cancer is one of the two)
        top = _dominant_alt(row["REF"], cancer)
        if top is None:
            return same + ("DELETE",)

        def one(info, i_ct):
            bc = int(info.split("|")[3].split(":")[top]); cc = int(info.split("|")[2].split(":")[top])
            return bc, cc, round(bc / int(row["Dp"].split(",")[i_ct]), 4), round(cc / int(row["Nc"].split(",")[i_ct]), 4)
        bc_c, cc_c, vaf_c, mcf_c = one(cancer, i_c)
        bc_n, cc_n, vaf_n, mcf_n = one(normal, 1 - i_c)
        base = "ACTG"[top]
        return (",".join([base, base]), _strip_multiallelic(flt), ctypes_s, ",".join([str(bc_n), str(bc_c)]), ",".join([str(cc_n), str(cc_c)]),
                ",".join([str(vaf_n), str(vaf_c)]), ",".join([str(mcf_n), str(mcf_c)]), "KEEP")
    if len(ctypes) == 1:
        if ctypes[0] != "Cancer":
            return same + ("DELETE",)
        top = _dominant_alt(row["REF"], cancer)
        if top is None:
            return same + ("DELETE",)
        bc = int(cancer.split("|")[3].split(":")[top]); cc = int(cancer.split("|")[2].split(":")[top])
        return ("ACTG"[top], _strip_multiallelic(flt), ctypes_s, bc, cc, round(bc / int(row["Dp"]), 4), round(cc / int(row["Nc"]), 4), "KEEP")
    return None                                                                  # the reference falls off the end here too


def _depth_verdict(a_info, b_info, min_dp) -> str:
    """DP_filtering (:202-212): both cell types need min_dp reads; a missing column (NaN) is 'NoCov'."""
    try:
        d1 = b_info.split("|")[0]
        d2 = a_info.split("|")[0]
    except AttributeError:
        return "NoCov"
    return "LowDepth" if int(d1) < min_dp or int(d2) < min_dp else "PASS"


def _delta_verdict(ctypes_s, vaf, mcf, d_vaf, d_mcf) -> str:
    """MCF_filtering (:215-257)."""
    ctypes = ctypes_s.split(",")
    if len(ctypes) == 1 and ctypes[0] == "Cancer":
        return "PASS" if float(vaf) >= d_vaf and float(mcf) >= d_mcf else "Low VAF/MCF"
    if len(ctypes) > 1:
        vafs, mcfs = vaf.split(","), mcf.split(",")
        i_c = 0 if ctypes[0] == "Cancer" else 1
        vaf_c, vaf_n = float(vafs[i_c]), float(vafs[1 - i_c])
        mcf_c, mcf_n = float(mcfs[i_c]), float(mcfs[1 - i_c])
        if vaf_c < 0.05:
            return "NonSig"
        if vaf_n > 0.1 and vaf_c - vaf_n < 2 * d_vaf:
            return "Heterozygous"
        if vaf_n > 0.2:
            return "Heterozygous"
        return "LowDeltaMCF" if mcf_c - mcf_n < d_mcf else "PASS"
    return "NonCancer"


def _cluster_tags(index: pd.Series, flt: pd.Series, clust_dist: int) -> pd.Series:
    """tag_clustered_SNVs + modify_filter (:163-199).  Neighbours are taken in the reference's order: sorted by
    (chromosome, position AS TEXT); chrM is exempt."""
    trip = sorted((tuple(i.split(":")) for i in index), key=lambda t: (t[0], t[1]))
    bad = set()
    for (c1, p1, b1), (c2, p2, b2) in zip(trip, trip[1:]):
        if c1 != c2 or c1 == "chrM":
            continue
        if abs(int(p1) - int(p2)) < clust_dist:
            bad.add(":".join([c1, p1, b1])); bad.add(":".join([c2, p2, b2]))
    tag = "Clust_dist{}".format(str(clust_dist))
    return pd.Series([(tag if f == "PASS" else f + "," + tag) if i in bad else f for i, f in zip(index, flt)], index=flt.index, dtype=object)


def _first_field(col: pd.Series, sep: str) -> pd.Series:
    """col.str.split(sep, n=1).str[0] without building the pieces (the cell-type columns are 60-character strings, a million of them):
    text up to the first sep, NaN where the value is not text"""
    def cut(x):
        if type(x) is not str:
            return np.nan
        i = x.find(sep)
        return x if i < 0 else x[:i]
    return pd.Series([cut(x) for x in col.tolist()], index=col.index, dtype=object)


def hccv_filter(step2_tsv: str, out_prefix: str, min_dp: float = 20, delta_vaf: float = 0.1, delta_mcf: float = 0.4, clust_dist: int = 10000) -> str:
    """Writes <out_prefix>.HCCV.tsv (and the reference's two intermediate tables, .HCCV.tsv2 / .HCCV.tsv3); returns its path.
    The reference applies three Python row functions to every row of the step-2 table (14 s for the 1.6 M rows of a 1 M-read sample,
    against 3 s for the whole SNV chain before it); here DP_filtering is two column operations and runs first,
    MultiAllelic_filtering runs on the rows it can change (FILTER naming
    Multi-allelic, or an ALT with '|'; every other row is returned as it came, :92-93), and
    MCF_filtering runs after the FILTER patterns, on the rows that are still there, as in the reference.  Same three files, byte
    for byte (tests/test_reanno_cpu.py: the goldens, and the row-wise form on shuffled, replicated and single-cell-type tables)."""
    if os.environ.get("LONGSOM_HCCV_ROW_PATH", "0") == "1":
        return _hccv_filter_rowwise(step2_tsv, out_prefix, min_dp, delta_vaf, delta_mcf, clust_dist)
    out = out_prefix + ".HCCV.tsv"
    cols = None
    with open(out, "w") as dst, open(step2_tsv) as src:
        for line in src:
            if not line.startswith("#"):
                break
            if "#CHROM" in line:
                cols = line.rstrip("\n").split("\t")
            else:
                dst.write(line)
        dst.write(HCCV_INFO_LINE)
    df = pd.read_csv(step2_tsv, sep="\t", comment="#", names=cols)
    df = df[df["Cell_types"] != "Non-Cancer"]
    # DP_filtering (:202-212) FIRST: the first field of both cell types' columns, a missing column is NoCov.  The reference runs it after
    # MultiAllelic_filtering, which neither reads nor writes those two columns and only drops rows: the rows that fail are gone either
    # way, and seven of eight fail (a row function that raises on such a row — the reference then stops — is not reached here).
    if len(df):
        def depth(col):
            return pd.to_numeric(_first_field(df[col], "|"), errors="coerce")
        d_c, d_n = depth("Cancer"), depth("Non-Cancer")
        df = df[(d_c >= min_dp) & (d_n >= min_dp)]
    df["INDEX"] = df["#CHROM"].astype(str) + ":" + df["Start"].astype(str) + ":" + _first_field(df["ALT"], ",")
    changed = ["ALT", "FILTER", "Cell_types", "Bc", "Cc", "VAF", "MCF"]
    if len(df) and not (df["FILTER"].map(type).eq(str).all() and df["ALT"].map(type).eq(str).all()):
        return _hccv_filter_rowwise(step2_tsv, out_prefix, min_dp, delta_vaf, delta_mcf, clust_dist)      # (a FILTER / ALT that is not text: the row functions decide)
    touch = df["FILTER"].str.contains("Multi-allelic", regex=False) | df["ALT"].str.contains("|", regex=False) if len(df) else pd.Series([], dtype=bool)
    if len(df) and touch.any():
        sub = df[touch]
        need = ["ALT", "FILTER", "Cell_types", "Bc", "Cc", "VAF", "MCF", "Cancer", "Non-Cancer", "REF", "Dp", "Nc"]
        lists = [sub[c].tolist() for c in need]
        res = [_resolve_multiallelic(dict(zip(need, vals))) for vals in zip(*lists)]      # (the row as a dict: a Series per row costs ten times the function)
        good = np.fromiter((r is not None and r[7] == "KEEP" for r in res), bool, len(res))      # (None: the reference's row of NaN, not KEEP)
        for c in changed:                                   # the reference's columns are object after its expand-assignment: the same values, as objects
            df[c] = df[c].astype(object)
        ok = sub.index[good]
        kept_res = [r for r, g in zip(res, good) if g]
        for j, c in enumerate(changed):
            df.loc[ok, c] = pd.Series([r[j] for r in kept_res], index=ok, dtype=object)
        df = df.drop(index=sub.index[~good])
    df = df[cols + ["INDEX"]]
    df["DP_FILTER"] = pd.Series("PASS", index=df.index, dtype=object)
    df.to_csv(out + "2", sep="\t", index=False, mode="a")
    # chrM keeps its own, shorter filter list (contaminants, :52-58)
    chrm = df[df["#CHROM"] == "chrM"].copy()
    df = df[df["#CHROM"] != "chrM"]
    chrm = chrm[~chrm["FILTER"].str.contains("Min|LR|gnomAD|LC|RNA", regex=True)]
    for pat in ("Noisy_site", "LC_Upstream|LC_Downstream", "gnomAD", "RNA_editing_db", "PoN"):      # filters 2-6 (:60-73)
        df = df[~df["FILTER"].str.contains(pat, regex=True)]
    df = pd.concat([df, chrm])
    if not len(df):
        df["HCCV_FILTER"] = df.apply(lambda r: _delta_verdict(r["Cell_types"], r["VAF"], r["MCF"], delta_vaf, delta_mcf), axis=1)      # (what the reference does with an empty frame)
    else:
        df["HCCV_FILTER"] = pd.Series([_delta_verdict(c, v, m, delta_vaf, delta_mcf) for c, v, m in zip(df["Cell_types"].tolist(), df["VAF"].tolist(), df["MCF"].tolist())],
                                      index=df.index, dtype=object)
    df.to_csv(out + "3", sep="\t", index=False, mode="a")
    df = df[df["HCCV_FILTER"] == "PASS"]
    df["FILTER"] = _cluster_tags(df["INDEX"], df["FILTER"], clust_dist)
    df = df[~df["FILTER"].str.contains("dist", regex=True)]
    df.to_csv(out, sep="\t", index=False, mode="a")
    return out


def _hccv_filter_rowwise(step2_tsv: str, out_prefix: str, min_dp: float = 20, delta_vaf: float = 0.1, delta_mcf: float = 0.4, clust_dist: int = 10000) -> str:
    """hccv_filter with every row function applied to every row, in the reference's order (HighConfidenceCancerVariants.py:8-88): what the
    reference-generated goldens pin directly, and what tests compare the column-wise form below with (LONGSOM_HCCV_ROW_PATH=1 selects it)."""
    out = out_prefix + ".HCCV.tsv"
    cols = None
    with open(out, "w") as dst, open(step2_tsv) as src:
        for line in src:
            if not line.startswith("#"):
                break
            if "#CHROM" in line:
                cols = line.rstrip("\n").split("\t")
            else:
                dst.write(line)
        dst.write(HCCV_INFO_LINE)
    df = pd.read_csv(step2_tsv, sep="\t", comment="#", names=cols)
    df["INDEX"] = df["#CHROM"].astype(str) + ":" + df["Start"].astype(str) + ":" + df["ALT"].str.split(",", n=1, expand=True)[0]
    df = df[df["Cell_types"] != "Non-Cancer"]
    fixed = df.apply(_resolve_multiallelic, axis=1, result_type="expand")
    df[["ALT", "FILTER", "Cell_types", "Bc", "Cc", "VAF", "MCF", "MultiAllelic_filter"]] = fixed
    df = df[df["MultiAllelic_filter"] == "KEEP"]
    df = df[cols + ["INDEX"]]
    df["DP_FILTER"] = df.apply(lambda r: _depth_verdict(r["Cancer"], r["Non-Cancer"], min_dp), axis=1)
    df = df[df["DP_FILTER"] == "PASS"]
    df.to_csv(out + "2", sep="\t", index=False, mode="a")
    # chrM keeps its own, shorter filter list (contaminants, :52-58)
    chrm = df[df["#CHROM"] == "chrM"].copy()
    df = df[df["#CHROM"] != "chrM"]
    chrm = chrm[~chrm["FILTER"].str.contains("Min|LR|gnomAD|LC|RNA", regex=True)]
    for pat in ("Noisy_site", "LC_Upstream|LC_Downstream", "gnomAD", "RNA_editing_db", "PoN"):      # filters 2-6 (:60-73)
        df = df[~df["FILTER"].str.contains(pat, regex=True)]
    df = pd.concat([df, chrm])
    df["HCCV_FILTER"] = df.apply(lambda r: _delta_verdict(r["Cell_types"], r["VAF"], r["MCF"], delta_vaf, delta_mcf), axis=1)
    df.to_csv(out + "3", sep="\t", index=False, mode="a")
    df = df[df["HCCV_FILTER"] == "PASS"]
    df["FILTER"] = _cluster_tags(df["INDEX"], df["FILTER"], clust_dist)
    df = df[~df["FILTER"].str.contains("dist", regex=True)]
    df.to_csv(out, sep="\t", index=False, mode="a")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Per-cell genotyping of the HCCV sites
# ---------------------------------------------------------------------------------------------------------------------
def read_target_windows(variant_file: str, window: int = 50000) -> "OrderedDict[str, List[List[str]]]":
    """build_dict_variants (HCCVSingleCellGenotype.py:243-265): lines grouped by CHROM_floor(POS / window), file order kept."""
    groups: "OrderedDict[str, List[List[str]]]" = OrderedDict()
    with open(variant_file) as f:
        for line in f:
            if line.startswith("#") or line.startswith("Chr"):
                continue
            el = line.rstrip("\n").split("\t")
            code = el[0] + "_" + str(math.floor(int(el[1]) / float(window)))
            groups.setdefault(code, []).append(el)
    return groups


def _p4_text(k: int) -> str:
    return str(k / 10000.0)


def single_cell_genotype(engine, variant_file: str, table, contig_names: Sequence[str], out_path: str, *, alt_flag: str = "All",
                         window: int = 50000, min_bq: int = 30, min_mq: int = 255, alpha2: float = 0.260288007167716,
                         beta2: float = 173.94711910763732, pvalue: float = 0.01, chrm_contaminant: str = "True",
                         strict_cb: bool = True, max_depth: int = 200000, comm=None, region=None, stats: Optional[dict] = None) -> int:
    """The reads, contigs and `table` (hostio.BarcodeTable) must be resident in `engine`.  Returns the number of rows written.
    comm (regions.Comm of several ranks) + region ((tid, pos) lo, hi of this rank): every rank genotypes the target sites of its
    own region on its own resident reads (its slice of the BAM holds every read that reaches into the region), the per-(site,
    barcode) depth / alt tables are summed over the ranks, and rank 0 writes the file (the other ranks' out_path is not touched).
    stats (a dict, filled in): per barcode string, the rows written with coverage ("covered") and with MutationStatus PASS ("mutated")
    — what CellTypeReannotation.py:9-18 counts when it reads the table back (celltype_reannotation takes them instead of the file).
    Row order = the reference's: windows by (chromosome text, smallest position), inside a window the positions in the
    iteration order of Python's set of them (:111,131 — reproduced by building that very set), per position every barcode
    of barcodes.tsv in file order."""
    from ._lib import GenotypeParams
    tid_of = {n: i for i, n in enumerate(contig_names)}
    groups = read_target_windows(variant_file, window)
    blocks = []                                   # (chrom, first pos0, [pos0 in set order], {pos0: (ref, alt, ctype, ncells)})
    for lines in groups.values():
        chrom = lines[0][0]
        sites: Dict[int, tuple] = {}
        for el in lines:
            sites[int(el[1]) - 1] = (el[3], el[4].split(",")[0], el[6], el[13])    # :92-105
        order = list(set(sites.keys()))                                               # CELLS is built by iterating this very set (:111,131)
        blocks.append((chrom, min(sites), order, sites))
    if not blocks:
        print("No temporary files found")                                             # :309; no output file
        return 0
    # one GPU call for all target sites
    key_of = {}
    for chrom, _, order, _ in blocks:
        if chrom not in tid_of:
            raise ValueError("contig %r of %s is not in the BAM header" % (chrom, variant_file))
        for p in order:
            key_of[(chrom, p)] = (tid_of[chrom] << 32) | p
    uniq = sorted(set(key_of.values()))
    row_of = {k: i for i, k in enumerate(uniq)}
    # a site's expected alt: the LAST line of its window that names it wins (dict update, :105); two windows never share a site
    alt_sym = np.full(len(uniq), 255, np.uint8)
    for chrom, _, order, sites in blocks:
        for p in order:
            alt_sym[row_of[key_of[(chrom, p)]]] = SYM_OF_BASE.get(sites[p][1], 255)
    params = GenotypeParams.longsom_defaults(min_bq=int(min_bq), min_mq=int(min_mq), alt_only=1 if alt_flag == "Alt" else 0,
                                             strict_cb=1 if strict_cb else 0)
    # the windows of target sites are the reference's pileup calls (bam.pileup(CHROM, START, END, ..., max_depth = 200000), :109-122):
    # consecutive in the sorted site list; the device replays the depth cap per window where a region can hold that many reads
    keys = np.asarray(uniq, np.int64)
    code = (keys >> 32) * (1 << 40) + ((keys & 0xFFFFFFFF) + 1) // int(window) if len(keys) else keys
    group_off = np.concatenate([[0], np.nonzero(np.diff(code))[0] + 1, [len(keys)]]).astype(np.int64)
    import time
    tm = {"t0": time.time()}
    if comm is not None and comm.world > 1:
        k_lo, k_hi = (int(region[0][0]) << 32) | int(region[0][1]), (int(region[1][0]) << 32) | int(region[1][1])
        mine = np.nonzero((keys >= k_lo) & (keys < k_hi))[0]
        n_cb = len(table.barcodes)
        both = np.zeros((2, len(keys), n_cb), np.int64)
        if len(mine):
            sub_code = code[mine]
            sub_off = np.concatenate([[0], np.nonzero(np.diff(sub_code))[0] + 1, [len(mine)]]).astype(np.int64)
            d_, a_ = engine.genotype_cells_grouped(keys[mine], alt_sym[mine], sub_off, params, max_depth)
            both[0, mine] = d_; both[1, mine] = a_
        both = comm.allreduce_sum(both.reshape(-1)).reshape(2, len(keys), n_cb)          # (a site belongs to one rank: the sum places the rows)
        dp, alt = both[0].astype(np.uint32), both[1].astype(np.uint32)
        if comm.rank != 0:
            out_path = os.devnull                      # (and no row is built for it, below)
    else:
        dp, alt = engine.genotype_cells_grouped(keys, alt_sym, group_off, params, max_depth)
    tm["device"] = time.time()
    # beta-binomial tails of the covered, mutated cells outside chrM
    need = []
    for chrom, _, order, _ in blocks:
        if chrm_contaminant == "True" and chrom == "chrM":
            continue
        for p in order:
            i = row_of[key_of[(chrom, p)]]
            cbs = np.nonzero(alt[i] > 0)[0]
            need.extend((i, int(cb)) for cb in cbs)
    p4 = {}
    if need:
        ii = np.array([a for a, _ in need]); cc = np.array([b for _, b in need])
        vals = engine.betabinom_sf4(alt[ii, cc], dp[ii, cc], alpha2, beta2)
        p4 = {(a, b): int(v) for (a, b), v in zip(need, vals)}
    tm["tails"] = time.time()
    names = [table.celltype_names[int(c)] for c in table.celltype_of]
    n_rows = 0
    n_cb = len(table.barcodes)
    # a site's table is mostly cells without a read there: those rows differ by the barcode only and are made once
    cell_cols = ["\t" + bc + "\t" + names[cb] + "\t" for cb, bc in enumerate(table.barcodes)]
    no_cov = [c + "0\t0\t.\t.\tNoCoverage\n" for c in cell_cols]
    n_cov, n_pass = np.zeros(n_cb, np.int64), np.zeros(n_cb, np.int64)
    with open(out_path, "w") as out:
        out.write("\t".join(GENOTYPE_HEADER) + "\n")
        by_chrom: Dict[str, Dict[int, tuple]] = {}
        for b in blocks:
            by_chrom.setdefault(b[0], {})[b[1]] = b                                  # a later window with the same (chrom, start) replaces (:288-294)
        for chrom in sorted(by_chrom):
            for start in sorted(by_chrom[chrom]):
                _, _, order, sites = by_chrom[chrom][start]
                for p in order:
                    if out_path == os.devnull:             # a rank that does not write the table: the count of its rows is all it needs
                        n_rows += n_cb
                        continue
                    i = row_of[key_of[(chrom, p)]]
                    ref_e, alt_e, ct_e, nc_e = sites[p]
                    head = "\t".join([str(chrom), str(p + 1), str(p + 1), ref_e, alt_e, str(ct_e), str(nc_e)])
                    d_row, a_row = dp[i], alt[i]
                    lines = [head + t for t in no_cov]
                    seen = np.nonzero(d_row)[0]
                    n_cov[seen] += 1
                    for cb in seen.tolist():
                        DP, ALT = int(d_row[cb]), int(a_row[cb])
                        bb = "."
                        if ALT > 0:
                            v = round(ALT / DP, 4)
                            vaf = str(v)
                            if chrm_contaminant == "True" and str(chrom) == "chrM":
                                status = "LowVAFChrM" if v < 0.3 else "PASS"
                            else:
                                k = p4[(i, cb)]
                                bb = _p4_text(k)
                                status = "PASS" if k / 10000.0 < pvalue else "BetaBin_problem"
                            if status == "PASS":
                                n_pass[cb] += 1
                        else:
                            vaf, status = str(float(0)), "NoAltReads"
                        lines[cb] = head + cell_cols[cb] + "\t".join([str(DP), str(ALT), vaf, bb, status]) + "\n"
                    out.write("".join(lines))
                    n_rows += n_cb
    if os.environ.get("LSG_TIMING"):
        import sys
        sys.stderr.write("[lsg] single_cell_genotype: %d sites x %d cells: device %.2f s, tails of %d cells %.2f s, table %.2f s\n"
                         % (len(keys), n_cb, tm["device"] - tm["t0"], len(need), tm["tails"] - tm["device"], time.time() - tm["tails"]))
    if stats is not None:
        cov_by, pass_by = {}, {}
        for cb, bc in enumerate(table.barcodes):              # (keyed by the string the rows carry, as a reader of the table would count)
            if n_cov[cb]:
                cov_by[bc] = cov_by.get(bc, 0) + int(n_cov[cb])
            if n_pass[cb]:
                pass_by[bc] = pass_by.get(bc, 0) + int(n_pass[cb])
        stats["covered"], stats["mutated"] = cov_by, pass_by
    return n_rows


# ---------------------------------------------------------------------------------------------------------------------
# Re-annotation
# ---------------------------------------------------------------------------------------------------------------------
def celltype_reannotation(snv_file: str, fusion_file: Optional[str], meta_file: str, out_file: str, min_variants: int = 3, min_frac: float = 0.2,
                          stats: Optional[dict] = None):
    """CellTypeReannotation.py:6-65: a cell is Cancer iff it has >= min_variants covered HCCVs and mutated / covered >= min_frac;
    cells with fewer covered HCCVs are dropped from the barcodes file.  stats = what single_cell_genotype counted while it wrote
    snv_file (rows with coverage and PASS rows per barcode): the table — sites x every barcode rows, 1.2 GB for 3 000 sites x 5 000
    cells — is then not read back (the fused loop; the script's drop-in reads the file, and a test holds the two together)."""
    if stats is not None:
        covered = Counter(stats["covered"])
        enough = [k for k, v in covered.items() if v >= min_variants]
        ok = set(enough)
        mutated = [k for k, v in stats["mutated"].items() if k in ok for _ in range(v)]
    else:
        snv = pd.read_csv(snv_file, sep="\t")
        covered = Counter(snv[snv["VAF"] != "."]["CB"])
        enough = [k for k, v in covered.items() if v >= min_variants]
        snv = snv[snv["CB"].isin(enough)]
        mutated = list(snv[snv["MutationStatus"] == "PASS"]["CB"])
    if fusion_file:
        fus = pd.read_csv(fusion_file, sep="\t")
        fus["INDEX"] = fus["#FusionName"] + ":" + fus["BC"]
        fus = fus.drop_duplicates(subset="INDEX", keep="last")
        mutated = mutated + list(fus["BC"])
    per_cell = Counter(mutated)
    frac = {k: (v / covered[k] if covered[k] >= min_variants else 0) for k, v in per_cell.items()}
    cancer = [k for k in per_cell if frac[k] >= min_frac]
    bcs = pd.read_csv(meta_file, sep="\t")
    bcs = bcs[bcs["Index"].isin(enough)]
    bcs["Before_Reannotation_cell_type"] = bcs["Cell_type"]
    cancer = set(cancer)
    bcs["Reannotated_cell_type"] = ["Cancer" if i in cancer else "Non-Cancer" for i in bcs["Index"]]
    bcs["Cell_type"] = bcs["Reannotated_cell_type"]
    bcs.to_csv(out_file, sep="\t", index=False)
    return len(bcs), int((bcs["Cell_type"] == "Cancer").sum())
