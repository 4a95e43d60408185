"""calling_oracle.py — TEST INFRASTRUCTURE ONLY (never imported by longsom_amd/).

CPU restatement, text in -> text out, of the file-level stages that follow the pileup:
  merge   MergeBaseCellCounts.merge_cell_types_files     /root/reference/workflow/scripts/SNVCalling/MergeBaseCellCounts.py:116-204
  step1   variant_calling_step1                           .../BaseCellCalling.step1.py:19-476 (+ :478-529)
  step2   variant_calling_step2 / GetExtraFilters         .../BaseCellCalling.step2.py:14-235
  step3   variant_calling_step3 + helpers                 .../BaseCellCalling.step3.py:8-316
The beta-binomial tests call scipy.stats.betabinom exactly as the reference does (the third-party
arithmetic of step1.py:196,201,329-330, unpinned version in workflow/envs/SComatic.yaml:25); the
gnomAD lookup (step2.py:100-108, database not in the tree) is an optional {"chrom:pos:ref:alt": AF}
mapping, default empty = AF 0.

Pinned: tests/test_oracle_cpu.py checks every function here against tests/golden/*, which are the
outputs of the reference's own code on the same inputs (tools/make_goldens.py).
"""
import collections
import math

from scipy.stats import betabinom

ALLELES = ["A", "C", "T", "G", "I", "D", "N", "O"]


# ---- merge --------------------------------------------------------------------------------------
def merge(texts, celltype_names):
    """texts: BaseCellCounter TSV contents (9 header lines each, MergeBaseCellCounts.py:131).
    Outer join on (chrom, pos) in python string order of chromosomes (:108-113) and ascending
    position (:89-106); 'NA' where a cell type has no row (:76-79)."""
    tables = []
    header8 = None
    for t in texts:
        lines = t.split("\n")
        if header8 is None:
            header8 = lines[:8]
        rows = collections.OrderedDict()
        for line in lines[9:]:
            line = line.strip()
            if line:
                chrom, pos, ref, info, bc = line.split("\t")
                rows[(chrom, int(pos))] = (ref, info, bc)
        tables.append(rows)
    sites = sorted(set().union(*[set(t) for t in tables]))
    out = header8 + ["\t".join(["#CHROM", "Start", "End", "REF", "INFO"] + list(celltype_names))]
    for chrom, pos in sites:
        refs, infos, cols = [], [], []
        for t in tables:
            r = t.get((chrom, pos))
            if r is None:
                cols.append("NA")
            else:
                refs.append(r[0]); infos.append(r[1]); cols.append(r[2])
        sort_set = lambda lst: "|".join(k for k, _ in sorted(collections.Counter(lst).items(), key=lambda kv: kv[1], reverse=True))
        out.append("\t".join([chrom, str(pos), str(pos), sort_set(refs), sort_set(infos)] + cols))
    return "\n".join(out) + "\n"


# ---- step 1 -------------------------------------------------------------------------------------
def _longest_run(s):
    best = run = 0
    prev = None
    for ch in s:
        run = run + 1 if ch == prev else 1
        best = max(best, run)
        prev = ch
    return best


def step1(merged_text, fasta, alpha1=0.21356677091082193, beta1=104.95163748636298, alpha2=0.2474528917555431,
          beta2=162.03696139428595, min_ac_cells=2, min_ac_reads=3, min_cells=5, min_reads=5, min_cell_types=2, max_cell_types=1,
          info_lines=None):
    """fasta: {chrom: upper-case sequence}.  Returns the step1 TSV text."""
    out = []
    cts = None
    for line in merged_text.split("\n"):
        if line.startswith("##"):
            out.append(line)
            continue
        if line.startswith("#CHROM"):
            el = line.split("\t")
            cts = el[5:]
            out.extend(info_lines)
            out.append("\t".join(el[:4] + ["ALT", "FILTER", "Cell_types", "Up_context", "Down_context", "N_ALT", "Dp", "Nc", "Bc", "Cc",
                                            "VAF", "MCF", "BCp", "CCp", "Cell_types_min_BC", "Cell_types_min_CC", "Rest_BC", "Rest_CC",
                                            "Fisher_p", "Cell_type_Filter"] + el[4:]))
            continue
        if not line:
            continue
        el = line.split("\t")
        chrom, pos, ref = el[0], int(el[1]), el[3]
        seq = fasta.get(chrom)
        if seq is None or pos - 6 < 0:                      # fetch raises -> '.' (step1.py:97-104)
            up, down = ".", "."
        else:
            ctx = seq[pos - 6:pos + 5]
            up, down = ctx[0:5], ctx[6:11]
        alts, ctypes, dps, ncs, bcs, ccs, bcp, ccp, vaf, mcf, flt = [], [], [], [], [], [], [], [], [], [], []
        n_min = 0
        s_alt_bc = s_alt_cc = s_dp = s_nc = 0
        for name, info in zip(cts, el[5:]):
            if info.startswith("NA"):
                continue
            DP, NC, CC, BC, _bq, _f, _r = info.split("|")
            DP, NC = int(DP), int(NC)
            if not (DP >= min_reads and NC >= min_cells):
                continue
            n_min += 1
            cc = [int(x) for x in CC.split(":")]
            bc = [int(x) for x in BC.split(":")]
            s_alt_bc += sum(bc[x] for x in range(len(bc)) if ALLELES[x] not in (ref, "O"))
            s_alt_cc += sum(cc[x] for x in range(len(cc)) if ALLELES[x] not in (ref, "O"))
            s_dp += DP; s_nc += NC
            a_bc = {ALLELES[x]: bc[x] for x in range(len(bc)) if ALLELES[x] not in (ref, "I", "D", "N", "O") and bc[x] > 0}
            a_cc = {ALLELES[x]: cc[x] for x in range(len(cc)) if ALLELES[x] not in (ref, "I", "D", "N", "O") and cc[x] > 0}
            p_bc = {x: round(betabinom.sf(a_bc[x] - 0.1, DP, alpha1, beta1), 4) for x in a_bc}
            p_cc = {x: round(betabinom.sf(a_cc[x] - 0.1, NC, alpha2, beta2), 4) for x in a_cc}
            cand = sorted(p_bc)
            if not cand:
                continue
            alts.append("|".join(cand)); ctypes.append(name); dps.append(str(DP)); ncs.append(str(NC))
            b = "|".join(str(a_bc[x]) for x in cand); c = "|".join(str(a_cc[x]) for x in cand)
            bcs.append(b); ccs.append(c)
            bcp.append("|".join(str(p_bc[x]) for x in cand)); ccp.append("|".join(str(p_cc[x]) for x in cand))
            vaf.append("|".join(str(round(a_bc[x] / float(DP), 4)) for x in cand))
            mcf.append("|".join(str(round(a_cc[x] / float(NC), 4)) for x in cand))
            b0 = sum(a_bc[x] for x in cand); c0 = sum(a_cc[x] for x in cand)
            s_dp -= b0; s_nc -= c0; s_alt_bc -= b0; s_alt_cc -= c0
            mb, mc = min(p_bc.values()), min(p_cc.values())
            if mb >= 0.05 or mc >= 0.05: flt.append("Non-Significant")
            elif 0.001 < mb < 0.05 or 0.001 < mc < 0.05: flt.append("Low-Significance")
            elif len(cand) > 1: flt.append("Multi-allelic")
            elif int(c) < min_ac_cells: flt.append("Low_cells")
            elif int(b) < min_ac_reads: flt.append("Low_reads")
            else: flt.append("PASS")
        if s_alt_bc > 0:
            pb = round(1 - betabinom.cdf(s_alt_bc - 0.1, s_dp, alpha1, beta1), 4)
            pc = round(1 - betabinom.cdf(s_alt_cc - 0.1, s_nc, alpha2, beta2), 4)
        else:
            pb = pc = 1
        rest_bc = ";".join([str(s_alt_bc), str(s_dp), str(pb)]); rest_cc = ";".join([str(s_alt_cc), str(s_nc), str(pc)])
        if alts:
            F = []
            n_pass = flt.count("PASS")
            if n_pass > max_cell_types: F.append("Multiple_cell_types")
            if len(set(alts)) > 1 or "Multi-allelic" in flt: F.append("Multi-allelic")
            if n_min < min_cell_types: F.append("Min_cell_types")
            if len(flt) - n_pass - flt.count("Non-Significant") > 0: F.append("Cell_type_noise")
            if pb < 0.05 or pc < 0.05: F.append("Noisy_site")
            if up != "." and max(_longest_run(up + x) for x in alts) >= 4: F.append("LC_Upstream")
            if down != "." and max(_longest_run(x + down) for x in alts) >= 4: F.append("LC_Downstream")
            FILTER = ",".join(F) if F else ("PASS" if "PASS" in flt else ",".join(flt))
            info = [",".join(alts), FILTER, ",".join(ctypes), up, down, str(len(set(alts))), ",".join(dps), ",".join(ncs), ",".join(bcs),
                    ",".join(ccs), ",".join(vaf), ",".join(mcf), ",".join(bcp), ",".join(ccp), str(n_min), str(n_min), rest_bc, rest_cc,
                    ".", ",".join(flt)]
        else:
            FILTER = "Noisy_site" if (pb < 0.001 or pc < 0.001) else "."
            info = [".", FILTER, ".", up, down, "."] + ["."] * 8 + [str(n_min), str(n_min), rest_bc, rest_cc, ".", "."]
        out.append("\t".join(el[:4] + info + el[4:]))
    return "\n".join(out) + "\n"


# ---- step 2 -------------------------------------------------------------------------------------
def read_posset(path):
    """build_dict (step2.py:197-221): {(chrom, pos)}; any failure -> empty set (bare except, :219-220)."""
    s = set()
    try:
        with open(path) as f:
            for line in f:
                if not line.startswith("#"):
                    el = line.split("\t")
                    s.add((el[0], int(el[1])))
    except Exception:
        return set()
    return s


def step2(step1_text, editing=frozenset(), pon_sr=frozenset(), pon_lr=frozenset(), distance=0, gnomad_af=None, gnomad_max=0.01):
    """Rows kept by the awk filter (step2.py:23), neighbour / position-set tags (:124-195), gnomAD tag (:223-235).
    The reference round-trips the table through pandas (:96,:117): 'NA' cells come back empty."""
    gnomad_af = gnomad_af or {}
    comments, header, rows = [], None, []
    for line in step1_text.split("\n"):
        if line.startswith("#"):
            if "#CHROM" in line: header = line
            else: comments.append(line)
        elif line:
            el = line.split("\t")
            if el[4] != "." and el[5] != ".":
                rows.append(el)
    out = []
    for i, el in enumerate(rows):
        el = list(el)
        # 3-window (:59-92): the first row sees rows 0..2, middle rows prev/next, the last row prev only;
        # with fewer than 3 rows every row sees all of them (:83-86)
        if len(rows) < 3: neigh = rows
        elif i == 0: neigh = rows[0:3]
        else: neigh = rows[i - 1:i + 2]
        chrom, pos = el[0], int(el[1])
        close = sum(1 for x in neigh if x[0] == chrom and int(x[1]) != pos and abs(int(x[1]) - pos) <= distance)
        F = el[5]
        for hit, tag in (((chrom, pos) in editing, "RNA_editing_db"), (close > 0, "Clustered"), ((chrom, pos) in pon_sr, "PoN_SR"),
                         ((chrom, pos) in pon_lr, "PoN_LR")):
            if hit: F = tag if F == "PASS" else F + "," + tag
        af = gnomad_af.get("%s:%s:%s:%s" % (chrom, el[1], el[3], el[4]), 0.0)
        if af != af: af = 0.0
        if af >= gnomad_max: F = "gnomAD" if F == "PASS" else F + ",gnomAD"
        el[5] = F
        out.append("\t".join("" if x == "NA" else x for x in el))
    return "\n".join(comments + [header] + out) + "\n"
