"""Panel-of-normals aggregation (scripts/PoN/PoN.py) against the real awk / sed / sort of this machine: the reference's pipeline
is `grep -v '^#' | awk '$6 != "."' | [sed 's/^chr//'] | sort -k1,1 -k2,2 | datamash groupby 1,2 count 3 collapse 3 | awk '$3 >= n'`;
datamash is not installed, so its stage (count + comma-collapse of consecutive equal keys) is restated in this test."""
import os
import shutil
import subprocess

import pytest

from longsom_amd import cli, pon

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STEP1 = os.path.join(GOLD, "sample.calling.step1.tsv")


def make_normals(tmp_path):
    """three 'normals': the golden step-1 table, a thinned copy, and a copy with other contig spellings (prefix handling, string
    order of positions, a contig name that is a prefix of another)."""
    lines = open(STEP1).read().split("\n")
    head = [l for l in lines if l.startswith("#")]
    body = [l for l in lines if l and not l.startswith("#")]
    a = tmp_path / "A_Norm.calling.step1.tsv"
    b = tmp_path / "sub" / "B_Norm.calling.step1.tsv"
    c = tmp_path / "C_Norm.calling.step1.tsv"
    os.makedirs(b.parent)
    a.write_text("\n".join(head + body) + "\n")
    b.write_text("\n".join(head + body[::3]) + "\n")
    ren = []
    for i, l in enumerate(body[::2]):
        el = l.split("\t")
        el[0] = ("chr" + el[0]) if i % 3 == 0 else (el[0] + "_alt" if i % 3 == 1 else el[0])
        ren.append("\t".join(el))
    c.write_text("\n".join(head + ren) + "\n")
    lst = tmp_path / "files.txt"
    lst.write_text("%s\n%s\n%s\n" % (a, b, c))
    return lst, [a, b, c]


def shell_pon(files, min_samples, rm_prefix):
    if not (shutil.which("awk") and shutil.which("sort") and shutil.which("sed")):
        pytest.skip("awk / sort / sed not available")
    sed = "" if rm_prefix == "No" else "| sed 's/^chr//g'"
    cmd = "for file in %s; do BASE=$(basename $file); grep -v '^#' $file | awk -F'\\t' -v OFS='\\t' -v var=\"$BASE\" '{if ($6 != \".\") {print $1,$2,var}}' %s; done | sort -k1,1 -k2,2" % (
        " ".join(str(f) for f in files), sed)
    out = subprocess.run(["bash", "-c", cmd], check=True, capture_output=True, env=dict(os.environ, LC_ALL="C")).stdout.decode()
    rows, cur = [], None
    for l in out.split("\n"):
        if not l:
            continue
        c, p, s = l.split("\t")
        if cur and cur[0] == (c, p):
            cur[1].append(s)
        else:
            cur = [(c, p), [s]]
            rows.append(cur)
    return ["%s\t%s\t%d\t%s" % (k[0], k[1], len(v), ",".join(v)) for k, v in rows if len(v) >= min_samples]


@pytest.mark.parametrize("min_samples,rm_prefix", [(1, "No"), (2, "Yes"), (3, "No"), (2, "No")])
def test_pon_equals_the_shell_pipeline(tmp_path, min_samples, rm_prefix):
    lst, files = make_normals(tmp_path)
    out = tmp_path / "PoN.tsv"
    n = pon.build_from_files(str(lst), str(out), min_samples, rm_prefix)
    lines = out.read_text().split("\n")
    assert lines[0].startswith("##fileDate=")
    assert lines[1:4] == pon.HEADER.rstrip("\n").split("\n")
    body = [l for l in lines[4:] if l]
    want = shell_pon(files, min_samples, rm_prefix)
    assert len(want) > 50 or min_samples == 3
    assert body == want
    assert n == len(body)


def test_cli_flag_surface(tmp_path, capsys):
    lst, files = make_normals(tmp_path)
    out = tmp_path / "PoN.cli.tsv"
    cli.pon(["--in_tsv", str(lst), "--out_file", str(out)])                       # defaults: min_samples 2, rm_prefix Yes (PoN.py:14-15)
    assert [l for l in out.read_text().split("\n")[4:] if l] == shell_pon(files, 2, "Yes")


def test_step2_reads_the_panel_back(tmp_path):
    """the panel is consumed by BaseCellCalling.step2's build_dict (columns 1-2); every listed site must come back as a key"""
    from longsom_amd import calling, tsvio
    lst, files = make_normals(tmp_path)
    out = tmp_path / "PoN_LR.tsv"
    n = pon.build_from_files(str(lst), str(out), 1, "No")
    names = sorted({l.split("\t")[0] for l in out.read_text().split("\n") if l and not l.startswith("#")})
    keys = calling.read_posset_keys(str(out), names, False)
    assert len(keys) == n
