"""CPU: the rule files under workflow/rules/*.gpu.smk and the CLI shims agree.  snakemake is not installed here, so the
rules cannot be executed; what can be checked is the contract between a rule's `shell:` line and the parser of the script
it names (the reference's contract: workflow/rules/SNVCalling.smk:4-221 calls `python {script} --flag value ...`):
  * every script a rule names exists under workflow/scripts_gpu/ and calls one longsom_amd.cli entry point;
  * every --flag of the shell line is an option of that entry point's parser;
  * the line parses as the reference's DEFAULT config renders it (Run.PoN False: `--pon_LR` with an empty value), and with
    every placeholder filled."""
import argparse
import glob
import os
import re

import pytest

from longsom_amd import cli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RULES = sorted(glob.glob(os.path.join(ROOT, "workflow", "rules", "*.gpu.smk")))


class _Captured(Exception):
    def __init__(self, parser):
        self.parser = parser


def parser_of(fn):
    """the ArgumentParser an entry point builds (captured at its parse_args call)"""
    orig = argparse.ArgumentParser.parse_args

    def grab(self, *a, **k):
        raise _Captured(self)
    argparse.ArgumentParser.parse_args = grab
    try:
        fn([])
    except _Captured as c:
        return c.parser
    finally:
        argparse.ArgumentParser.parse_args = orig
    raise AssertionError("%s never parsed arguments" % fn.__name__)


def rule_blocks(text):
    """(rule name, script path relative to scripts_gpu, shell text) for every rule with a script= param"""
    out = []
    for m in re.finditer(r"^rule (\w+):\n(.*?)(?=^rule |\Z)", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        s = re.search(r'script=GPU_SCRIPTS\+"(/[^"]+)"', body)
        sh = re.search(r'shell:\s*\n(.*)', body, re.S)
        if s and sh:
            out.append((name, s.group(1).lstrip("/"), sh.group(1)))
    return out


def entry_point(script_rel):
    path = os.path.join(ROOT, "workflow", "scripts_gpu", script_rel)
    assert os.path.exists(path), "rule names a script that does not exist: " + script_rel
    m = re.search(r"cli\.(\w+)\(\)", open(path).read())
    assert m, script_rel + " does not call a longsom_amd.cli entry point"
    return getattr(cli, m.group(1))


CASES = [(os.path.basename(f), n, s, sh) for f in RULES for n, s, sh in rule_blocks(open(f).read())]


def test_rule_files_found():
    assert len(RULES) >= 3 and len(CASES) >= 8


@pytest.mark.parametrize("smk,rule,script,shell", CASES, ids=["%s:%s" % (c[0], c[1]) for c in CASES])
def test_shell_flags_exist_in_the_parser(smk, rule, script, shell):
    parser = parser_of(entry_point(script))
    known = set(parser._option_string_actions)
    flags = re.findall(r"(?<![\w{\[])(--[A-Za-z_][\w-]*)", shell)
    assert flags, "no flags found in the shell line of " + rule
    unknown = [f for f in flags if f not in known]
    assert not unknown, "%s: rule %s passes %s, which %s does not accept" % (smk, rule, unknown, script)


def render(shell, empty=()):
    """the shell text as snakemake would render it: every {placeholder} -> a value ('' for the names in `empty`);
    returns argv after the script path"""
    text = " ".join(re.findall(r'"((?:[^"\\]|\\.)*)"', shell)) if '"""' not in shell else shell.split('"""')[1]
    text = text.replace("\\\n", " ")

    def sub(m):
        key = m.group(1)
        if key in empty:
            return ""
        if key.startswith("params.") and ("compat" in key or "htslib" in key or "no_gnomad" in key or "launcher" in key or key == "params.tables"):
            return "python" if "launcher" in key else ""      # switches that render as a flag or as nothing
        if "alt_flag" in key:
            return "All"                                      # config.yaml:66 (a choice option)
        return "1"
    text = re.sub(r"\{([^{}]+)\}", sub, text)
    text = text.split(">")[0]
    argv = text.split()
    i = next(k for k, a in enumerate(argv) if a.startswith("--"))
    return argv[i:]


@pytest.mark.parametrize("smk,rule,script,shell", CASES, ids=["%s:%s" % (c[0], c[1]) for c in CASES])
def test_shell_line_parses_under_the_default_config(smk, rule, script, shell):
    parser = parser_of(entry_point(script))
    for empty in ((), ("input.pon_LR",)):                     # Run.PoN False (the reference's default) renders an empty --pon_LR
        argv = render(shell, empty)
        try:
            parser.parse_args(argv)
        except SystemExit:
            pytest.fail("%s: rule %s does not parse with %s empty: %s" % (smk, rule, empty or "nothing", " ".join(argv)))


def test_bare_optional_file_flags():
    for fn in (cli.snv, cli.reannotation):
        p = parser_of(fn)
        a = p.parse_args(["--bam", "b", "--meta", "m", "--ref", "r", "--id", "i", "--outdir", "o", "--pon_LR", "--editing", "e", "--pon_SR"])
        assert a.pon_LR == "" and a.pon_SR == "" and a.editing == "e" and a.gnomAD_db == ""


def test_unusable_gnomad_source_stops_unless_allowed(capsys, tmp_path, monkeypatch):
    from longsom_amd import calling
    monkeypatch.delenv("LONGSOM_ALLOW_MISSING_GNOMAD", raising=False)
    assert calling.open_gnomad(None) is None and capsys.readouterr().err == ""
    with pytest.raises(calling.GnomadUnusable, match="allow_missing_gnomad"):          # the reference stops in gnomAD_DB(); so does this
        calling.open_gnomad(str(tmp_path / "nowhere"))
    assert calling.open_gnomad(str(tmp_path / "nowhere"), allow_missing=True) is None
    assert "gnomAD filter of step 2 is OFF" in capsys.readouterr().err
    monkeypatch.setenv("LONGSOM_ALLOW_MISSING_GNOMAD", "1")
    assert calling.open_gnomad(str(tmp_path / "nowhere")) is None


def test_gnomad_sqlite_reader(tmp_path):
    """a database with the gnomad_db package's layout (table gnomad_db, chrom without 'chr') drives the same tags as the JSON form"""
    import sqlite3
    from longsom_amd import calling
    d = tmp_path / "gnomAD_v4.1"; d.mkdir()
    db = sqlite3.connect(str(d / "gnomad_db.sqlite3"))
    db.execute("CREATE TABLE gnomad_db (chrom TEXT, pos INTEGER, ref TEXT, alt TEXT, AF REAL, PRIMARY KEY (chrom, pos, ref, alt))")
    db.executemany("INSERT INTO gnomad_db VALUES (?,?,?,?,?)", [("1", 100, "A", "G", 0.5), ("M", 7, "C", "T", None)])
    db.commit(); db.close()
    src = calling.open_gnomad(str(d))
    assert src.get("chr1:100:A:G") == 0.5 and src.get("chr1:100:A:C") == 0.0 and src.get("chrM:7:C:T") == 0.0
