#!/usr/bin/env python3
# MI355X drop-in for LongSom's workflow/scripts/CellTypeReannotation/CellTypeReannotation.py: same flags, same output files
# (longsom_amd.cli.celltype_reannotation).
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from longsom_amd import cli  # noqa: E402

if __name__ == "__main__":
    cli.celltype_reannotation()
