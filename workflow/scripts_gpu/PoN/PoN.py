#!/usr/bin/env python3
# MI355X-side drop-in for LongSom's workflow/scripts/PoN/PoN.py: same flags, same output file (longsom_amd.cli.pon); needs neither
# datamash nor a GPU (it aggregates step-1 tables that already exist).
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from longsom_amd import cli  # noqa: E402

if __name__ == "__main__":
    cli.pon()
