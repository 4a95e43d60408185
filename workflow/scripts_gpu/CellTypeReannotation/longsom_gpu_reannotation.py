#!/usr/bin/env python3
# Fused MI355X run of LongSom's re-annotation loop (rules/CellTypeReannotation.smk + rules/SNVCalling.smk): one decode, reads
# resident across both passes (longsom_amd.cli.reannotation).
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from longsom_amd import cli  # noqa: E402

if __name__ == "__main__":
    cli.reannotation()
