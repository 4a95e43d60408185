#!/usr/bin/env python3
# Fused replacement of rules SplitBam_PoN .. PoN of LongSom's workflow/rules/PoN.smk (longsom_amd.cli.pon_chain): every normal is
# decoded once, counted and called on the GPU, and the panel is written from the call records.
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from longsom_amd import cli  # noqa: E402

if __name__ == "__main__":
    cli.pon_chain()
