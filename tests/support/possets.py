"""Synthetic position sets of BASELINE.json's configuration C4 ("full PoN.scRNAseq.hg38 + AllEditingSites.hg38 hash filter resident in
HBM"): seeded random (contig, position) keys drawn over the workload's genome, optionally salted with some of the sample's own
candidate sites so that the filters have something to find.  The real files are not in either repository; their
sizes are: 5 M PoN positions, 15 M editing positions."""
import numpy as np

C4_SIZES = {"pon": 5_000_000, "editing": 15_000_000}


def random_keys(seed: int, n: int, contig_len, salt=None, salt_frac: float = 0.02) -> np.ndarray:
    """sorted unique int64 keys (tid << 32) | pos1 — the form lsg_load_posset takes (build_dict's sets, BaseCellCalling.step2.py:197-221)"""
    rng = np.random.default_rng(seed)
    lens = np.asarray(contig_len, np.int64)
    tid = rng.choice(len(lens), size=n, p=lens / lens.sum()).astype(np.int64)
    pos = (rng.random(n) * lens[tid]).astype(np.int64) + 1
    keys = (tid << 32) | pos
    if salt is not None and len(salt):
        take = rng.choice(len(salt), size=min(len(salt), max(1, int(len(salt) * salt_frac))), replace=False)
        keys = np.concatenate([keys, np.asarray(salt, np.int64)[take]])
    return np.unique(keys)
