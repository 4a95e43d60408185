"""CPU, world_size 2 over gloo: the region sharding used for N > 1 and the variable-length all-gather.
Each rank evaluates its shard of the workload model on the host, counts its own region with the oracle, and the
gathered per-rank tables put together equal the unsharded result."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from longsom_amd import hostio, shard, synth
from oracle import loader


def model():
    return synth.named("C1", n_reads=1200, n_genes=30, n_cb=40)


def shard_rows(m, rank, world, refs):
    lo, hi, g_lo, g_hi = shard.region_shards(m, world)[rank]
    rec = hostio.synth_records(shard.sub_model(m, g_lo, g_hi))
    out = []
    for ct in (0, 1):
        k, r, c, _ = loader.count(rec, m.contig_len, refs, m.celltype_of, ct, 20, 60, 3, 2)
        keep = shard.in_region(k, lo, hi)
        out.append((k[keep], c[keep]))
    return out


def refs_of(m):
    return [hostio.ref_bases(m.seed, t, int(L)) for t, L in enumerate(m.contig_len)]


def worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = model()
    refs = refs_of(m)
    rows = shard_rows(m, rank, world, refs)
    for ct in (0, 1):
        k, c = rows[ct]
        table = np.concatenate([k.view(np.uint8).reshape(-1, 8), c.view(np.uint8).reshape(len(k), -1)], axis=1) if len(k) else np.zeros((0, 8 + 168), np.uint8)
        parts = shard.all_gather_rows(torch.from_numpy(np.ascontiguousarray(table)), dist)
        if rank == 0:
            allrows = np.concatenate([p.numpy() for p in parts])
            np.save(os.path.join(tmp, "gathered_%d.npy" % ct), allrows)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_regions_partition_and_gather(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    m = model()
    refs = refs_of(m)
    rec = hostio.synth_records(m)
    for ct in (0, 1):
        k, r, c, _ = loader.count(rec, m.contig_len, refs, m.celltype_of, ct, 20, 60, 3, 2)
        got = np.load(tmp_path / ("gathered_%d.npy" % ct))
        gk = got[:, :8].copy().view(np.int64).ravel()
        gc = got[:, 8:].copy().view(np.uint32).reshape(len(gk), 42)
        assert len(k) > 20
        np.testing.assert_array_equal(gk, k)        # rank order = genomic order, no duplicates, nothing lost
        np.testing.assert_array_equal(gc, c)


def test_region_shards_cover_genome_for_any_world():
    m = model()
    for world in (1, 2, 3, 4, 8):
        sh = shard.region_shards(m, world)
        assert sh[0][0] == (0, 0) and sh[-1][1] == (len(m.contig_len), 0)
        for a, b in zip(sh, sh[1:]):
            assert a[1] == b[0] and a[1][1] % 64 == 0
        total = sum(int(np.diff(m.gene_read_off)[g_lo:g_hi].sum()) for _, _, g_lo, g_hi in sh)
        assert total >= m.n_reads                   # boundary genes are loaded by both neighbours
