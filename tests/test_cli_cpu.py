"""CPU: MakeWindows' region arithmetic behind BaseCellCounter's --bed / --bed_out (BaseCellCounter.py:81-113), hand-derived: bedtools merge
with d = 1, the clip to [1, contig length), subtract.  (pybedtools is neither in the reference tree nor in this image: unpinned.)"""
import numpy as np

from longsom_amd import cli


def keys(tid, positions):
    return (np.int64(tid) << 32) | np.asarray(positions, np.int64)


def test_whole_contigs_without_a_bed_skip_position_zero(tmp_path):
    out = tmp_path / "out.bed"
    out.write_text("chr1\t5\t8\n")
    k = np.concatenate([keys(0, [0, 1, 4, 5, 7, 8, 99]), keys(1, [0, 3])])
    assert cli.bed_mask(k, ["chr1", "chr2"], [100, 50], "", "").tolist() == [False, True, True, True, True, True, True, False, True]
    assert cli.bed_mask(k, ["chr1", "chr2"], [100, 50], "", str(out)).tolist() == [False, True, True, False, False, True, True, False, True]


def test_bed_intervals_merge_at_one_base_clip_and_subtract(tmp_path):
    bed = tmp_path / "in.bed"
    # unsorted on purpose; [0,10) + [11,20): one base apart -> merged, position 10 is inside; [30,40) + [42,50): two apart -> not merged;
    # [95,200) is clipped to the contig; chrU is not in the genome
    bed.write_text("track name=x\nchr1\t11\t20\nchr1\t0\t10\nchr1\t42\t50\nchr1\t30\t40\nchr1\t95\t200\nchrU\t1\t9\nchr2\t3\t4\n")
    out = tmp_path / "out.bed"
    out.write_text("chr1\t15\t17\nchr1\t39\t43\n")
    pos = [0, 1, 9, 10, 11, 14, 15, 16, 17, 19, 20, 29, 30, 38, 39, 40, 41, 42, 43, 49, 50, 94, 95, 99]
    k = np.concatenate([keys(0, pos), keys(1, [2, 3, 4])])
    got = cli.bed_mask(k, ["chr1", "chr2"], [100, 50], str(bed), "")
    want1 = {1, 9, 10, 11, 14, 15, 16, 17, 19, 30, 38, 39, 42, 43, 49, 95, 99}
    assert [p for p, g in zip(pos, got[:len(pos)]) if g] == sorted(want1)
    assert got[len(pos):].tolist() == [False, True, False]
    got2 = cli.bed_mask(k, ["chr1", "chr2"], [100, 50], str(bed), str(out))
    assert [p for p, g in zip(pos, got2[:len(pos)]) if g] == sorted(want1 - {15, 16, 39, 42})
    assert cli.bed_mask(np.zeros(0, np.int64), ["chr1"], [100], str(bed), "").tolist() == []
