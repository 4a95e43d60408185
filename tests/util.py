"""Helpers shared by the parity tests."""
import re

_NEG_ZERO = re.compile(r"(?<=;)-0\.0(?=[\t\n]|$)")


def neg_zero(text: str) -> str:
    """round(1 - betabinom.cdf(...), 4) prints `-0.0` when scipy's sum lands a hair above 1; the device's tail
    prints `0.0`.  Only a WHOLE p-value token — the third ';'-field of Rest_BC / Rest_CC — is rewritten: a value such as -0.05, or
    `-0.0` inside any other field, stays as it is and would fail the comparison."""
    return _NEG_ZERO.sub("0.0", text)


def assert_same_records(a, b, phased_a=False):
    """two sets of read-record arrays hold the same reads, segments and events, wherever their events lie (seg_ev_off is the producer's
    choice: include/longsom_hip.h).  phased_a: a's events are tile-phased (LSG_LAYOUT_PHASED) - every segment at an offset congruent to
    its reference start modulo 128, every read's region a multiple of 128, zeros between the segments."""
    import numpy as np
    for name in ("read_tid", "read_pos", "read_flag", "read_mapq", "read_cb", "seg_read", "seg_start", "seg_len"):
        np.testing.assert_array_equal(getattr(a, name), getattr(b, name), err_msg=name)
    ln = a.seg_len.astype(np.int64)
    if len(ln):
        idx = np.repeat(np.arange(len(ln)), ln)
        within = np.arange(int(ln.sum())) - np.repeat(np.cumsum(ln) - ln, ln)
        ia, ib = a.seg_ev_off[idx] + within, b.seg_ev_off[idx] + within
        np.testing.assert_array_equal(a.events[ia], b.events[ib], err_msg="events")
        if phased_a:
            assert ((a.seg_ev_off - a.seg_start) % 128 == 0).all() and len(a.events) % 128 == 0
            gaps = np.ones(len(a.events), bool); gaps[ia] = False
            assert not a.events[gaps].any()
            first = np.r_[True, a.seg_read[1:] != a.seg_read[:-1]]
            assert ((a.seg_ev_off[first] - (a.seg_start[first] % 128)) % 128 == 0).all()      # a read's region starts at a multiple of 128


def phased_records(rec):
    """the same read-record arrays with their events laid out tile-phased (LSG_LAYOUT_PHASED, include/longsom_hip.h): what the device BAM
    decoder and the synthetic generators produce, made here from any compact arrays"""
    import dataclasses
    import numpy as np
    S = len(rec.seg_len)
    if S == 0:
        return rec
    st, ln, rd = rec.seg_start.astype(np.int64), rec.seg_len.astype(np.int64), rec.seg_read
    off = np.zeros(S, np.int64)
    cur, region0 = 0, 0
    for s in range(S):
        if s == 0 or rd[s] != rd[s - 1]:
            cur = region0 = (cur + 127) // 128 * 128
        cur = cur + ((st[s] - (cur - region0)) % 128)
        off[s] = cur
        cur += ln[s]
    n = (cur + 127) // 128 * 128
    ev = np.zeros(n, np.uint16)
    idx = np.repeat(np.arange(S), ln)
    within = np.arange(int(ln.sum())) - np.repeat(np.cumsum(ln) - ln, ln)
    ev[off[idx] + within] = rec.events[rec.seg_ev_off[idx] + within]
    return dataclasses.replace(rec, seg_ev_off=off, events=ev)
