"""tests/schedule_check.py on synthetic traces (CPU) and on traces libsfgpu.so wrote on the MI355X box and that are
committed under tests/golden/schedule/ (made by tests/golden/make_schedule_golden.py through gpurun): the production
schedule must come out clean, and the round-2 race — re-introduced by `SF_TRACE_SCHEDULE=<file>,inject=trap` — must
be reported, deterministically."""
import glob
import json
import os

import pytest

import schedule_check as SC

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "schedule")


def op(name, slab, stream, *acc):
    return {"t": "op", "name": name, "slab": slab, "stream": stream, "acc": [list(a) for a in acc], "_line": 0}


def rec(slab, stream, ev):
    return {"t": "rec", "slab": slab, "stream": stream, "ev": ev}


def wait(slab, stream, ev, evslab=None):
    return {"t": "wait", "slab": slab, "stream": stream, "ev": ev, "evslab": slab if evslab is None else evslab}


def two_stream_sweep(with_wait):
    """Boundary launch on bs and interior launch on cs of one sweep, then the next sweep's interior launch, which
    overwrites the buffer the FIRST boundary launch read: ordered only if cs waits for boundary_done."""
    r = [
        rec(0, "cs", "cs_mark"), wait(0, "bs", "cs_mark"),
        op("B0", 0, "bs", ("r", 0, 0, 6), ("w", 1, 2, 4)),
        rec(0, "bs", "boundary"),
        op("I0", 0, "cs", ("r", 0, 3, 12), ("w", 1, 4, 10)),
    ]
    if with_wait:
        r.append(wait(0, "cs", "boundary"))
    r += [
        rec(0, "cs", "cs_mark"), wait(0, "bs", "cs_mark"),
        op("B1", 0, "bs", ("r", 1, 0, 6), ("w", 0, 2, 4)),
        rec(0, "bs", "boundary"),
        op("I1", 0, "cs", ("r", 1, 3, 12), ("w", 0, 4, 10)),  # writes buffer 0 planes 4..9; B0 read 0..5
    ]
    return r


def test_ordered_schedule_is_clean():
    assert SC.hazards(two_stream_sweep(True)) == []


def test_missing_wait_is_a_write_after_read_hazard():
    hz = SC.hazards(two_stream_sweep(False))
    kinds = {(k, a.name, b.name) for k, a, b, _, _ in hz}
    assert ("WAR", "B0", "I1") in kinds, kinds
    # and the read of what the boundary launch wrote (I1 reads buffer 1 planes 3.., B0 wrote 2..3) is a RAW
    assert ("RAW", "B0", "I1") in kinds, kinds


def test_wait_refers_to_the_latest_record_at_issue_time():
    """A wait issued BEFORE a later record of the same event does not cover the later work."""
    r = [
        op("A", 0, "bs", ("w", 0, 0, 4)), rec(0, "bs", "boundary"),
        wait(0, "cs", "boundary"),
        op("B", 0, "bs", ("w", 1, 0, 4)), rec(0, "bs", "boundary"),
        op("C", 0, "cs", ("r", 0, 0, 4), ("r", 1, 0, 4)),
    ]
    hz = SC.hazards(r)
    assert [(k, a.name, b.name) for k, a, b, _, _ in hz] == [("RAW", "B", "C")]


def test_wait_on_a_never_recorded_event_orders_nothing():
    r = [op("A", 0, "bs", ("w", 0, 0, 4)), wait(0, "cs", "halo"), op("B", 0, "cs", ("r", 0, 0, 4))]
    assert len(SC.hazards(r)) == 1


def test_disjoint_planes_and_transitive_edges():
    r = [
        op("A", 0, "cs", ("w", 0, 0, 4)), rec(0, "cs", "cs_mark"),
        wait(0, "bs", "cs_mark"), op("B", 0, "bs", ("w", 1, 0, 4)), rec(0, "bs", "boundary"),
        wait(1, "hs", "boundary", evslab=0), op("C", 1, "hs", ("r", 0, 0, 4), ("r", 1, 0, 4), ("w", 2, 0, 2)),  # A -> B -> C
        op("D", 0, "cs", ("w", 2, 2, 6)),  # other planes of buffer 2: no conflict with C
    ]
    assert SC.hazards(r) == []


def golden_files(kind):
    return sorted(glob.glob(os.path.join(GOLD, f"{kind}_*.jsonl")))


def test_golden_traces_exist():
    assert golden_files("head") and golden_files("inject"), "run tests/golden/make_schedule_golden.py through gpurun"


@pytest.mark.parametrize("path", golden_files("head"), ids=os.path.basename)
def test_production_schedule_traces_are_clean(path):
    ctxs, bad = SC.check_file(path)
    assert ctxs and not bad, SC.describe(bad)
    # a decomposed context really is in there (several streams, exchanges)
    assert any(c["ctx"]["P"] > 1 for c in ctxs)
    assert any(r["t"] == "xchg" for c in ctxs for r in c["records"])


@pytest.mark.parametrize("path", golden_files("inject"), ids=os.path.basename)
def test_round2_trapezoid_race_is_flagged(path):
    """Growth S_j instead of max(S_j, S_{j-1}): an interior launch overwrites the ping-pong buffer the previous, deeper
    boundary launch may still be reading (DESIGN.md §4 log, round 2). The checker must report exactly that pair."""
    ctxs, bad = SC.check_file(path, limit=10000)
    assert ctxs[0]["ctx"]["inject"] == 1
    assert bad, "the injected race was not detected"
    pairs = {(k, a.name[:6], a.stream[2], b.name[:6], b.stream[2]) for _, hz in bad for k, a, b, _, _ in hz}
    assert any(k == "WAR" and an == "jacobi" and as_ == "bs" and bn == "jacobi" and bs_ == "cs"
               for k, an, as_, bn, bs_ in pairs), pairs
