"""Deterministic hazard check of libsfgpu.so's multi-stream launch schedule.

Input: the JSON-lines trace `SF_TRACE_SCHEDULE=<file>` makes a context write (fluidsolvergpu_amd/csrc/sf_solver.hpp,
"schedule trace"): every launch / exchange with the plane ranges it reads (stencil reach included) and writes per
buffer, and every hipEventRecord / hipStreamWaitEvent. This module rebuilds the happens-before relation the HIP
runtime guarantees — program order within a stream, plus record -> wait edges (a wait refers to the event's latest
record at the time the wait is issued; a wait on a never-recorded event is a no-op) — as vector clocks, and reports
every pair of accesses to overlapping planes of one buffer, at least one of them a write, that the relation leaves
unordered (RAW / WAR / WAW). It needs no GPU and does not depend on timing: a race that shows once in fifty runs on
the device is a missing edge here every time.

Counterpart in the reference: the hand-placed cudaDeviceSynchronize / cudaMemcpy sequence of its two-GPU exchange
(solver-unidyn.cu:396-470), which is correct by being fully serial.
"""
import json
from collections import defaultdict


class Op:
    __slots__ = ("idx", "name", "stream", "seq", "vc", "acc", "line")

    def __init__(self, idx, name, stream, seq, vc, acc, line):
        self.idx, self.name, self.stream, self.seq, self.vc, self.acc, self.line = idx, name, stream, seq, vc, acc, line

    def __repr__(self):
        return f"#{self.idx} {self.name}@{self.stream[1]}.{self.stream[2]} (line {self.line})"


def split_contexts(lines):
    """A trace file holds one or more contexts, each opened by a {"t":"ctx"} line."""
    out, cur = [], None
    for n, ln in enumerate(lines, 1):
        ln = ln.strip()
        if not ln:
            continue
        rec = json.loads(ln)
        rec["_line"] = n
        if rec["t"] == "ctx":
            cur = {"ctx": rec, "records": []}
            out.append(cur)
        elif cur is not None:
            cur["records"].append(rec)
    return out


def build(records):
    """-> (ops, exchanges). Streams are (0, slab, name)."""
    seq = defaultdict(int)  # stream -> ops issued
    clock = defaultdict(dict)  # stream -> vector clock {stream: seq}
    events = {}  # (slab, ev) -> vector clock at its latest record
    ops, xchg = [], []
    for r in records:
        t = r["t"]
        if t == "op":
            s = (0, r["slab"], r["stream"])
            seq[s] += 1
            vc = dict(clock[s])
            vc[s] = seq[s]
            clock[s] = vc
            ops.append(Op(len(ops), r["name"], s, seq[s], vc, [tuple(a) for a in r["acc"]], r["_line"]))
        elif t == "rec":
            s = (0, r["slab"], r["stream"])
            vc = dict(clock[s])
            vc[s] = seq[s]
            events[(r["slab"], r["ev"])] = vc
        elif t == "wait":
            s = (0, r["slab"], r["stream"])
            ev = events.get((r["evslab"], r["ev"]))
            if ev is not None:
                vc = dict(clock[s])
                for k, v in ev.items():
                    if vc.get(k, 0) < v:
                        vc[k] = v
                clock[s] = vc
        elif t == "xchg":
            xchg.append((r["seq"], r["G"], tuple(r["fields"])))
    return ops, xchg


def happens_before(a, b):
    return b.vc.get(a.stream, 0) >= a.seq


def hazards(records, limit=20):
    """List of (kind, earlier op, later op, buffer, overlap) for every unordered conflicting pair."""
    ops, _ = build(records)
    per_buf = defaultdict(list)
    for op in ops:
        for kind, buf, lo, hi in op.acc:
            per_buf[buf].append((op, kind == "w", lo, hi))
    found = []
    for buf, accs in per_buf.items():
        writes = [a for a in accs if a[1]]
        for w in writes:
            for a in accs:
                if a[0] is w[0]:
                    continue
                if a[1] and a[0].idx < w[0].idx:
                    continue  # a write-write pair is visited once, from its earlier member
                lo, hi = max(a[2], w[2]), min(a[3], w[3])
                if lo >= hi:
                    continue
                first, second = (a, w) if a[0].idx < w[0].idx else (w, a)
                if happens_before(first[0], second[0]):
                    continue
                kind = "WAW" if (a[1] and w[1]) else ("WAR" if second[1] else "RAW")
                found.append((kind, first[0], second[0], buf, (lo, hi)))
                if len(found) >= limit:
                    return found
    return found


def check_file(path, limit=20):
    """-> list of (context record, hazards) for every context of the trace that has any."""
    with open(path) as f:
        ctxs = split_contexts(f.readlines())
    bad = []
    for c in ctxs:
        h = hazards(c["records"], limit)
        if h:
            bad.append((c["ctx"], h))
    return ctxs, bad


def describe(bad):
    out = []
    for ctx, hz in bad:
        out.append(f"context N={ctx['N']} P={ctx['P']} G={ctx['G']} nzl={ctx['nzl']} trap={ctx['trap']}:")
        for kind, a, b, buf, (lo, hi) in hz:
            out.append(f"  {kind} on buffer {buf} planes [{lo},{hi}): {a!r} is not ordered before {b!r}")
    return "\n".join(out)


def exchange_sequence(path):
    """The (seq, G, fields) list of every context: what tests/slab_emulator.py must reproduce."""
    with open(path) as f:
        return [build(c["records"])[1] for c in split_contexts(f.readlines())]


if __name__ == "__main__":
    import sys

    for p in sys.argv[1:]:
        ctxs, bad = check_file(p)
        nops = sum(len([r for r in c["records"] if r["t"] == "op"]) for c in ctxs)
        print(f"{p}: {len(ctxs)} context(s), {nops} ops, {'CLEAN' if not bad else 'HAZARDS'}")
        if bad:
            print(describe(bad))
