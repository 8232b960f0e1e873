"""tools/summarize_profiles.py <round-tag> — turn gpurun_out/profiles_<tag>/ (tools/collect_profiles.sh) into the
tracked summaries under profiles/: <tag>_bench.json, <tag>_bench_kernel_stats.csv, <tag>_jacobi_pmc.md and
traffic_latest.json (HBM bytes per Jacobi launch, FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"profiles_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)



def newest(pattern):
    """gpurun MERGES a call's output into the local gpurun_out/: a second collection leaves the first one's files
    (named by process id) beside its own. Only the newest process's files of a directory count."""
    by_dir = defaultdict(list)
    for f in glob.glob(pattern, recursive=True):
        by_dir[os.path.dirname(f)].append(f)
    out = []
    for fs in by_dir.values():
        pid = os.path.basename(max(fs, key=os.path.getmtime)).split("_")[0]
        out += [f for f in fs if os.path.basename(f).split("_")[0] == pid]
    return out


shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
for f in newest(os.path.join(src, "bench_stats", "**", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "rank_share.jsonl")):
    shutil.copy(os.path.join(src, "rank_share.jsonl"), os.path.join(dst, f"{tag}_rank_share.jsonl"))


def counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    for f in newest(os.path.join(path, "**", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def kernel_avg_ns(path):
    out = {}
    for f in newest(os.path.join(path, "**", "*kernel_stats.csv")):
        for row in csv.DictReader(open(f)):
            out[row["Name"].split("(")[0]] = (int(row["Calls"]), float(row["AverageNs"]))
    return out


import re


def sweeps_of(kernel):
    """Sweeps per launch from the kernel's name: jacobi_sk_kernel<T, NF, WL, NT, S, ...> carries S; jacobi2_* is two."""
    m = re.match(r".*jacobi_sk_kernel<[^,]+, \d+, \d+, (?:true|false), (\d+)", kernel)
    if m:
        return int(m.group(1))
    return 2 if "jacobi2" in kernel else 1


traffic = {}
lines = [f"# Jacobi lin_solve — rocprofv3 PMC summary ({tag})", "",
         "Separate `--pmc` passes (FETCH_SIZE | WRITE_SIZE | TCC_HIT/MISS | TCP_TCC_READ_REQ, TCC_EA0_RD/WRREQ) plus a",
         "`--kernel-trace --stats` pass of `SF_SWEEP_K=20 python3 tools/jacobi_sweep.py N`: one 20-sweep solve after a",
         "2-sweep warm-up (one launch of the register-blocked pair kernel `jacobi2_kernel`) = 5 launches of the four-sweep",
         "marching kernel `jacobi_sk_kernel<T, NF, WL, NT, S = 4, TJ, NW, ISH, FIRST>`: the first reads caller data on the",
         "i-shell (FIRST = 1), three plain ones, the last writes the i-shell (ISH = true).",
         "FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 B, so read bytes =",
         "2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM). `alg` = 3 words x N^3 x sweeps of the launch (SURVEY.md §8d).", ""]
for N in (256, 512):
    base = os.path.join(src, f"pmc_{N}")
    if not os.path.isdir(base):
        continue
    merged = defaultdict(dict)
    for sub in ("fetch", "write", "tcc", "req"):
        for k, cs in counters(os.path.join(base, sub)).items():
            for c, v in cs.items():
                merged[k][c] = (len(v), sum(v) / len(v))
    times = kernel_avg_ns(os.path.join(base, "trace"))
    lines += [f"## N = {N} fp32 (algorithmic bytes per SWEEP: {N ** 3 * 12 / 1e6:.1f} MB)", "",
              "| kernel | sweeps | launches | avg us (kernel-trace) | read MB (FETCH x2) | written MB | counter bytes / alg bytes | "
              "counter TB/s | alg TB/s (frac of 8) | L1->L2 read requests (M) | TCC hit rate |",
              "|---|---|---|---|---|---|---|---|---|---|---|"]
    for k, cs in sorted(merged.items()):
        if "jacobi" not in k:
            continue
        S = sweeps_of(k)
        alg = N ** 3 * 12 * S
        fetch = cs.get("FETCH_SIZE", (0, 0))[1]
        write = cs.get("WRITE_SIZE", (0, 0))[1]
        hit, miss = cs.get("TCC_HIT_sum", (0, 0))[1], cs.get("TCC_MISS_sum", (0, 0))[1]
        req = cs.get("TCP_TCC_READ_REQ_sum", (0, 0))[1]
        rd, wr = 2 * fetch * 1024, write * 1024
        calls, ns = times.get(k, (0, 0.0))
        if ns <= 0:
            continue
        lines.append(f"| `{k[:72]}` | {S} | {calls} | {ns / 1e3:.1f} | {rd / 1e6:.1f} | {wr / 1e6:.1f} | {(rd + wr) / alg:.3f} | "
                     f"{(rd + wr) / ns / 1e3:.2f} | {alg / ns / 1e3:.2f} ({alg / ns / 8e3:.2f}) | {req / 1e6:.2f} | {hit / max(hit + miss, 1):.3f} |")
        plain = re.search(r"false(, 0)?>$", k.strip()) is not None  # no i-shell writes, not a first pass
        if S == 4 and (plain or f"jacobi_f32_{N}" not in traffic):
            traffic[f"jacobi_f32_{N}"] = {"bytes_per_launch": rd + wr, "kernel": k[:80], "tag": tag,
                                          "source": f"profiles/{tag}_jacobi_pmc.md (FETCH_SIZE x2 + WRITE_SIZE, separate "
                                                    "rocprofv3 --pmc passes of tools/jacobi_sweep.py)",
                                          "kernel_trace_us": ns / 1e3}
    lines.append("")
open(os.path.join(dst, f"{tag}_jacobi_pmc.md"), "w").write("\n".join(lines))
json.dump(traffic, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
print("\n".join(lines))

# ---- instruction-issue picture: marching kernel against the round-1 pair kernel -------------------------------
issue = [f"# Jacobi kernels — instruction-issue counters ({tag})", "",
         "`tools/collect_profiles.sh`: separate `rocprofv3 --pmc` passes over `SF_SWEEP_K=20 python3 tools/jacobi_sweep.py N`,",
         "with the default build (four-sweep marching kernel `jacobi_sk_kernel<..,4,4,8,..>`) and with",
         "`SF_MARCH=0` (the register-blocked pair kernel `jacobi2_kernel` of round 1 for every pass). SQ_INSTS_* are sums over",
         "all waves; *_CYCLES / ACTIVE / WAIT are in quad-cycles (x4 clocks). `per cell-sweep` divides by N^3 x sweeps of the",
         "launch — the figure that can be compared across kernels that fuse a different number of sweeps.", ""]
for N in (256, 512):
    for label, sub in (("marching (default)", f"sq_{N}"), ("pair kernel (SF_MARCH=0)", f"sq_old_{N}")):
        base = os.path.join(src, sub)
        if not os.path.isdir(base):
            continue
        merged = defaultdict(dict)
        for part in ("insts", "active", "wait"):
            for k, cs in counters(os.path.join(base, part)).items():
                for c, v in cs.items():
                    merged[k][c] = sum(v) / len(v)
        times = kernel_avg_ns(os.path.join(base, "trace"))
        for k, cs in sorted(merged.items()):
            if "jacobi" not in k or "SQ_WAVES" not in cs:
                continue
            calls, ns = times.get(k, (0, 0.0))
            if calls < 2:
                continue
            S = sweeps_of(k)
            cells = N ** 3 * S
            w = cs["SQ_WAVES"]
            issue += [f"## N = {N}, {label}: `{k[:70]}` — {S} sweeps per launch, {ns / 1e3:.1f} us per launch "
                      f"({ns / 1e3 / S:.1f} us per sweep), {w:.0f} waves", "",
                      "| counter | per launch | per wave | per 1000 cell-sweeps |", "|---|---|---|---|"]
            for c in sorted(cs):
                issue.append(f"| {c} | {cs[c]:.4g} | {cs[c] / w:.1f} | {cs[c] / cells * 1000:.3f} |")
            wc = cs.get("SQ_WAVE_CYCLES", 0)
            if wc:
                issue += ["", f"wave-cycle split: issuing {cs.get('SQ_ACTIVE_INST_ANY', 0) / wc:.0%}, "
                              f"SQ_WAIT_ANY (s_waitcnt / barrier) {cs.get('SQ_WAIT_ANY', 0) / wc:.0%}, "
                              f"SQ_WAIT_INST_ANY (issue stall) {cs.get('SQ_WAIT_INST_ANY', 0) / wc:.0%}"]
            issue.append("")
open(os.path.join(dst, f"{tag}_pair_kernel_issue.md"), "w").write("\n".join(issue))
print("\n".join(issue))
if os.path.exists(os.path.join(src, "configs.txt")):
    shutil.copy(os.path.join(src, "configs.txt"), os.path.join(dst, f"{tag}_configs.txt"))
print(json.dumps(traffic))
