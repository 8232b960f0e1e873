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

shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
for f in glob.glob(os.path.join(src, "bench_stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "rank_share.jsonl")):
    shutil.copy(os.path.join(src, "rank_share.jsonl"), os.path.join(dst, f"{tag}_rank_share.jsonl"))


def counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def kernel_avg_ns(path):
    out = {}
    for f in glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            out[row["Name"].split("(")[0]] = (int(row["Calls"]), float(row["AverageNs"]))
    return out


traffic = {}
lines = [f"# Jacobi lin_solve sweep — rocprofv3 PMC summary ({tag})", "",
         "Separate `--pmc` passes (FETCH_SIZE | WRITE_SIZE | TCC_HIT/MISS | TCC_EA0_RD/WRREQ) plus a `--kernel-trace --stats`",
         "pass of `python3 tools/jacobi_sweep.py N` (6 sweeps after 2 warm-up sweeps). FETCH_SIZE / WRITE_SIZE are in KiB;",
         "on gfx950 FETCH_SIZE counts 128-byte requests as 64 B, so read bytes = 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM).",
         "jacobi2_* kernels perform TWO sweeps per launch: compare their bytes with 2 x the algorithmic bytes.", ""]
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
    alg = N ** 3 * 12
    lines += [f"## N = {N} fp32 (algorithmic bytes per SWEEP: {alg / 1e6:.1f} MB; a jacobi2 launch is two sweeps)", "",
              "| kernel | launches | avg ns (kernel-trace) | FETCH_SIZE KiB | read MB (x2) | WRITE_SIZE KiB | written MB | HBM bytes / algorithmic bytes of the launch | TCC hit rate |",
              "|---|---|---|---|---|---|---|---|---|"]
    for k, cs in sorted(merged.items()):
        if "jacobi" not in k:
            continue
        fetch = cs.get("FETCH_SIZE", (0, 0))[1]
        write = cs.get("WRITE_SIZE", (0, 0))[1]
        hit, miss = cs.get("TCC_HIT_sum", (0, 0))[1], cs.get("TCC_MISS_sum", (0, 0))[1]
        rd, wr = 2 * fetch * 1024, write * 1024
        calls, ns = times.get(k, (0, 0.0))
        lines.append(f"| `{k[:60]}` | {calls} | {ns:.0f} | {fetch:.0f} | {rd / 1e6:.1f} | {write:.0f} | {wr / 1e6:.1f} | "
                     f"{(rd + wr) / (alg * (2 if 'jacobi2' in k else 1)):.3f} | {hit / max(hit + miss, 1):.3f} |")
        if calls >= 2 or f"jacobi_nf1_f32_{N}" not in traffic:
            traffic[f"jacobi_nf1_f32_{N}"] = rd + wr  # HBM bytes per LAUNCH (a jacobi2 launch is two sweeps)
    lines.append("")
open(os.path.join(dst, f"{tag}_jacobi_pmc.md"), "w").write("\n".join(lines))
json.dump(traffic, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
print("\n".join(lines))
print(json.dumps(traffic))
