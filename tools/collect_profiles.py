#!/usr/bin/env python3
"""Copy the judged summaries from gpurun_out/ (scratch) into profiles/ (tracked).  Usage: collect_profiles.py rNN [workload]"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
wl = sys.argv[2] if len(sys.argv) > 2 else "C3"
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
out = {}
for name in ("fetch", "write", "sq", "lds", "grbm"):
    f = os.path.join(ROOT, "gpurun_out", f"pmc_{name}", "pmc_counter_collection.csv")
    if not os.path.exists(f):
        continue
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        out.setdefault(k, {}).update({c: round(sum(x) / len(x), 1) for c, x in v.items()})
if out:
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_{wl}.json"), "w"), indent=1)
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "prof", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_bench_{wl}_kernel_stats.csv"))
b = os.path.join(ROOT, "gpurun_out", "bench.log")
if os.path.exists(b) and os.path.getsize(b) > 10:
    shutil.copy(b, os.path.join(ROOT, "profiles", f"{tag}_bench_{wl}.json"))
pj = os.path.join(ROOT, "gpurun_out", f"pmc_{wl}.json")          # written by `bench.py --pmc` on the GPU box (tools/gpu_evidence.sh)
if os.path.exists(pj):       # ONE tracked copy per round (bench.py reads the newest rNN_pmc_<wl>.json; it carries the library's sha256)
    shutil.copy(pj, os.path.join(ROOT, "profiles", f"{tag}_pmc_{wl}.json"))
for name in ("C3_s2", "C3_s4", "C3_fill", "C3_random", "C3_morton"):                       # sensitivity lines
    b = os.path.join(ROOT, "gpurun_out", f"bench_{name}.log")
    if os.path.exists(b) and os.path.getsize(b) > 10:
        shutil.copy(b, os.path.join(ROOT, "profiles", f"{tag}_bench_{name}.json"))
for other in ("C1", "C2", "C4", "C5"):
    b = os.path.join(ROOT, "gpurun_out", f"bench_{other}.log")
    if os.path.exists(b) and os.path.getsize(b) > 10:
        shutil.copy(b, os.path.join(ROOT, "profiles", f"{tag}_bench_{other}.json"))
for other in ("C3", "C4"):
    b = os.path.join(ROOT, "gpurun_out", f"bench_field_mlp_{other}.log")
    if os.path.exists(b) and os.path.getsize(b) > 10:
        shutil.copy(b, os.path.join(ROOT, "profiles", f"{tag}_bench_field_mlp_{other}.json"))
for src, dst in (("prof_mlp/mlp_kernel_stats.csv", f"{tag}_field_mlp_kernel_stats.csv"),
                 ("prof_step/step_kernel_stats.csv", f"{tag}_training_step_color_kernel_stats.csv"),
                 ("prof_step_color_summary.txt", f"{tag}_training_step_color_summary.txt")):
    f = os.path.join(ROOT, "gpurun_out", src)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(ROOT, "profiles", dst))
for src, dst in (("bench_color_train.log", f"{tag}_bench_color_train_C4.json"), ("force_cut_C3_k1.log", f"{tag}_force_cut_C3_k1.json"),
                 ("force_cut_C3_k2.log", f"{tag}_force_cut_C3_k2.json"), ("force_cut_C3_k4.log", f"{tag}_force_cut_C3_k4.json"),
                 ("force_cut_C4_k1.log", f"{tag}_force_cut_C4_k1.json"), ("force_cut_C3_k4_sequential.log", f"{tag}_force_cut_C3_k4_sequential.json"), (f"composite_diag_{wl}.json", f"{tag}_composite_diag_{wl}.json")):
    f = os.path.join(ROOT, "gpurun_out", src)
    if os.path.exists(f) and os.path.getsize(f) > 10:
        shutil.copy(f, os.path.join(ROOT, "profiles", dst))
for src, dst in (("bench_C3_views2.log", f"{tag}_bench_C3_views2.json"), ("bench_C3_views8.log", f"{tag}_bench_C3_views8.json"),
                 ("pmc_C3_views4.json", f"{tag}_pmc_C3_views4.json"), ("prof_mlp_summary.txt", f"{tag}_field_mlp_summary.txt"),
                 ("pytest_gpu.log", f"{tag}_pytest_gpu_tail.txt"), ("driver_like.log", f"{tag}_driver_like.log")):
    f = os.path.join(ROOT, "gpurun_out", src)
    if os.path.exists(f) and os.path.getsize(f) > 10:
        shutil.copy(f, os.path.join(ROOT, "profiles", dst))
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "prof_views", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_views4_{wl}_kernel_stats.csv"))
for m in ("frame_cycle", "train_cycle", "pair_cycle"):
    f = os.path.join(ROOT, "gpurun_out", f"prof_host_{m}.txt")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_host_profile_{m}.txt"))
print(sorted(os.listdir(os.path.join(ROOT, "profiles"))))
