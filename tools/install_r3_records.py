#!/usr/bin/env python3
"""Copy the round-3 measurement records that tools/r3_profile_run.sh (and the bench / probe calls named below) left
under gpurun_out/ into profiles/ (tracked): rocprofv3 kernel statistics of one single-stream sampler evaluation (B1 at
4096 series, D at 1024), the FETCH_SIZE / WRITE_SIZE passes and the per-launch traffic summaries bench.py reads, the
launch tables, the steady-state train-step statistics (difference of the 60-step and 10-step runs), bench lines and
the conv3x3_h2_m16 probe / ablation logs.  Missing inputs are reported and skipped."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r3prof")
DST = os.path.join(ROOT, "profiles")
done, missing = [], []


def put(src, name):
    if src and os.path.exists(src):
        shutil.copyfile(src, os.path.join(DST, name))
        done.append(name)
    else:
        missing.append(name)


def first(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    return hits[0] if hits else None


for wl, b in (("B1_weather96", 4096), ("D_etth1_768", 1024)):
    put(first(f"{SRC}/eval_{wl}/**/*kernel_stats.csv"), f"r03_eval_{wl}_b{b}_single_stream_kernel_stats.csv")
    put(first(f"{SRC}/pmc_fetch_{wl}/**/*counter_collection.csv"), f"r03_pmc_eval_{wl}_b{b}_FETCH_SIZE.csv")
    put(first(f"{SRC}/pmc_write_{wl}/**/*counter_collection.csv"), f"r03_pmc_eval_{wl}_b{b}_WRITE_SIZE.csv")
    put(f"{SRC}/r03_traffic_{wl}_b{b}.json", f"r03_traffic_{wl}_b{b}.json")
    put(f"{SRC}/launch_table_{wl}_b{b}.csv", f"r03_launch_table_{wl}_b{b}.csv")
    put(f"{SRC}/bench_{wl}.json", f"r03_bench_{wl}_b{b}_for_launch_table.json")
    s10 = first(f"{SRC}/train_{wl}_10/**/*kernel_stats.csv")
    s60 = first(f"{SRC}/train_{wl}_60/**/*kernel_stats.csv")
    put(s60, f"r03_train_{wl}_b32_graph_kernel_stats_60steps.csv")
    put(f"{SRC}/train_{wl}_60.json", f"r03_train_{wl}_b32_bench_under_rocprof_60steps.json")
    if s10 and s60:
        out = os.path.join(DST, f"r03_train_{wl}_b32_steady_state_step_kernel_stats.csv")
        txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kstats_diff.py"), s10, s60, "10", "60", out],
                             capture_output=True, text=True, check=True).stdout
        open(os.path.join(DST, f"r03_train_{wl}_b32_steady_state_step.txt"), "w").write(txt)
        done += [os.path.basename(out), f"r03_train_{wl}_b32_steady_state_step.txt"]
    else:
        missing.append(f"r03_train_{wl}_b32_steady_state_step_kernel_stats.csv")

GO = os.path.join(ROOT, "gpurun_out")
for src, name in (("r3_bench_default_final.json", "r03_bench_default.json"),
                  ("r3_rehearsal_2rank.json", "r03_rehearsal_2rank_gloo_host.json"),
                  ("r3_m16_pair_probe.log", "r03_conv3x3_m16_probe.log"),
                  ("r3_m16_ablate4.log", "r03_conv3x3_m16_ablation.log"),
                  ("r3_m16_zero2.log", "r03_conv3x3_m16_zero_data.log"),
                  ("r3_mfma_shape_probe.txt", "r03_mfma_shape_probe.txt"),
                  ("r3_full_gpu_final.log", "r03_gpu_tests_final.log")):
    put(os.path.join(GO, src), name)
print("installed:", *done, sep="\n  ")
if missing:
    print("MISSING:", *missing, sep="\n  ")
