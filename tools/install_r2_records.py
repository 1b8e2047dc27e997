#!/usr/bin/env python3
"""Copies the records tools/r2_profile_run.sh and tools/r2_train_profile.sh left under gpurun_out/ into profiles/ (the
PMC extracts filtered to the path's own kernels) and prints the figures DESIGN.md quotes from them."""
import csv
import json
import shutil

O = 'gpurun_out/r2prof/'
shutil.copy(O + 'bench_default.json', 'profiles/r02_bench_default.json')
shutil.copy(O + 'eval_stats/eval_kernel_stats.csv', 'profiles/r02_eval_B1_b4096_single_stream_kernel_stats.csv')
shutil.copy(O + 'launch_table_B1_b4096.csv', 'profiles/r02_launch_table_B1_b4096.csv')
shutil.copy(O + 'r02_traffic_B1_weather96_b4096.json', 'profiles/r02_traffic_B1_weather96_b4096.json')
shutil.copy(O + 'rehearsal2.json', 'profiles/r02_rehearsal_2rank_gloo_host.json')
shutil.copy('gpurun_out/r2train/B1_weather96/t_kernel_stats.csv', 'profiles/r02_train_B1_b32_graph_kernel_stats.csv')
shutil.copy('gpurun_out/r2train/D_etth1_768/t_kernel_stats.csv', 'profiles/r02_train_D_b32_graph_kernel_stats.csv')
for src, dst, name in ((O + 'pmc_fetch/f_counter_collection.csv', 'profiles/r02_pmc_eval_B1_b4096_FETCH_SIZE.csv', 'FETCH_SIZE'),
                       (O + 'pmc_write/w_counter_collection.csv', 'profiles/r02_pmc_eval_B1_b4096_WRITE_SIZE.csv', 'WRITE_SIZE')):
    rows = list(csv.DictReader(open(src)))
    with open(dst, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Dispatch_Id', 'Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'Counter_Name', 'Counter_Value'])
        for r in rows:
            if r['Counter_Name'] == name and not r['Kernel_Name'].startswith('__amd') and 'at::native' not in r['Kernel_Name']:
                w.writerow([r['Dispatch_Id'], r['Kernel_Name'], r['Grid_Size'], r['Workgroup_Size'], r['Counter_Name'],
                            r['Counter_Value']])
rows = list(csv.DictReader(open('profiles/r02_eval_B1_b4096_single_stream_kernel_stats.csv')))
k3 = [r for r in rows if 'conv_fwd_h2_kernel<3' in r['Name']]
calls = sum(int(r['Calls']) for r in k3)
tot = sum(float(r['TotalDurationNs']) for r in k3)
r = json.loads(open('profiles/r02_bench_default.json').read().strip().splitlines()[-1])
rf = r['roofline']
tf = rf['flop_per_launch'] * rf['launches'] * 3 / (tot * 1e-9) * 1e-12
print("rocprof 3x3: %d launches, %.4f ms avg, %.1f TF, frac %.4f" % (calls, tot / calls / 1e6, tf, tf / 833.33))
print("bench: %.1f series/s, frac %.4f, %.1f TF, %.4f ms avg; traffic %.0f vs algorithmic %.0f" % (
    r['value'], rf['frac'], rf['achieved'], rf.get('avg_launch_ms', 0), rf['traffic'], rf['algorithmic_bytes_per_launch']))
print("cpu_baseline", r['cpu_baseline']['value'])
print("B1 train %.1f (%.2f ms) eager %.1f (%.2f ms)" % (r['train']['value'], r['train']['ms_per_step'],
                                                       r['train_eager']['value'], r['train_eager']['ms_per_step']))
d = r['extra_workloads']['D_etth1_768']
print("D %.1f series/s; train %.1f (%.2f ms) eager %.1f (%.2f ms)" % (d['value'], d['train']['value'], d['train']['ms_per_step'],
                                                                     d['train_eager']['value'], d['train_eager']['ms_per_step']))
for m in r.get('roofline_more', []):
    print(" ", m['kernel'], round(m['achieved'], 1), m['unit'], round(m['frac'], 3))
fam = {}
for x in rows:
    n = x['Name']
    k = ('k3' if 'conv_fwd_h2_kernel<3' in n else 'k1' if ('conv_fwd_h2_kernel<1' in n or 'persist' in n) else
         'gn' if ('gn_' in n or 'groupnorm' in n) else 'attn' if 'attn' in n else 'other')
    fam[k] = fam.get(k, 0) + float(x['TotalDurationNs']) / 3e6
print({k: round(v, 2) for k, v in fam.items()}, round(sum(fam.values()), 2))
