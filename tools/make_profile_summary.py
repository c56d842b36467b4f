"""Turn the rocprofv3 CSVs of tools/gpu_profile.sh (gpurun_out/prof_<tag>/) into the committed summaries
under profiles/<round>/.   usage: python tools/make_profile_summary.py <tag> <round-dir>"""
import collections
import csv
import json
import os
import shutil
import sys

tag, out = sys.argv[1], sys.argv[2]
src = os.path.join('gpurun_out', 'prof_' + tag)
os.makedirs(out, exist_ok=True)
if os.path.isfile(os.path.join(src, 'trace', 'bench_kernel_stats.csv')):
    shutil.copy(os.path.join(src, 'trace', 'bench_kernel_stats.csv'), os.path.join(out, 'bench_kernel_stats.csv'))
if os.path.isfile(os.path.join(src, 'counters_available.txt')):
    shutil.copy(os.path.join(src, 'counters_available.txt'), os.path.join(out, 'counters_available.txt'))
for sub, name in (('trace_vanilla', 'vanilla_kernel_stats.csv'), ('trace_bf16x3', 'bf16x3_kernel_stats.csv'),
                  ('trace_fp16x2', 'fp16x2_kernel_stats.csv'), ('trace_volume', 'volume_kernel_stats.csv')):
    if os.path.isfile(os.path.join(src, sub, name)):
        shutil.copy(os.path.join(src, sub, name), os.path.join(out, name))
for log, name in (('trace_stdout.log', 'bench_profiled.json'), ('bf16x3_stdout.log', 'bench_bf16x3_profiled.json'),
                  ('fp16x2_stdout.log', 'bench_fp16x2_profiled.json'),
                  ('vanilla_stdout.log', 'vanilla_profiled.json'), ('volume_stdout.log', 'volume_profiled.json')):
    if os.path.isfile(os.path.join(src, log)):
        js = [l for l in open(os.path.join(src, log)).read().split('\n') if l.startswith('{')]
        if js:
            open(os.path.join(out, name), 'w').write(js[-1] + '\n')
raw, rows = {}, []
for name in sorted(d for d in os.listdir(src) if d.startswith('pmc') and os.path.isdir(os.path.join(src, d))):
    acc = collections.defaultdict(list)
    if not os.path.isfile(os.path.join(src, name, 'pmc_counter_collection.csv')):      # a counter name this rocprofv3 does not know
        continue
    for r in csv.DictReader(open(os.path.join(src, name, 'pmc_counter_collection.csv'))):
        k = r['Kernel_Name']
        if 'p2s_' not in k:
            continue
        short = k.split('p2s_')[1].split('(')[0]
        if short.startswith('chain_kernel<'):          # r04: template <bool SUM>; <false> = the max-pool kernel of every model
            short = 'chain_kernel' if short == 'chain_kernel<false>' else short
        # r04: template <NS, F16, PIPE, SUM>; the default (max-pool, r03 schedule) instances keep their r03 names
        short = short.replace('chain_bf16_kernel<2, true, false, false>', 'chain_bf16_kernel<2, true>') \
                     .replace('chain_bf16_kernel<3, false, false, false>', 'chain_bf16_kernel<3, false>') \
                     .replace('chain_bf16_kernel<1, false, false, false>', 'chain_bf16_kernel<1>')
        # r05: template <NS, F16, SUM>
        short = {'chain_bf16_kernel<2, true, false>': 'chain_bf16_kernel<2, true>',
                 'chain_bf16_kernel<3, false, false>': 'chain_bf16_kernel<3, false>',
                 'chain_bf16_kernel<1, false, false>': 'chain_bf16_kernel<1>'}.get(short, short)
        acc[(short, r['Counter_Name'])].append(float(r['Counter_Value']))
        rows.append([name, short, r['Dispatch_Id'], r['Grid_Size'], r['Workgroup_Size'], r['LDS_Block_Size'],
                     r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'], r['Counter_Name'], r['Counter_Value'],
                     int(r['End_Timestamp']) - int(r['Start_Timestamp'])])
    for (k, c), v in acc.items():
        raw.setdefault('p2s_' + k, {})[c] = {'dispatches': len(v), 'avg_per_dispatch': sum(v) / len(v)}
with open(os.path.join(out, 'pmc_p2s_kernels.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['pass', 'kernel', 'dispatch', 'grid', 'wg', 'lds_bytes', 'vgpr', 'agpr', 'sgpr', 'counter', 'value', 'duration_ns'])
    w.writerows(rows)
ch = raw['p2s_chain_kernel']
dur = [r[-1] for r in rows if r[1] == 'chain_kernel' and r[9] == 'GRBM_GUI_ACTIVE']
fetch = ch['FETCH_SIZE']['avg_per_dispatch'] * 1024 * 2      # KB -> B, x2 gfx950 correction (MI355X_MICROARCH.md, HBM)
write = ch['WRITE_SIZE']['avg_per_dispatch'] * 1024
cyc = ch['GRBM_GUI_ACTIVE']['avg_per_dispatch'] / 8
summ = {'workload': 'tools/quick_bench.py --B 4096 --iters 1 (p2s_max encoders+decoder, 4096 synthetic queries): 6 chain launches',
        'chain_kernel': {
            'queries_per_launch': 4096,
            'avg_duration_ms_under_pmc': sum(dur) / len(dur) / 1e6,
            'cycles_per_launch': cyc, 'clock_GHz': cyc / (sum(dur) / len(dur)),
            'mfma_busy_frac': ch['SQ_VALU_MFMA_BUSY_CYCLES']['avg_per_dispatch'] / (1024 * cyc),
            'executed_mfma_flop_per_launch': ch['SQ_INSTS_VALU_MFMA_MOPS_F32']['avg_per_dispatch'] * 512,
            'fetch_bytes_per_launch_corrected_x2': fetch, 'write_bytes_per_launch': write,
            'hbm_traffic_bytes_per_launch': fetch + write,
            'lds_bank_conflict_cycles': ch['SQ_LDS_BANK_CONFLICT']['avg_per_dispatch']},
        'raw': raw}
def split_kernel(names, label, workload, mops):
    key = next((n for n in names if n in raw), None)
    if not key:
        return
    bf = raw[key]
    short = key[len('p2s_'):]
    durb = [r[-1] for r in rows if r[1] == short and r[9] == 'GRBM_GUI_ACTIVE']
    cycb = bf['GRBM_GUI_ACTIVE']['avg_per_dispatch'] / 8
    summ[label] = {
        'workload': workload, 'kernel': key,
        'queries_per_launch': 4096, 'avg_duration_ms_under_pmc': sum(durb) / len(durb) / 1e6,
        'cycles_per_launch': cycb, 'clock_GHz': cycb / (sum(durb) / len(durb)),
        'mfma_busy_frac': bf['SQ_VALU_MFMA_BUSY_CYCLES']['avg_per_dispatch'] / (1024 * cycb),
        'executed_mfma_flop_per_launch': bf[mops]['avg_per_dispatch'] * 512 if mops in bf else None,
        'fetch_bytes_per_launch_corrected_x2': bf['FETCH_SIZE']['avg_per_dispatch'] * 1024 * 2 if 'FETCH_SIZE' in bf else None,
        'write_bytes_per_launch': bf['WRITE_SIZE']['avg_per_dispatch'] * 1024 if 'WRITE_SIZE' in bf else None,
        'lds_bank_conflict_cycles': bf['SQ_LDS_BANK_CONFLICT']['avg_per_dispatch'] if 'SQ_LDS_BANK_CONFLICT' in bf else None}


split_kernel(('p2s_chain_bf16_kernel<3, false>', 'p2s_chain_bf16_kernel<3>'), 'chain_bf16x3_kernel',
             'tools/quick_bench.py --B 4096 --iters 1 --bf16 3', 'SQ_INSTS_VALU_MFMA_MOPS_BF16')
split_kernel(('p2s_chain_bf16_kernel<2, true>',), 'chain_fp16x2_kernel',
             'tools/quick_bench.py --B 4096 --iters 1 --bf16 4 (fp16 pair per operand)', 'SQ_INSTS_VALU_MFMA_MOPS_F16')
b1 = raw.get('p2s_chain_bf16_kernel<1>')
if b1:
    dur1 = [r[-1] for r in rows if r[1] == 'chain_bf16_kernel<1>' and r[9] == 'GRBM_GUI_ACTIVE']
    cyc1 = b1['GRBM_GUI_ACTIVE']['avg_per_dispatch'] / 8
    summ['chain_bf16_kernel'] = {
        'workload': 'tools/quick_bench.py --B 4096 --iters 1 --bf16 1 (plain bf16: outside the 1e-4 contract)',
        'queries_per_launch': 4096, 'avg_duration_ms_under_pmc': sum(dur1) / len(dur1) / 1e6,
        'cycles_per_launch': cyc1, 'clock_GHz': cyc1 / (sum(dur1) / len(dur1)),
        'mfma_busy_frac': b1['SQ_VALU_MFMA_BUSY_CYCLES']['avg_per_dispatch'] / (1024 * cyc1),
        'executed_mfma_bf16_flop_per_launch': b1['SQ_INSTS_VALU_MFMA_MOPS_BF16']['avg_per_dispatch'] * 512,
        'fetch_bytes_per_launch_corrected_x2': b1['FETCH_SIZE']['avg_per_dispatch'] * 1024 * 2,
        'write_bytes_per_launch': b1['WRITE_SIZE']['avg_per_dispatch'] * 1024,
        'lds_bank_conflict_cycles': b1['SQ_LDS_BANK_CONFLICT']['avg_per_dispatch']}
json.dump(summ, open(os.path.join(out, 'pmc_summary.json'), 'w'), indent=1)
print(json.dumps(summ['chain_kernel'], indent=1))
print(json.dumps(summ.get('chain_bf16x3_kernel'), indent=1))
print(json.dumps(summ.get('chain_fp16x2_kernel'), indent=1))
print(json.dumps(summ.get('chain_bf16_kernel'), indent=1))
