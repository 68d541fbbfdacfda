"""Summarise rocprofv3 --pmc CSVs of tools/pmc_probe2.py into profiles/r02/pmc_traffic.json.

    python tools/pmc_summary.py <dir with *_counter_collection.csv from the passes> <out.json>
Per gconv2_kernel launch group (grid size identifies the layer): HBM bytes = 2 x FETCH_SIZE (KB; gfx950 reports half
the bytes of 16-B/lane coalesced reads -- MI355X_MICROARCH.md, HBM section) + WRITE_SIZE (KB), MFMA-busy cycles,
GRBM_GUI_ACTIVE -> clock."""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, out = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gconv2_kernel' not in r['Kernel_Name']:
            continue
        key = (r['Kernel_Name'].split('(')[0], int(r['Grid_Size']))
        d = acc.setdefault(key, {})
        d.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        d.setdefault('_dur_' + r['Counter_Name'], []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
h = hashlib.sha256()
for f in ('ofx_gemm2.hip', 'ofx_gemm.hip', 'ofx_gemm_common.h'):
    h.update(open(os.path.join(ROOT, 'octfusion_amd', 'csrc', f), 'rb').read())
res = {'kernel_source_sha16': h.hexdigest()[:16], 'workload': 'hr',
       'source': 'rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_* | GRBM_GUI_ACTIVE, one pass each) over '
                 'tools/pmc_probe2.py on MI355X: 4 launches per layer of gconv2_kernel (planes GraphConv), shell-6 B=8',
       'fetch_correction': 'x2 (gfx950 FETCH_SIZE counts 64 B per 128-B request; MI355X_MICROARCH.md)', 'layers': []}
mean = lambda v: sum(v) / len(v)
for (name, grid), d in sorted(acc.items()):
    L = {'kernel': name, 'grid_size': grid}
    if 'FETCH_SIZE' in d:
        L['fetch_kb_raw'] = mean(d['FETCH_SIZE'])
    if 'WRITE_SIZE' in d:
        L['write_kb'] = mean(d['WRITE_SIZE'])
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        L['hbm_bytes_per_launch'] = 1024.0 * (2.0 * L['fetch_kb_raw'] + L['write_kb'])
    for k in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY',
              'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU_MFMA_MOPS_BF16', 'GRBM_GUI_ACTIVE', 'SQ_LDS_BANK_CONFLICT',
              'SQ_LDS_IDX_ACTIVE', 'SQ_INST_CYCLES_VMEM', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM'):
        if k in d:
            L[k] = mean(d[k])
            L['launch_ns_in_pass_' + k] = mean(d['_dur_' + k])
    res['layers'].append(L)
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1)[:3000])
