"""Summarise the rocprofv3 --pmc passes over tools/pmc_probe2.py into profiles/r05/pmc_traffic[_feature].json.

    python tools/pmc_summary.py <dir with <pass>/p_counter_collection.csv> <out.json>
Passes (one rocprofv3 run each: counters of different blocks do not share a pass reliably):
    FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
    SQ_ACTIVE_INST_ANY | GRBM_GUI_ACTIVE | SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM
The probe launches every layer 4 times in a fixed order, so dispatch order identifies the layer.
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB): gfx950 FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md,
HBM section).  GRBM_GUI_ACTIVE is summed over the 8 XCDs: clock = counter / 8 / kernel duration."""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, out_path = sys.argv[1], sys.argv[2]
which = sys.argv[3] if len(sys.argv) > 3 else 'hr'
# order of tools/pmc_probe2.py <which>
LAYERS = {'hr': [(6, 128, 128), (5, 256, 256), (6, 384, 128), (5, 512, 512)],
          'hr_cond': [(6, 128, 128), (5, 256, 256), (6, 384, 128), (5, 512, 512)], 'feature': [(8, 64, 64), (8, 128, 64)]}[which]
NODES = {6: 108504, 5: 33800} if which == 'hr_cond' else {6: 217008, 5: 67600, 8: 3248400}
per = {}
for f in glob.glob(os.path.join(src, '*', '*counter_collection.csv')):
    byd = {}
    for r in csv.DictReader(open(f)):
        if 'gconv3_kernel' in r['Kernel_Name']:
            byd.setdefault(int(r['Dispatch_Id']), []).append(r)
    ids = sorted(byd)
    assert len(ids) == 4 * len(LAYERS), (f, len(ids))
    for k, did in enumerate(ids):
        d = per.setdefault(LAYERS[k // 4], {})
        for r in byd[did]:
            d.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
            d.setdefault('ns:' + r['Counter_Name'], []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
            d['kernel'], d['grid'] = r['Kernel_Name'].split('(')[0], int(r['Grid_Size'])
mean = lambda v: sum(v) / len(v)
h = hashlib.sha256()
for f in ('ofx_gemm3.hip', 'ofx_planes.h', 'ofx_gemm2.hip', 'ofx_gemm.hip', 'ofx_gemm_common.h'):
    h.update(open(os.path.join(ROOT, 'octfusion_amd', 'csrc', f), 'rb').read())
out = {'kernel_source_sha16': h.hexdigest()[:16], 'workload': which,
       'source': 'rocprofv3 --pmc, one pass per counter group, over tools/pmc_probe2.py on MI355X: 4 launches per layer '
                 'of gconv3_kernel (persistent planes GraphConv, default fp16-pair instantiation, emb + residual + fused statistics epilogue), shell-6 B=8 (hr_cond: shell-6 B=4; feature: shell-8 B=8); raw rows '
                 'in profiles/r06/pmc_*_probe2*.csv',
       'fetch_correction': 'HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB)',
       'clock_note': 'GRBM_GUI_ACTIVE is summed over the 8 XCDs: clock = counter / 8 / kernel duration', 'layers': []}
for L in LAYERS:
    d = per[L]
    dd, cin, cout = L
    flops = 2.0 * NODES[dd] * 7 * (cin + dd - 1) * cout
    ns = mean(d['ns:SQ_VALU_MFMA_BUSY_CYCLES'])
    busy = mean(d['SQ_VALU_MFMA_BUSY_CYCLES'])
    clk = mean(d['GRBM_GUI_ACTIVE']) / 8.0 / mean(d['ns:GRBM_GUI_ACTIVE'])
    wc = mean(d['SQ_WAVE_CYCLES'])
    out['layers'].append({
        'layer': 'depth %d, %d -> %d, N = %d' % (dd, cin, cout, NODES[dd]), 'kernel': d['kernel'], 'grid_size': d['grid'],
        'algorithmic_flops': flops, 'fetch_kb_raw': mean(d['FETCH_SIZE']), 'write_kb': mean(d['WRITE_SIZE']),
        'hbm_bytes_per_launch': 1024.0 * (2 * mean(d['FETCH_SIZE']) + mean(d['WRITE_SIZE'])),
        'launch_us_in_sq_pass': ns / 1e3, 'mfma_busy_cycles': busy, 'gpu_clock_ghz_under_kernel': clk,
        'mfma_busy_frac_of_clocked_simd_cycles': busy / (1024 * clk * ns), 'mfma_busy_frac_at_2.4GHz': busy / (1024 * 2.4 * ns),
        'mfma_ideal_cycles_three_term': flops * 3 / (2 * 32 * 32 * 16) * 32,
        'wave_cycle_split': {'issuing': mean(d['SQ_ACTIVE_INST_ANY']) / wc, 'stalled_at_issue': mean(d['SQ_WAIT_INST_ANY']) / wc,
                             'parked_waitcnt_or_barrier': mean(d['SQ_WAIT_ANY']) / wc},
        'lds_bank_conflict_cycles': mean(d['SQ_LDS_BANK_CONFLICT']), 'lds_idx_active_cycles': mean(d['SQ_LDS_IDX_ACTIVE'])})
out['kernels'] = sorted({L_['kernel'] for L_ in out['layers']})
first = out['layers'][0]
out['hbm_bytes_per_launch'] = first['hbm_bytes_per_launch']
out['hbm_bytes_per_launch_layer'] = first['layer']
out['mfma'] = {k: first[k] for k in ('mfma_busy_cycles', 'launch_us_in_sq_pass', 'gpu_clock_ghz_under_kernel',
                                     'mfma_busy_frac_of_clocked_simd_cycles', 'mfma_busy_frac_at_2.4GHz', 'wave_cycle_split')}
out['mfma']['layer'] = first['layer']
json.dump(out, open(out_path, 'w'), indent=1)
for r in out['layers']:
    print('%s: HBM %.0f MB, clock %.2f GHz, MFMA busy %.3f of clocked / %.3f of 2.4 GHz SIMD cycles' % (
        r['layer'], r['hbm_bytes_per_launch'] / 1e6, r['gpu_clock_ghz_under_kernel'],
        r['mfma_busy_frac_of_clocked_simd_cycles'], r['mfma_busy_frac_at_2.4GHz']))
