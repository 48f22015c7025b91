#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass into per-kernel MFMA utilisation.

Usage: tools_pmc_mfma.py <pmc_dir> <out.json>
utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs): the fraction of SIMD-cycles during which
the matrix pipe was busy while the kernel ran.  rocprofv3 reports both counters summed over the chip: BUSY over all 1024
SIMDs, GUI_ACTIVE over the 8 XCDs (checked on gemm_ares<false,7,2,false>: 24.8 M busy cycles for 0.39 M MFMAs x 64 cycles).
Kernels run serialised and slower under counter collection, so this is a lower bound of the un-instrumented utilisation (MI355X_MICROARCH.md: the counter ticks in cycles, 32 per
v_mfma_f32_32x32x16_bf16, 64 per v_mfma_f32_32x32x2_f32).
"""
import csv, glob, json, os, sys
from collections import defaultdict


def main():
    d, out = sys.argv[1:3]
    busy, act = defaultdict(list), defaultdict(list)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            tgt = busy if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES' else act if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' else None
            if tgt is not None:
                tgt[r['Kernel_Name']].append(float(r['Counter_Value']))
    res = {'definition': 'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 * 4)', 'kernels': {}}
    for name in sorted(busy):
        if not act.get(name) or sum(busy[name]) == 0:
            continue
        short = name.replace('(anonymous namespace)::', '').split('(')[0][-100:]
        b, a = sum(busy[name]) / len(busy[name]), sum(act[name]) / len(act[name])
        res['kernels'][short] = {'launches': len(busy[name]), 'mfma_busy_cycles': round(b), 'gui_active_cycles': round(a),
                                 'mfma_util': round(b / (a / 8.0 * 1024.0), 4)}
    json.dump(res, open(out, 'w'), indent=1)
    for k, v in sorted(res['kernels'].items(), key=lambda kv: -kv[1]['mfma_util']):
        print('%-90s n=%4d util=%.3f' % (k[-90:], v['launches'], v['mfma_util']))


if __name__ == '__main__':
    main()
