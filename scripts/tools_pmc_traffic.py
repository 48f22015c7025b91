#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-launch HBM traffic of the decode-step GEMM.

Usage: tools_pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Both passes are separate rocprofv3 runs of the same bench command (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not
fit one pass).  Units/corrections per that guide: the counters are in KiB; on gfx950 FETCH_SIZE reports half the
bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is uncalibrated and taken as is.
"""
import csv, glob, json, os, sys
from collections import defaultdict


def load(d, counter, classes=None):
    """{kernel name: [counter value per launch]} in dispatch order.  The loader / consumer instance gemm_lc_kernel<true,..> serves three
    launch classes (bench.py's): [small] (< 128 workgroups: h2att), [stream] (>= 24 MB of weights: LSTM gates) and, r4, [segment]
    (the 17-MB token-embedding segment left of the attention-LSTM gate GEMM).  The class of a launch is decided by its FETCH_SIZE;
    the WRITE_SIZE pass -- a separate run of the same deterministic launch sequence -- takes the classes by position (`classes`)."""
    rows = defaultdict(list)
    recs = []
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                recs.append(r)
    recs.sort(key=lambda r: int(r.get('Dispatch_Id', 0)))
    seq, i = [], 0
    for r in recs:
        name = r['Kernel_Name']
        if 'gemm_lc_kernel<true' in name:
            try:
                wgs = int(r['Grid_Size']) // max(1, int(r['Workgroup_Size']))
            except (KeyError, ValueError):
                wgs = 0
            if classes is not None and i < len(classes):
                cls = classes[i]
            else:
                cls = ' [small]' if wgs < 128 else (' [stream]' if float(r['Counter_Value']) * 2048 >= 22e6 else ' [segment]')
            i += 1
            seq.append(cls)
            name = name[:name.rfind('(')] + cls + '('
        rows[name].append(float(r['Counter_Value']))
    return rows, seq


def main():
    fetch, write, out = sys.argv[1:4]
    F, seq = load(fetch, 'FETCH_SIZE')
    W, _ = load(write, 'WRITE_SIZE', classes=seq)
    res = {'unit': 'bytes per launch', 'corrections': 'FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read halving); WRITE_SIZE KiB x 1024', 'kernels': {}}
    for name in sorted(set(F) | set(W)):
        short = name.replace('(anonymous namespace)::', '').split('(')[0][-100:]
        f = F.get(name, []); w = W.get(name, [])
        res['kernels'][short] = {
            'launches': len(f) or len(w),
            'fetch_bytes_raw': sum(f) / len(f) * 1024 if f else None,
            'fetch_bytes_corrected': sum(f) / len(f) * 2048 if f else None,
            'write_bytes': sum(w) / len(w) * 1024 if w else None,
        }
    dec_f = [v for k, vs in F.items() if 'gemm_ares_kernel<true' in k or 'gemm_lc_kernel<true' in k for v in vs]
    dec_w = [v for k, vs in W.items() if 'gemm_ares_kernel<true' in k or 'gemm_lc_kernel<true' in k for v in vs]
    if dec_f and dec_w:
        res['decode_gemm'] = {
            'launches': len(dec_f),
            'fetch_bytes_corrected': sum(dec_f) / len(dec_f) * 2048,
            'write_bytes': sum(dec_w) / len(dec_w) * 1024,
        }
        res['decode_gemm']['traffic_bytes'] = res['decode_gemm']['fetch_bytes_corrected'] + res['decode_gemm']['write_bytes']
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res.get('decode_gemm')))


if __name__ == '__main__':
    main()
