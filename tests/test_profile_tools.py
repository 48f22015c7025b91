"""The profile post-processing scripts the numbers in profiles/ come from (CPU only, synthetic rocprofv3 output)."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LC = 'void capmi_gemm::(anonymous namespace)::gemm_lc_kernel<true, 2, 0, 0>(capmi_gemm::KArgs)'


def _counter_csv(path, counter, rows):
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, 'x_counter_collection.csv'), 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=['Dispatch_Id', 'Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'Counter_Name', 'Counter_Value'])
        w.writeheader()
        for i, (name, wgs, val) in enumerate(rows):
            w.writerow(dict(Dispatch_Id=i + 1, Kernel_Name=name, Grid_Size=wgs * 768, Workgroup_Size=768, Counter_Name=counter, Counter_Value=val))


def test_pmc_traffic_classes_follow_the_fetch_pass_by_position(tmp_path):
    """scripts/tools_pmc_traffic.py: gemm_lc_kernel<true,..> launches are [small] (< 128 workgroups), [stream] (>= 22 MB fetched after
    the gfx950 x2 correction) or [segment]; FETCH_SIZE and WRITE_SIZE come from separate runs of the same launch sequence, so the
    write pass takes its classes from the fetch pass by position -- a segment launch writes as much as a stream launch"""
    seq = [(LC, 256, 23000.0), (LC, 16, 1400.0), (LC, 256, 9500.0), ('other_kernel(int)', 64, 100.0), (LC, 256, 24000.0)]
    _counter_csv(tmp_path / 'f', 'FETCH_SIZE', seq)
    _counter_csv(tmp_path / 'w', 'WRITE_SIZE', [(n, g, 7000.0 if n == LC and g == 256 else 1900.0) for n, g, _ in seq])
    out = tmp_path / 'o.json'
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'scripts', 'tools_pmc_traffic.py'), str(tmp_path / 'f'), str(tmp_path / 'w'), str(out)],
                          stdout=subprocess.DEVNULL)
    k = json.load(open(out))['kernels']
    by = {name.split('[')[1].rstrip('] '): v for name, v in k.items() if '[' in name}
    assert by['stream']['launches'] == 2 and by['segment']['launches'] == 1 and by['small']['launches'] == 1
    assert abs(by['stream']['fetch_bytes_corrected'] - 23500.0 * 2048) < 1 and abs(by['segment']['fetch_bytes_corrected'] - 9500.0 * 2048) < 1
    assert abs(by['stream']['write_bytes'] - 7000.0 * 1024) < 1 and abs(by['segment']['write_bytes'] - 7000.0 * 1024) < 1
    assert abs(by['small']['write_bytes'] - 1900.0 * 1024) < 1


def test_prof_gaps_reports_idle_time_between_kernels(tmp_path):
    """scripts/prof_gaps.py on a synthetic kernel trace: three Adam-delimited steps, a 40-us hole after every 'a' kernel"""
    import sqlite3
    db = sqlite3.connect(str(tmp_path / 't.db'))
    db.execute('create table kernels (start integer, end integer, name text)')
    t = 0
    for step in range(4):
        for name, dur, gap in (('a_kernel', 10_000, 40_000), ('b_kernel', 20_000, 1_000), ('adam2_kernel', 5_000, 2_000)):
            db.execute('insert into kernels values (?, ?, ?)', (t, t + dur, name))
            t += dur + gap
    db.commit()
    db.close()
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, 'scripts', 'prof_gaps.py'), str(tmp_path / 't.db'), '5', '5', '2'], text=True)
    assert 'last 2 steps' in out
    line = [l for l in out.split('\n') if 'after a_kernel' in l][0]
    assert '80.0 us in    2 gaps' in line and 'before b_kernel' in line
