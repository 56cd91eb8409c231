"""gpurun_out/prof_r6/ (experiments/profile_r6.sh) -> profiles/: r06_<config>[_bilinear]_kernel_stats.csv (our kernels' rows), r06_<config>_pmc.txt and
one traffic_<config>_<budget>.json per config for bench.py's roofline.traffic (FETCH_SIZE x 2: calibration in profiles/r02_counter_calibration.json)."""
import csv, json, re
SRC, DST = 'gpurun_out/prof_r6', 'profiles'
def pmc(path, kernel_prefix):
    cur, out = None, {}
    for line in open(path):
        if not line.startswith(' '):
            cur = line.strip()
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+mean=([0-9.e+]+)", line)
            if m and cur and kernel_prefix in cur: out[m.group(1)] = (float(m.group(3)), int(m.group(2)), cur)
    return out
import os
budgets = {'c1': 7168, 'c2': 7168, 'c3': 7168, 'c5': 7168, 'c4shard': 7168, 'c5shard': 7168, 'c1_bilinear': 7168, 'c2_bilinear': 7168, 'c3_bilinear': 7168, 'c5_bilinear': 7168}
for cfg, bud in budgets.items():
    if not os.path.exists(f'{SRC}/{cfg}_bench.json'):
        continue
    b = json.load(open(f'{SRC}/{cfg}_bench.json'))  # (bench.py --detail: the verbose record)
    bil = cfg.endswith('_bilinear')
    kern = ('pb_bilinear_double_hot_kernel' if cfg.startswith('c5') else 'pb_bilinear_hot_kernel') if bil else ('pb_hot_double_kernel' if cfg.startswith('c5') else 'pb_hot_win_kernel')
    f = pmc(f'{SRC}/{cfg}_pmc_FETCH_SIZE.txt', kern)['FETCH_SIZE']
    w = pmc(f'{SRC}/{cfg}_pmc_WRITE_SIZE.txt', kern)['WRITE_SIZE']
    r = b['roofline']
    assert r['window_budget'] == bud, (cfg, r['window_budget'])
    hbm = int(2 * f[0] * 1024 + w[0] * 1024)
    rows = [row for row in csv.reader(open(f'{SRC}/{cfg}_kernel_stats.csv'))]
    keep = [rows[0]] + [row for row in rows[1:] if row[0].startswith(('void pb_', 'pb_'))]
    csv.writer(open(f'{DST}/r06_{cfg}_kernel_stats.csv', 'w')).writerows(keep)
    with open(f'{DST}/r06_{cfg}_pmc.txt', 'w') as out:
        for c in ('FETCH_SIZE', 'WRITE_SIZE'):
            out.write(f'# rocprofv3 --kernel-trace --pmc {c} (own pass)\n' + open(f'{SRC}/{cfg}_pmc_{c}.txt').read())
    kst = [row for row in keep[1:] if kern in row[0]][0]
    t = {
        'kernel': f[2], 'config': cfg, 'window_budget': bud, 'frames_per_launch': b['config']['frames_per_launch'], 'round': 6,
        'fetch_size_kb_raw': f[0], 'write_size_kb_raw': w[0], 'pmc_dispatches': f[1],
        'correction': 'FETCH_SIZE x 2 (calibrated in round 2 for 16-B streams and unaligned dword gathers: profiles/r02_counter_calibration.json); WRITE_SIZE as reported',
        'hbm_bytes_per_launch': hbm,
        'algorithmic_bytes_per_launch': r['algorithmic_bytes_per_launch'], 'must_move_bytes_per_launch': r['must_move_bytes_per_launch'],
        'traffic_over_algorithmic': round(hbm / r['algorithmic_bytes_per_launch'], 3), 'traffic_over_must_move': round(hbm / r['must_move_bytes_per_launch'], 3),
        'rocprof_kernel_avg_ns': float(kst[3]), 'rocprof_calls': int(kst[1]),
        'bench_kernel_ms_mean_same_run': r['kernel_ms_mean'], 'plan': r['plan'],
        'source_files': [f'profiles/r06_{cfg}_kernel_stats.csv', f'profiles/r06_{cfg}_pmc.txt', 'experiments/profile_r6.sh'],
    }
    if bil:
        t['bilinear_window_budget'] = 12288  # the mode's own launch table (PB_BIL_WIN_BUDGET); `window_budget` above is the nearest mode's
    json.dump(t, open(f'{DST}/traffic_{cfg}_{bud}.json', 'w'), indent=1)
    print(cfg, bud, 'fetch x2 %.1f MB write %.1f MB total %.1f MB = %.2fx algorithmic, %.2fx must-move; rocprof avg %.2f us vs hipEvent %.2f us' % (
        2 * f[0] * 1024 / 1e6, w[0] * 1024 / 1e6, hbm / 1e6, t['traffic_over_algorithmic'], t['traffic_over_must_move'], t['rocprof_kernel_avg_ns'] / 1e3, r['kernel_ms_mean'] * 1e3))
