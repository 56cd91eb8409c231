"""Same-box A/B of library builds: the probe in alternating child processes, ROUNDS rounds.
    python experiments/r6/ab.py ROUNDS "case ..." label=lib.so[:ENV=VAL] ..."""
import json, os, subprocess, sys
root = os.environ.get('GRAFT_REPO_ROOT', os.getcwd())
rounds, cases, specs = int(sys.argv[1]), sys.argv[2].split(), sys.argv[3:]
acc = {}
for r in range(rounds):
    for spec in specs:
        label, rest = spec.split('=', 1)
        parts = rest.split(':')
        env = dict(os.environ)
        if parts[0]:
            env['PB_LIB_PATH'] = os.path.join(root, parts[0])
        for kv in parts[1:]:
            k, v = kv.split('=')
            env[k] = v
        out = subprocess.run([sys.executable, 'experiments/r6/c5_bil_probe.py', *cases, '--json', '--bil-only'], env=env, capture_output=True, text=True, timeout=300)
        if out.returncode != 0:
            print(label, 'FAILED', out.stderr[-2000:]); sys.exit(1)
        for line in out.stdout.splitlines():
            if line.startswith('{'):
                j = json.loads(line)
                acc.setdefault((label, j['case']), []).append(j['us_min_p10_med']['bilinear'])
                acc.setdefault(('shape', label, j['case']), j['shape'])
for k, v in acc.items():
    if k[0] == 'shape':
        continue
    print(f"{k[0]:12s} {k[1]:8s} min {min(x[0] for x in v):7.2f}  p10s {[x[1] for x in v]}  medians {[x[2] for x in v]}  {acc[('shape',) + k]}", flush=True)
