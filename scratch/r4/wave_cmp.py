"""dev (-DRL_SPEC_TIMERS build): per-wave lifetimes of k_stream_spec for a list of env variants."""
import os, sys, json, subprocess
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
for v in sys.argv[1:]:
    env = dict(os.environ); env.update(dict(a.split('=') for a in v.split(',') if a)); env['RL_SPEC_STATS'] = '1'; env['RL_SPEC_WAVE_TIMES'] = '/tmp/wt.txt'; env['RL_SPEC_ONLY'] = '1'
    out = subprocess.run([sys.executable, os.path.join(here, 'dev_spec.py'), '1920', '1080', '128', 'cbox'], env=env, capture_output=True, text=True)
    print('==', v); print('\n'.join(l for l in (out.stdout + out.stderr).splitlines() if 'serial walks' in l or 'chain_ms' in l or 'slow walks' in l))
    a = np.loadtxt('/tmp/wt.txt'); d = a[:, 2] - a[:, 1]; hv = d > 1.0
    print('  heavy waves %d mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f; iterations full/serial mean %.0f / %.0f, slow samples mean %.0f; top 1%% waves: full %.0f serial %.0f slow %.0f' % (
        hv.sum(), d[hv].mean(), np.percentile(d[hv], 50), np.percentile(d[hv], 90), np.percentile(d[hv], 99), d.max(), a[hv, 3].mean(), a[hv, 4].mean(), a[hv, 8].mean(),
        a[d >= np.percentile(d[hv], 99), 3].mean(), a[d >= np.percentile(d[hv], 99), 4].mean(), a[d >= np.percentile(d[hv], 99), 8].mean()))
