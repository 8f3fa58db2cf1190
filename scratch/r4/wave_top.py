"""dev (-DRL_SPEC_TIMERS build): what the slowest waves of k_stream_spec spend their time on (cbox 1080p x 128 spp)."""
import os, sys, subprocess
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
env = dict(os.environ); env.update(dict(a.split('=') for a in sys.argv[1].split(',') if a) if len(sys.argv) > 1 else {}); env['RL_SPEC_STATS'] = '1'; env['RL_SPEC_WAVE_TIMES'] = '/tmp/wt.txt'; env['RL_SPEC_ONLY'] = '1'
out = subprocess.run([sys.executable, os.path.join(here, 'dev_spec.py'), '1920', '1080', '128', 'cbox'], env=env, capture_output=True, text=True)
print('\n'.join(l for l in (out.stdout + out.stderr).splitlines() if 'spec]' in l or 'chain_ms' in l))
a = np.loadtxt('/tmp/wt.txt'); d = a[:, 2] - a[:, 1]
print('waves %d; start time p50 %.1f max %.1f; duration mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f; end max %.1f' % (len(a), np.median(a[:, 1]), a[:, 1].max(), d.mean(), np.median(d), np.percentile(d, 90), np.percentile(d, 99), d.max(), a[:, 2].max()))
order = np.argsort(-a[:, 2])[:24]
print('wave  blocks(bx,by)          start   end   full  serial  ms_full ms_serial idle  slow')
G = int(env.get('RL_SPEC_GROUP', 32)); per = 64 // G
for w in order:
    blocks = [(int(b) // 68, int(b) % 68) for b in range(int(w) * per, int(w) * per + per)]
    print('%5d %-22s %6.1f %6.1f %6d %6d %7.1f %7.1f %5d %6d' % (w, blocks, a[w, 1], a[w, 2], a[w, 3], a[w, 4], a[w, 5], a[w, 6], a[w, 7], a[w, 8]))
for q in (50, 90, 99):
    sel = d >= np.percentile(d, q)
    print('waves above p%d: full %.0f serial %.0f iterations, %.1f + %.1f ms, slow samples %.0f' % (q, a[sel, 3].mean(), a[sel, 4].mean(), a[sel, 5].mean(), a[sel, 6].mean(), a[sel, 8].mean()))
