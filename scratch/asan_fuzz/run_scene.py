"""Dev: the same for the scene loaders (pbrt.cpp, mitsuba.cpp + scene / bvh / lighttree / meshio / io, no HIP):
  H=rustlight_amd/csrc/host; g++ -O1 -g -std=c++17 -fsanitize=address,undefined -ffp-contract=off -Iinclude scratch/asan_fuzz/main_scene.cpp scratch/asan_fuzz/stubs_scene.cpp \
      $H/meshio.cpp $H/io.cpp $H/scene.cpp $H/bvh.cpp $H/pbrt.cpp $H/mitsuba.cpp $H/lighttree.cpp -lz -o /tmp/asan/fz2 && python scratch/asan_fuzz/run_scene.py 150
Round 1: 450 runs (truncate / replace / insert / delete / duplicate), no finding, no leak."""
import os, random, subprocess, sys, shutil
sys.path.insert(0, '/root/repo')
import numpy as np
from rustlight_amd import export, scenes
D = '/tmp/asan/corpus2'; shutil.rmtree(D, ignore_errors=True); os.makedirs(D)
sd = scenes.cbox(32, 32); sd.flip = True; sd.fov_axis = 0
sd.lights.append({"type": "point", "a": (0.2, 1.2, 0.1), "intensity": (1.0, 2.0, 3.0)})
sd.medium = scenes.Medium((0.01, 0.02, 0.03), (0.5, 0.4, 0.3), scenes.PHASE_HG, 0.3)
export.write_mitsuba(sd, f'{D}/m.xml', 'obj')
scenes.write_pbrt(scenes.sky_scene(16, 16, keep_area_light=True), f'{D}/sky.pbrt')
scenes.write_pbrt(scenes.many_lights(16, 16, 2), f'{D}/ml.pbrt')
rnd = random.Random(5); bad = 0; n = 0
for f in ('m.xml', 'sky.pbrt', 'ml.pbrt'):
    orig = open(f'{D}/{f}', 'rb').read()
    for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
        data = bytearray(orig); mode = trial % 5
        if mode == 0: data = data[:rnd.randrange(0, len(data))]
        elif mode == 1:
            for _ in range(rnd.randrange(1, 10)): data[rnd.randrange(len(data))] = rnd.randrange(32, 127)
        elif mode == 2: i = rnd.randrange(len(data)); data[i:i] = bytes(rnd.choice(b' "[]<>/=0123456789.-eE\n') for _ in range(rnd.randrange(1, 20)))
        elif mode == 3: i = rnd.randrange(len(data)); del data[i:i + rnd.randrange(1, 60)]
        else:
            i = rnd.randrange(len(data)); j = rnd.randrange(len(data)); a, b = min(i, j), max(i, j); data[a:a] = data[a:min(b, a + 200)]
        p = f'{D}/case_{f}'
        open(p, 'wb').write(bytes(data))
        r = subprocess.run(['/tmp/asan/fz2', p], capture_output=True, text=True, timeout=120)
        n += 1
        if r.returncode != 0 or 'runtime error' in r.stderr or 'AddressSanitizer' in r.stderr:
            bad += 1
            shutil.copy(p, f'/tmp/asan/bad2_{bad}_{f}')
            print('FAIL', f, 'mode', mode, 'rc', r.returncode, '\n'.join(r.stderr.strip().splitlines()[:8]), flush=True)
            if bad > 5: break
    if bad > 5: break
print('runs', n, 'failures', bad)
