"""Dev: ASAN + UBSAN build of the file readers (meshio.cpp / io.cpp with stubs, no HIP) fuzzed byte-wise over a generated corpus.
  g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -Iinclude scratch/asan_fuzz/main.cpp scratch/asan_fuzz/stubs.cpp \
      rustlight_amd/csrc/host/meshio.cpp rustlight_amd/csrc/host/io.cpp -lz -o /tmp/asan/fz && python scratch/asan_fuzz/run.py 200
Round 1: 5800 runs over OBJ / PLY / serialized / EXR / PNG / JPEG / TGA / PFM, no finding."""
import os, random, subprocess, sys, shutil
sys.path.insert(0, '/root/repo')
import numpy as np
from rustlight_amd import export, scenes
D = '/tmp/asan/corpus'; shutil.rmtree(D, ignore_errors=True); os.makedirs(D)
sd = scenes.cbox(32, 32); sd.flip = True; sd.fov_axis = 0
for fmt in ('obj', 'ply', 'serialized'): export.write_mitsuba(sd, f'{D}/{fmt}.xml', fmt)
img = np.random.default_rng(1).uniform(0, 3, (21, 13, 3)).astype(np.float32)
export.write_exr(img, f'{D}/z.exr', 'zip', True); export.write_exr(img, f'{D}/r.exr', 'rle', False, extra_channel=True); export.write_exr(img, f'{D}/n.exr', 'none', False, data_origin=(2, -5)); export.write_exr(img, f'{D}/s.exr', 'zips', True)
export.write_png(np.random.default_rng(0).integers(0, 256, (9, 7, 3), dtype=np.uint8), f'{D}/t.png')
fx = np.load('/root/repo/tests/golden/jpeg_fixture.npz')
for k in ('yuv420_q60_rst', 'yuv422_q75_opt', 'gray_q80', 'progressive_q80', 'yuv444_q90'): open(f'{D}/{k}.jpg', 'wb').write(fx[k + '_file'].tobytes())
import struct
open(f'{D}/a.tga', 'wb').write(struct.pack("<BBBHHBHHHHBB", 0, 0, 10, 0, 0, 0, 0, 0, 5, 4, 24, 0) + bytes([0x83, 1, 2, 3, 0x03] + list(range(12)) + [0x8b, 9, 9, 9]))
scenes.write_pbrt(scenes.sky_scene(16, 16), f'{D}/sky.pbrt')
files = [f for f in sorted(os.listdir(D)) if f.split('.')[-1] in ('obj', 'ply', 'serialized', 'exr', 'png', 'jpg', 'tga', 'pfm')]
rnd = random.Random(3); bad = 0; n = 0
for f in files:
    orig = open(f'{D}/{f}', 'rb').read()
    ext = f.split('.')[-1]
    for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
        data = bytearray(orig); mode = trial % 5
        if mode == 0: data = data[:rnd.randrange(0, len(data))]
        elif mode == 1:
            for _ in range(rnd.randrange(1, 10)): data[rnd.randrange(len(data))] = rnd.randrange(256)
        elif mode == 2: i = rnd.randrange(len(data)); data[i:i] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 40)))
        elif mode == 3: i = rnd.randrange(len(data)); del data[i:i + rnd.randrange(1, 60)]
        else:
            i = rnd.randrange(len(data)); data[i] = rnd.choice([0, 0xff, 0x7f, 0x80])
        p = f'/tmp/asan/case.{ext}'
        open(p, 'wb').write(bytes(data))
        r = subprocess.run(['/tmp/asan/fz', p], capture_output=True, text=True, timeout=60)
        n += 1
        if r.returncode != 0 or 'runtime error' in r.stderr or 'AddressSanitizer' in r.stderr:
            bad += 1
            shutil.copy(p, f'/tmp/asan/bad_{bad}.{ext}')
            print('FAIL', f, 'mode', mode, 'rc', r.returncode, '\n'.join(r.stderr.strip().splitlines()[:6]), flush=True)
            if bad > 8: break
    if bad > 8: break
print('runs', n, 'failures', bad)
