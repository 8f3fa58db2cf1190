"""Dev: byte-level fuzz of the scene / mesh / image file readers (truncate, flip, insert, delete), each case in a subprocess — a malformed
file must come back as an error through the C-ABI, never crash.  usage: python scratch/fuzz_loaders.py"""
import os, sys, subprocess, random, shutil
sys.path.insert(0, '/root/repo')
import numpy as np
from rustlight_amd import api, export, scenes
D = '/tmp/fuzz_dir'; shutil.rmtree(D, ignore_errors=True); os.makedirs(D)
sd = scenes.cbox(32, 32); sd.flip = True; sd.fov_axis = 0
for fmt in ('obj', 'ply', 'serialized'):
    export.write_mitsuba(sd, f'{D}/{fmt}.xml', fmt)
scenes.write_pbrt(scenes.sky_scene(32, 32, keep_area_light=True), f'{D}/sky.pbrt')
export.write_png(np.random.default_rng(0).integers(0, 256, (9, 7, 3), dtype=np.uint8), f'{D}/t.png')
img = np.random.default_rng(1).uniform(0, 3, (21, 13, 3)).astype(np.float32)
export.write_exr(img, f'{D}/z.exr', 'zip', True); export.write_exr(img, f'{D}/r.exr', 'rle', False, extra_channel=True); export.write_exr(img, f'{D}/n.exr', 'none', False, data_origin=(2, -5))
fx = np.load('/root/repo/tests/golden/jpeg_fixture.npz')
for k in ('yuv420_q60_rst', 'yuv422_q75_opt', 'gray_q80', 'progressive_q80'): open(f'{D}/{k}.jpg', 'wb').write(fx[k + '_file'].tobytes())
targets = [(f'{D}/obj.xml', 'scene'), (f'{D}/obj_3.obj', 'dep:obj.xml'), (f'{D}/ply_0.ply', 'dep:ply.xml'), (f'{D}/ply_1.ply', 'dep:ply.xml'),
           (f'{D}/serialized.serialized', 'dep:serialized.xml'), (f'{D}/sky.pbrt', 'scene'), (f'{D}/sky_env.pfm', 'dep:sky.pbrt'), (f'{D}/t.png', 'image'), (f'{D}/z.exr', 'image'), (f'{D}/r.exr', 'image'), (f'{D}/n.exr', 'image'), (f'{D}/yuv420_q60_rst.jpg', 'image'), (f'{D}/yuv422_q75_opt.jpg', 'image'), (f'{D}/gray_q80.jpg', 'image'), (f'{D}/progressive_q80.jpg', 'image')]
code = '''
import sys; sys.path.insert(0, "/root/repo")
from rustlight_amd import api
kind, path = sys.argv[1], sys.argv[2]
try:
    if kind == "image": api.load_image(path)
    else: api.Scene.load(path)
    print("ok")
except api.RustlightError as e:
    print("err")
'''
rnd = random.Random(1)
bad = 0; n = 0
for path, kind in targets:
    orig = open(path, 'rb').read()
    main = path if not kind.startswith('dep:') else f'{D}/' + kind[4:]
    k = 'image' if kind == 'image' else 'scene'
    for trial in range(40):
        data = bytearray(orig)
        mode = trial % 4
        if mode == 0: data = data[:rnd.randrange(0, len(data))]
        elif mode == 1:
            for _ in range(8): data[rnd.randrange(len(data))] = rnd.randrange(256)
        elif mode == 2:
            i = rnd.randrange(len(data)); data[i:i] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 40)))
        else:
            i = rnd.randrange(len(data)); del data[i:i + rnd.randrange(1, 60)]
        open(path, 'wb').write(bytes(data))
        r = subprocess.run([sys.executable, '-c', code, k, main], capture_output=True, text=True, timeout=60)
        n += 1
        if r.returncode != 0:
            bad += 1
            print('CRASH', os.path.basename(path), 'mode', mode, 'rc', r.returncode, r.stderr.strip().splitlines()[-1:] )
    open(path, 'wb').write(orig)
print('runs', n, 'crashes', bad)
