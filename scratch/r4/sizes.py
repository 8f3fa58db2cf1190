import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from rustlight_amd import api, scenes
for name, sd in (('cbox', scenes.cbox(1920, 1080)), ('living', scenes.living_room(1920, 1080))):
    ctx = api.Context(api.Scene(sd), 0)
    print(name, ctx.debug_sizes())
