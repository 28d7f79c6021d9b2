import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
capi = importlib.import_module("large-scale-sfm_amd.capi"); scene = importlib.import_module("large-scale-sfm_amd.scene")
sc = scene.make_scene(sys.argv[1] if len(sys.argv) > 1 else "C3")
eng = capi.Engine(capi.default_config()); eng.load_scene(sc); eng.prepare()
ret, info = eng.run()
tr = eng.trace()
print("it  mu        cost_trial      acc pcg")
for r in tr: print("%2d  %.2e  %.6e  %d   %3d" % (r[0], r[1], r[3], r[4], r[5]))
