"""probe: the drop-in binary end to end on a C2-sized cluster directory (text in, text out)"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import conftest, helpers
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
m, n, d = {"C2": (200, 100000, 10), "C3": (1000, 1000000, 10)}[name]
sc = helpers.vsfm_scene(m=m, n=n, d=d, seed=2)
wd = tempfile.mkdtemp(); path = os.path.join(wd, "outputVSFM_GB.nvm")
t0 = time.time(); helpers.write_nvm(path, sc); print("python wrote %.0f MB in %.1f s" % (os.path.getsize(path) / 1e6, time.time() - t0))
exe = os.path.join(os.getcwd(), "large-scale-sfm_amd", "bin", "optimizer_all")
t0 = time.time(); out = subprocess.run([exe, wd + "/", "0", "100", "16", "32", "0.5", "0"], capture_output=True, text=True); dt = time.time() - t0
print(out.stdout.strip()); print("process wall time %.2f s" % dt)
