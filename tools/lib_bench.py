"""bench.py against another build of the library (probe builds: `make stamp`, `make occ1`):  python tools/lib_bench.py libmi_ba_occ1.so [bench.py arguments]"""
import importlib, os, runpy, sys
sys.path.insert(0, os.getcwd())
capi = importlib.import_module("large-scale-sfm_amd.capi")
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
