"""scripts/gpu_repro_seed.py against ANOTHER build of librekf.so (e.g. last round's, to tell a regression from arithmetic):
    python scripts/gpu_repro_seed_oldlib.py <lib.so> <abi> <seed>"""
import sys
sys.path.insert(0, ".")
from reflector_ekf_slam_amd import _lib
path, abi = sys.argv[1], int(sys.argv[2])
_lib.REKF_ABI_VERSION = abi
_lib.lib_path = lambda name, _p=path: _p if name == "librekf.so" else __import__("os").path.join(_lib._HERE, name)
sys.argv = [sys.argv[0], sys.argv[3]]
exec(open("scripts/gpu_repro_seed.py").read())
