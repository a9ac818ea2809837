"""LAB: AIT_LAB_LIB=<name> points the ctypes loader at scripts/_lab/libait_hip_<name>.so (scripts/build_variant.py) -- import
this module before the first call into the library.  Scripts only; the product never reads the variable."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ait_amd import _lib
_n = os.environ.get("AIT_LAB_LIB")
if _n:
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lab", "libait_hip_%s.so" % _n)
    assert os.path.exists(_lib.LIB_PATH), _lib.LIB_PATH
    print("LAB library:", _lib.LIB_PATH, file=sys.stderr)
