"""LAB: bench.py's step with a lab build of the library (AIT_LAB_LIB=<name>, scripts/build_variant.py) -- for same-box
A/Bs of compile-time knobs.  Same arguments as bench.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _lab_lib  # noqa: F401
import bench
bench.main()
