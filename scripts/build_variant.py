"""LAB: a second copy of the library with ONE source compiled under extra defines, for same-box A/Bs of a compile-time lab
knob:   python scripts/build_variant.py <name> <source.hip> -DKNOB[=v] ...   ->  ait_amd/libait_hip_<name>.so
A script selects it with  AIT_LAB_LIB=<name>  (scripts/_lab_lib.py; never read by the product)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ait_amd import build as B

name, src, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()
flags = B.COMMON + B.PER_FILE.get(src, []) + defs
obj = os.path.join(B.OBJ, src[:-4] + "." + name + ".o")
subprocess.check_call([B.hipcc()] + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj])
objs = [os.path.join(B.OBJ, s[:-4] + ".o") if s != src else obj for s in B.sources()]
out = os.path.join(B.HERE, "libait_hip_%s.so" % name)
subprocess.check_call([B.hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", out] + objs)
print(out)
