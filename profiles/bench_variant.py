"""run bench.py against another build of the library: python profiles/bench_variant.py path/to/lib.so [bench args]"""
import os, sys
sys.path.insert(0, '.')
from radarslampy_amd import _ffi
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
