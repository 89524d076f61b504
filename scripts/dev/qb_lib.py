"""Dev helper: quick_search_bench with an alternative build of libsss (scripts/dev/<name>.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sessionsimilaritysearch_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), sys.argv[1])
import ctypes
import torch                                    # first: libsss must bind to the HIP runtime torch loads, not a second copy
_have = ctypes.CDLL(_lib.LIB_PATH)
for _name in list(_lib._SIGNATURES):            # an older build may lack entry points added since: A/B runs do not need them
    if not hasattr(_have, _name):
        del _lib._SIGNATURES[_name]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quick_search_bench as qb
for a in sys.argv[2:]:
    qb.run(*tuple(int(v) if v.isdigit() else v for v in a.split(",")))
