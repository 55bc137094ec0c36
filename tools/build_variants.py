#!/usr/bin/env python3
"""Build tuning variants of the HIP library next to the default one: python tools/build_variants.py name=-DFLAG=V ..."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smartpy_amd import build as b

for spec in sys.argv[1:]:
    name, flags = spec.split('=', 1)
    path = os.path.join(b.CSRC, 'libsmart_amd_%s.so' % name)
    print(b.build(force=True, extra_flags=flags.split(), lib_path=path))
