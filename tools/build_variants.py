#!/usr/bin/env python3
"""Build tuning variants of the HIP library: python tools/build_variants.py name=-DFLAG=V ...  ->
tools/variants/libsmart_amd_<name>.so (git-ignored; travels to the GPU box, unlike anything under smartpy_amd/csrc/ that
build.py does not produce: .gpurunignore).  Delete tools/variants/ when the A/B is done."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smartpy_amd import build as b

for spec in sys.argv[1:]:
    name, flags = spec.split('=', 1)
    os.makedirs(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'variants'), exist_ok=True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'variants', 'libsmart_amd_%s.so' % name)
    print(b.build(force=True, extra_flags=flags.split(), lib_path=path))
