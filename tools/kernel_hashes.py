#!/usr/bin/env python3
"""A hash of the instruction stream of every smart_fast_* kernel of the built library (or of the library named on the
command line): an edit that is meant to leave a kernel's code alone shows here whether it did."""
import hashlib
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_report as R      # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_names():
    text = open(os.path.join(ROOT, 'smartpy_amd', 'csrc', 'smart_capi.hip')).read()
    table = re.search(r'kFastKernelNames\[kNumFastKernels\] = \{(.*?)\};', text, re.S).group(1)
    return re.findall(r'"(smart_fast_\w+)"', table)


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'smartpy_amd', 'csrc', 'libsmart_amd.so')
    for k in kernel_names():
        start, sym, body = R.disassemble(lib, k)
        ins = R.parse(start, body)
        h = hashlib.sha256('\n'.join(x['op'] + ' ' + x['args'] for x in ins).encode()).hexdigest()[:12]
        print('%-30s %6d instructions  %s' % (k, len(ins), h))


if __name__ == '__main__':
    main()
