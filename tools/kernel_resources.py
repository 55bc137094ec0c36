#!/usr/bin/env python3
"""Register / scratch / occupancy figures of every gfx950 kernel of the library, as hipcc reports them
(-Rpass-analysis=kernel-resource-usage).  usage: python tools/kernel_resources.py > profiles/<round>_kernel_resources.txt"""
import os
import re
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'smartpy_amd', 'csrc')
UNITS = {'smart_fast_intervals': ['-ffp-contract=fast-honor-pragmas', '-fno-honor-nans'],
         'smart_fast_runs': ['-ffp-contract=fast-honor-pragmas', '-fno-honor-nans'],
         'smart_fast_steps': ['-ffp-contract=fast-honor-pragmas', '-fno-honor-nans'],
         'smart_fast_reports': ['-ffp-contract=fast-honor-pragmas', '-fno-honor-nans'],
         'smart_fast_guarded': ['-ffp-contract=fast-honor-pragmas'],
         'smart_literal': ['-ffp-contract=off'], 'smart_capi': []}
ver = subprocess.check_output(['/opt/rocm/bin/hipcc', '--version']).decode().splitlines()[0]
print('hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage (%s)' % ver)
print("(rocprofv3's VGPR_Count column shows half the allocation: 176 allocated -> 88)")
print('%-34s %6s %6s %6s %11s %11s %8s %10s' % ('kernel', 'VGPRs', 'AGPRs', 'SGPRs', 'SGPR spill', 'VGPR spill',
                                                'scratch', 'waves/SIMD'))
for unit, flags in UNITS.items():
    out = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fno-gpu-rdc',
                          '--cuda-device-only', '-c', os.path.join(CSRC, unit + '.hip'), '-o', '/dev/null',
                          '-Rpass-analysis=kernel-resource-usage'] + flags, stderr=subprocess.PIPE).stderr.decode()
    rec = {}
    for line in out.splitlines():
        m = re.search(r'remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|'
                      r'Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]):\s+(\S+)', line)
        if not m:
            continue
        rec[m.group(1)] = m.group(2)
        if m.group(1).startswith('LDS Size'):
            name = subprocess.check_output(['c++filt', rec['Function Name']]).decode().split('(')[0].split('smart::')[-1]
            print('%-34s %6s %6s %6s %11s %11s %8s %10s' % (
                name.strip(), rec.get('VGPRs'), rec.get('AGPRs'), rec.get('TotalSGPRs'), rec.get('SGPRs Spill'),
                rec.get('VGPRs Spill'), rec.get('ScratchSize [bytes/lane]'), rec.get('Occupancy [waves/SIMD]')))
            rec = {}
