"""Registers, scratch and occupancy of the kernels in the last build (gapflow_amd/lib/resource_usage.txt, written by
`python -m gapflow_amd.build` from hipcc's -Rpass-analysis=kernel-resource-usage).  Usage: python tools/resusage.py [regex]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
text = open(os.path.join(ROOT, 'gapflow_amd', 'lib', 'resource_usage.txt')).read()
pattern = sys.argv[1] if len(sys.argv) > 1 else 'k_step|k_ghost|k_begin'
for block in re.split(r'remark: Function Name: ', text)[1:]:
    name = block.split()[0]
    if re.search(pattern, name):
        get = lambda key: re.search(key + r': (\d+)', block).group(1)
        print(name[:70], 'VGPR', get('VGPRs'), 'SGPR', get('TotalSGPRs'), 'scratch', get(r'ScratchSize \[bytes/lane\]'),
              'occ', get(r'Occupancy \[waves/SIMD\]'))
