#!/usr/bin/env python3
"""Per-kernel resource table from the *.res files the Makefile keeps (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
for path in sys.argv[1:]:
    txt = open(path).read()
    for b in re.split(r'remark: Function Name: ', txt)[1:]:
        name = b.split()[0]
        try:
            dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        except FileNotFoundError:
            dem = name
        dem = re.sub(r'\(vivim_ssm_\w+ const[^)]*\)', '', dem).replace('void vivim::', '')
        g = lambda k: re.search(k + r': (\d+)', b).group(1)
        print(dem[:84].ljust(84), 'SGPR', g('TotalSGPRs').rjust(3), 'VGPR', g('VGPRs').rjust(3), 'scratch', g(r'ScratchSize \[bytes/lane\]').rjust(4),
              'occ', g(r'Occupancy \[waves/SIMD\]'), 'LDS', g(r'LDS Size \[bytes/block\]'))
