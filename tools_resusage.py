import re, sys
t=open('/root/repo/gapflow_amd/lib/resource_usage.txt').read()
pat=sys.argv[1] if len(sys.argv)>1 else 'k_step|k_ghost|k_finish'
for b in re.split(r'remark: Function Name: ', t)[1:]:
    name=b.split()[0]
    if re.search(pat,name):
        g=lambda k: re.search(k+r': (\d+)',b).group(1)
        print(name[:70], 'VGPR',g('VGPRs'),'SGPR',g('TotalSGPRs'),'scratch',g(r'ScratchSize \[bytes/lane\]'),'occ',g(r'Occupancy \[waves/SIMD\]'))
