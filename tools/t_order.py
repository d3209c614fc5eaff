import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'oracle')
order = sys.argv[1]
import numpy as np
if order == 'torch_first':
    import torch
    print('torch sees', torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.ones(4, device='cuda'); print(x.sum().item())
import rkmh_amd
c = rkmh_amd.Context(0)
print('ctx ok', c.calc_hash(b"ACGTACGTACGTACGT"))
if order == 'lib_first':
    import torch
    try:
        print('torch sees', torch.cuda.is_available(), torch.cuda.device_count())
        x = torch.ones(4, device='cuda'); print(x.sum().item())
    except Exception as e:
        print('torch failed:', e)
maps = open('/proc/self/maps').read()
print(sorted(set(l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l or 'libhsa-runtime' in l)))
