import sys, os
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
def maps():
    return sorted(set(l.split()[-1] for l in open('/proc/self/maps') if any(k in l for k in ('amdhip','hsa-runtime','comgr'))))
if order == 'torch_first':
    import torch
    print('torch avail', torch.cuda.is_available())
    import mrs_multirotor_simulator_amd as M
else:
    import mrs_multirotor_simulator_amd as M
    M.load_library()
    import torch
    print('torch avail', torch.cuda.is_available())
print(maps())
try:
    s = M.Swarm(128); print('swarm ok'); s.step(0.001); s.synchronize(); print('step ok')
except Exception as e: print('ERR', e)
x = torch.ones(4, device='cuda'); print('torch tensor ok', x.sum().item())
print(maps())
