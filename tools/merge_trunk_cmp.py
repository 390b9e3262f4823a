"""train-mode forward bs 4: two trunk streams (default) vs one merged query+support launch per layer"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import dana_amd
from dana_amd import synthetic as S
dev = torch.device('cuda:0')
m = dana_amd.get_model('DAnA', pretrained=False, use_BA_block=False, way=2, shot=3, classes=['fg', 'bg'])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile='test')); m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
for merge in (False, True, False, True):
    m.merge_trunk = merge
    np.random.seed(0)
    with torch.no_grad():
        for _ in range(5): m(*inputs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m(*inputs)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print('merge_trunk=%s  %.3f ms/step' % (merge, 1e3 * (t1 - t0) / 20))
