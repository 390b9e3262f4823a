"""host enqueue time vs total time of the eval forward at bs 1 / 4 (is it launch-bound?)"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import dana_amd
from dana_amd import synthetic as S
dev = torch.device('cuda:0')
for bs in (1, 4):
    m = dana_amd.get_model('DAnA', pretrained=False, use_BA_block=False, way=1, shot=3, classes=['fg', 'bg'])
    m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile='test')); m.to(dev).eval()
    inputs = [t.to(dev) for t in S.episode_inputs(bs, 1, 3, 600, 1000, seed=1996)]
    with torch.no_grad():
        for _ in range(5): m(*inputs)
        torch.cuda.synchronize()
        hs, ts = [], []
        for _ in range(20):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m(*inputs); t1 = time.perf_counter()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            hs.append(t1 - t0); ts.append(t2 - t0)
    print('eval bs %d: host enqueue %.2f ms, total %.2f ms' % (bs, 1e3 * np.median(hs), 1e3 * np.median(ts)))
