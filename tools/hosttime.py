import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import dana_amd
from dana_amd import synthetic as S, ops
dev = torch.device('cuda:0')
import os
for mode in ('train-split', 'train-seq2', 'train-seq4'):
    training = mode.startswith('train')
    way = 2 if training else 1
    m = dana_amd.get_model('DAnA', pretrained=False, use_BA_block=False, way=2, shot=3, classes=['fg','bg'])
    m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile='test')); m.to(dev)
    m.train() if training else m.eval()
    m.merge_trunk = (mode == 'train')
    m.query_streams = {'train-seq2': 2, 'train-seq4': 4}.get(mode, 1)
    m.query_sequential = mode.startswith('train-seq')
    inputs = [t.to(dev) for t in S.episode_inputs(4, way, 3, 600, 1000, seed=1996)]
    np.random.seed(0)
    with torch.no_grad():
        for _ in range(3): m(*inputs)
        torch.cuda.synchronize()
        hs, ts = [], []
        for _ in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m(*inputs); t1 = time.perf_counter()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            hs.append(t1 - t0); ts.append(t2 - t0)
    print(mode, 'host enqueue ms %.2f  total ms %.2f' % (1e3*np.median(hs), 1e3*np.median(ts)))
    m._timeline = []
    m._gpu_events = []
    with torch.no_grad():
        m(*inputs)
    torch.cuda.synchronize()
    e0 = m._gpu_events[0][1]
    for name, e in m._gpu_events:
        print('   GPU %-55s @%.2f ms' % (name, e0.elapsed_time(e)))
    m._gpu_events = None
    t0 = m._timeline[0][1]
    for name, t in m._timeline:
        print('   %-55s +%.2f ms' % (name, 1e3 * (t - t0)))
