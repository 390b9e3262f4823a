import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dana_amd import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
B, n, topn = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
s = torch.from_numpy(rng.uniform(size=(B, n)).astype(np.float32)).to(dev)
for _ in range(20):
    ops.topk_desc(s, topn)
torch.cuda.synchronize()
