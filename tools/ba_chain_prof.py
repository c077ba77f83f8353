"""One dense-BA iteration at C640 (M=96 edges, P=10 poses, 60x80 maps) back to back: linearise -> accumulate -> Schur ->
finalise -> solve + retract -> depth update.  Prints us per iteration (HIP events); run under `rocprofv3 --kernel-trace --stats`
for the per-kernel split."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "nerf-slam_amd"), os.path.join(root, "tools")]
import torch
from hot_path_chain import HotPath

dev = torch.device("cuda:0")
hp = HotPath(dev, seed=0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for cov in (False, True):
    for _ in range(20):
        hp.op_ba_iteration(cov)
    hp.cTw.copy_(hp.cTw0); hp.wTb.copy_(hp.wTb0); hp.disps.copy_(hp.disps0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        if i % 8 == 0:      # keep the synthetic problem stationary (the chain's own reset, hot_path_chain.HotPath.step)
            hp.cTw.copy_(hp.cTw0); hp.wTb.copy_(hp.wTb0); hp.disps.copy_(hp.disps0)
        hp.op_ba_iteration(cov)
    b.record()
    torch.cuda.synchronize()
    print("BA iteration, covariance factors %s: %.1f us (eager launches, host included)" % (cov, 1e3 * a.elapsed_time(b) / n))
