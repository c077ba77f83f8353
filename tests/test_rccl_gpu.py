"""The RCCL path, executed: `torch.distributed` backend "nccl" (= RCCL on ROCm) with ONE rank on the GPU box.

Every other multi-rank test of this repo runs `gloo` (world size 2 or 3, host-staged collectives).  This one initialises the
backend the N-GPU run uses and drives the product's collectives through it on device tensors: the int64 all-to-all of the
packed table gradient, the f16 all-gather of the table, the f32 all-reduces of the MLP / pose gradients and of the reduced
camera system, the uint8 packet broadcast, `PacketChannel` publish / poll, and the replicated trainer's step as two HIP graphs
around its collectives.  (examples/slam_demo.py:63-77 is the reference's split; visual_frontend.py:1355-1360 its CPU bounce.)
Runs in a subprocess: the process group must not leak into the other tests of the session."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_rccl_collectives_and_replicated_step_with_one_rank(dev):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_worker.py"), str(port)], env=env, capture_output=True, text=True,
                       timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    res = json.loads(lines[-1])
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "rccl_one_rank.json"), "w") as f:
            json.dump(res, f, indent=1)
    assert res["backend"] == "nccl"
    assert all(res["checks"].values()), res
    assert p.returncode == 0, p.stderr[-4000:]
    # two graphs + stream-ordered collectives: the replicated step costs what the one-trainer step costs plus its extra passes
    # (gradient buffer, exchange of the whole table with itself, streaming Adam), not a host round trip per collective
    ms = res["ms_per_step"]
    assert ms["replicated_two_graphs"] <= ms["replicated_eager"] * 1.05 + 0.05, ms
