"""Host-side pieces of bench.py that need no GPU: the line-granular byte count of the lookup and the refusal to measure
with kernel-selection variables in the environment."""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def test_lookup_line_bytes_geometry():
    """an 8x8-tap window randomly aligned on 8x8-half tiles touches (1 + 7/8)^2 = 3.52 lines of 128 B on average; windows outside
    the volume touch none; an aligned window exactly one"""
    b = _bench()
    ht, wd, E = 64, 96, 6                                    # multiples of 64: every level's plane is whole tiles / whole lines
    rng = np.random.default_rng(0)
    # level-0 windows well inside the image, uniformly random alignment
    c = np.stack([rng.uniform(16, wd - 16, (E, ht, wd)), rng.uniform(16, ht - 16, (E, ht, wd))], -1).astype(np.float32)
    total, n = b.lookup_line_bytes(torch.from_numpy(c), ht, wd, tiled=True)
    assert n == E * ht * wd
    per = total / n
    # levels 0 and 1 (tiled): 3.52 lines each away from the border; levels 2, 3 (row-major planes of 16x24 / 8x12 halves):
    # 8 rows of 48 B / 24 B pitch -> ~3.5 and ~2 lines.  Between 10 and 13 lines per window in all.
    assert 10 * 128 < per < 13 * 128, per
    l0_only = np.full((1, 8, 8, 2), 3.0, np.float32)         # floor(3) - 3 = 0: the window IS tile (0, 0) of level 0
    t0, _ = b.lookup_line_bytes(torch.from_numpy(l0_only), 64, 64, tiled=True)
    assert t0 >= 64 * 128 and t0 % 128 == 0
    far = np.full((2, 4, 4, 2), 1e6, np.float32)
    far[0, 0, 0] = [np.nan, 1.0]
    assert b.lookup_line_bytes(torch.from_numpy(far), 64, 64, tiled=True)[0] == 0


def test_bench_refuses_kernel_selection_variables():
    env = dict(os.environ, NS_CONV_UT="2")
    env.pop("NS_VARIANTS", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "NS_CONV_UT" in (p.stderr + p.stdout) and "--allow-env-overrides" in (p.stderr + p.stdout)


def test_variant_switches_need_the_master_switch(monkeypatch):
    from nerfslam._lib import variant_env
    monkeypatch.delenv("NS_VARIANTS", raising=False)
    monkeypatch.setenv("NS_LOOKUP_UNFUSED", "1")
    assert variant_env("NS_LOOKUP_UNFUSED") is None and variant_env("NS_LOOKUP_UNFUSED", "x") == "x"
    monkeypatch.setenv("NS_VARIANTS", "1")
    assert variant_env("NS_LOOKUP_UNFUSED") == "1"


def test_device_collectives_by_backend_name(monkeypatch):
    """'nccl' anywhere in the backend string -> device collectives; gloo -> host staging; anything else raises"""
    import torch.distributed as dist
    from nerfslam import parallel
    for name, want in (("nccl", True), ("cpu:gloo,cuda:nccl", True), ("gloo", False)):
        monkeypatch.setattr(dist, "get_backend", lambda group=None, n=name: n)
        assert parallel._device_collectives(None) is want
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "undefined")
    try:
        parallel._device_collectives(None)
    except RuntimeError as e:
        assert "undefined" in str(e)
    else:
        raise AssertionError("an unknown backend must raise")
