"""Plumbing of the plugin surface (pipeline_module.py:7-182, slam_module.py, fusion_module.py) on CPU with
stand-in SLAM / fusion objects: queue registration, one-packet sequential spin, shutdown propagation."""
import argparse
from queue import Queue

from nerfslam.pipeline import DataModule, FusionModule, MIMOPipelineModule, SlamModule


class _Slam:
    def __init__(self):
        self.n = 0

    def __call__(self, batch):
        self.n += 1
        return [None, {"k": batch["data"]["k"][0]}]

    def stop_condition(self):
        return self.n >= 3


class _Fusion:
    def __init__(self):
        self.seen, self.idle = [], 0

    def fuse(self, packets):
        if packets:
            self.seen.append(packets["slam"][1]["k"])
        else:
            self.idle += 1
        return True

    def stop_condition(self):
        return self.idle >= 2


def test_sequential_pipeline():
    args = argparse.Namespace(parallel_run=False)
    data = DataModule("seq", args, dataset=[{"k": [k]} for k in range(5)])
    slam, fusion = SlamModule("VioSLAM", args), FusionModule("nerf", args)
    slam.slam, slam.is_initialized = _Slam(), True
    fusion.fusion, fusion.is_initialized = _Fusion(), True
    q1, q2 = Queue(), Queue()
    data.register_output_queue(q1)
    slam.register_input_queue("data", q1)
    slam.register_output_queue(q2)
    fusion.register_input_queue("slam", q2)
    got = []
    slam.register_output_callback(lambda o: got.append(o[1]["k"]))
    while data.spin() and slam.spin() and fusion.spin():
        pass
    assert slam.shutdown and got == [0, 1, 2] and fusion.fusion.seen == [0, 1, 2]
    while fusion.spin():     # trainer keeps spinning on empty input until its own stop condition
        pass
    assert fusion.shutdown and fusion.fusion.idle == 2


def test_mimo_contract():
    m = MIMOPipelineModule("m", False)
    q = Queue()
    m.register_input_queue("a", q)
    assert m.get_input_packet() is None
    q.put(7)
    assert m.get_input_packet() == {"a": 7}
    fails = []
    m.register_on_failure_callback(lambda: fails.append(1))
    m.spin_once = lambda p: None
    q.put(1)
    assert m.spin() is True and fails == [1]
    m.shutdown_module()
    assert m.spin() is False
    m.restart()
    assert not m.shutdown
