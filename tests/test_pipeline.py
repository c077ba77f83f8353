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


def test_dead_consumer_does_not_block_the_producer():
    """--parallel_run on one GPU: the mapper spins in a host thread and consumes a BOUNDED StreamQueue.  If that thread dies
    (a graph-capture fault, an out-of-memory), the module must read as shut down and a blocking put must raise instead of
    waiting forever on a queue nobody drains."""
    import time

    import pytest

    from nerfslam.pipeline import StreamQueue, spin_in_thread

    class _Boom:
        def fuse(self, packets):
            raise MemoryError("simulated mapper fault")

        def stop_condition(self):
            return False

    args = argparse.Namespace(parallel_run=True)
    fusion = FusionModule("nerf", args)
    fusion.fusion, fusion.is_initialized = _Boom(), True
    q = StreamQueue(maxsize=2)
    fusion.register_input_queue("slam", q)
    called = []
    fusion.register_on_failure_callback(lambda: called.append(1))
    worker = spin_in_thread(fusion, "cpu")
    q.consumer_alive = lambda: worker.is_alive() and getattr(fusion, "error", None) is None
    worker.join(timeout=10)
    assert not worker.is_alive() and fusion.shutdown and isinstance(fusion.error, MemoryError) and called == [1]
    q.put({"k": 0}); q.put({"k": 1})          # fills the queue
    t0 = time.time()
    with pytest.raises(RuntimeError):
        q.put({"k": 2})                       # would have blocked forever
    assert time.time() - t0 < 5


def test_graph_capture_keeps_the_collector_off_until_the_last_capture_ends(monkeypatch):
    """nerfslam._lib.graph_capture: the cyclic GC goes off with the first capture and comes back with the LAST one (captures of
    two threads may overlap), and every captured graph object stays referenced (never destroyed).  torch.cuda.graph itself is
    replaced by a no-op here: the bookkeeping is host logic."""
    import contextlib
    import gc
    import torch
    from nerfslam import _lib

    @contextlib.contextmanager
    def fake_graph(g, **kw):
        yield
    monkeypatch.setattr(torch.cuda, "graph", fake_graph)
    a, b = object(), object()
    n0 = len(_lib._immortal_graphs)
    assert gc.isenabled()
    with _lib.graph_capture(a):
        assert not gc.isenabled()
        with _lib.graph_capture(b, capture_error_mode="thread_local"):
            assert not gc.isenabled()
        assert not gc.isenabled()                 # the outer capture is still running
    assert gc.isenabled() and _lib._captures == 0
    assert _lib._immortal_graphs[n0:] == [b, a]
    gc.disable()                                  # a caller that runs without the collector keeps it off
    try:
        with _lib.graph_capture(object()):
            pass
        assert not gc.isenabled()
    finally:
        gc.enable()
