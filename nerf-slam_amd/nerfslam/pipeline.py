"""Pipeline plugin surface: the module contract of the reference's pipeline
(/root/reference/pipeline/pipeline_module.py:7-182, slam/slam_module.py:5-22, fusion/fusion_module.py:3-45),
re-implemented compactly: same class names, constructor arguments and method names
(`register_input_queue`, `register_output_queue`, `register_output_callback`, `spin`, `spin_once`,
`initialize_module`, `shutdown_module`, `get_input_packet`, `push_output_packet`), so a driver written
against the reference's modules (examples/slam_demo.py:62-190) runs against these.

Only the transport under --multi_gpu differs: the SLAM -> fusion packet can travel as device tensors
over RCCL (nerfslam.transport) instead of being moved to the CPU and pickled
(visual_frontend.py:1355-1360).
"""
import logging
import queue as _queue
import threading

log = logging.getLogger("nerfslam.pipeline")


def _walk_tensors(obj):
    import torch
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _walk_tensors(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _walk_tensors(v)


class StreamQueue(_queue.Queue):
    """Queue between two modules that run in different host threads on different HIP streams of ONE process -- the
    single-GPU form of the reference's --parallel_run (one process per module and a torch.multiprocessing queue,
    examples/slam_demo.py:100-160).  put() records an event on the producer's current stream; get() makes the consumer's
    current stream wait for it and tells the caching allocator that the packet's tensors are in use there."""

    consumer_alive = None      # optional callable: False once the consuming thread has died (set by the driver)

    def put(self, item, block=True, timeout=None):
        import torch
        ev = None
        if item is not None and torch.cuda.is_available():
            ev = torch.cuda.Event()
            ev.record()
        if not block or timeout is not None or self.consumer_alive is None:
            return super().put((item, ev), block, timeout)
        # a blocking put into a bounded queue must not outlive its consumer: with the mapper thread dead (a graph-capture
        # fault, an out-of-memory) the tracker used to block here forever once the queue was full
        while True:
            try:
                return super().put((item, ev), True, 0.25)
            except _queue.Full:
                if not self.consumer_alive():
                    raise RuntimeError("StreamQueue: the consumer of this queue has died") from None

    def get(self, block=True, timeout=None):
        import torch
        item, ev = super().get(block, timeout)
        if ev is not None:
            s = torch.cuda.current_stream()
            s.wait_event(ev)
            for t in _walk_tensors(item):
                if t.is_cuda:
                    t.record_stream(s)
        return item


def spin_in_thread(module, device, stream=None):
    """run `module.spin()` (its parallel_run loop) in a host thread under its own HIP stream; returns the thread"""
    import contextlib
    import torch
    on_gpu = torch.device(device).type == "cuda"
    if on_gpu:
        stream = stream or torch.cuda.Stream(device=device)

    def work():
        if on_gpu:
            torch.cuda.set_device(device)
        torch.set_grad_enabled(False)
        with (torch.cuda.stream(stream) if on_gpu else contextlib.nullcontext()):
            try:
                module.spin()
            except BaseException as e:       # noqa: BLE001 -- surfaced through the module's failure callbacks
                log.error("module %s died: %s", module.name, e)
                module.error = e
                module.notify_on_failure()
                module.shutdown_module()         # drivers poll `shutdown`: a dead module must not look like a busy one
    t = threading.Thread(target=work, name=f"nerfslam-{module.name}", daemon=True)
    t.start()
    return t


class PipelineModuleBase:
    def __init__(self, name, parallel_run, args=None, grad=False):
        self.name, self.parallel_run, self.args, self.grad = name, parallel_run, args, grad
        self.shutdown = False
        self.is_initialized = False
        self.is_thread_working = False
        self.on_failure_callbacks = []

    def initialize_module(self):
        self.is_initialized = True

    def shutdown_queues(self):
        pass

    def shutdown_module(self):
        if self.shutdown:
            log.warning("module %s: shutdown requested twice", self.name)
        self.shutdown_queues()
        self.shutdown = True

    def restart(self):
        self.shutdown = False

    def register_on_failure_callback(self, callback):
        self.on_failure_callbacks.append(callback)

    def notify_on_failure(self):
        for cb in self.on_failure_callbacks:
            cb()


class PipelineModule(PipelineModuleBase):
    def get_input_packet(self):
        raise NotImplementedError

    def push_output_packet(self, output_packet):
        raise NotImplementedError

    def spin_once(self, input):
        raise NotImplementedError

    def spin(self):
        """Run until shutdown (parallel) or for one packet (sequential): pipeline_module.py:83-122."""
        if not self.is_initialized:
            self.initialize_module()
        while not self.shutdown:
            self.is_thread_working = False
            packet = self.get_input_packet()
            self.is_thread_working = True
            if packet is not None:
                out = self.spin_once(packet)
                if out is not None:
                    if not self.push_output_packet(out):
                        log.warning("module %s: output push failed", self.name)
                else:
                    self.notify_on_failure()
            if not self.parallel_run:
                self.is_thread_working = False
                return True
        self.is_thread_working = False
        return False


class MIMOPipelineModule(PipelineModule):
    def __init__(self, name_id, parallel_run, args=None, grad=False):
        super().__init__(name_id, parallel_run, args, grad)
        self.input_queues, self.output_callbacks, self.output_queues = {}, [], []

    def register_input_queue(self, name, input_queue):
        self.input_queues[name] = input_queue

    def register_output_callback(self, output_callback):
        self.output_callbacks.append(output_callback)

    def register_output_queue(self, output_queue):
        self.output_queues.append(output_queue)

    def push_output_packet(self, output_packet):
        ok = True
        for sink in [q.put for q in self.output_queues] + self.output_callbacks:
            try:
                sink(output_packet)
            except Exception as e:  # pushes are logged and swallowed (pipeline_module.py:141-157)
                log.warning("module %s: %s", self.name, e)
                ok = False
        return ok

    def get_input_packet(self, timeout=0.1):
        inputs = {}
        for name, q in self.input_queues.items():
            try:
                inputs[name] = q.get(timeout=timeout) if self.parallel_run else q.get_nowait()
            except _queue.Empty:
                pass
            except Exception as e:
                log.debug("module %s: %s", self.name, e)
        return inputs or None


class SlamModule(MIMOPipelineModule):
    """slam/slam_module.py:5-22.  `name` selects the SLAM class; "VioSLAM" maps to nerfslam.slam.TrackingSLAM."""

    def __init__(self, name, args, device="cpu"):
        super().__init__(name, args.parallel_run, args)
        self.device = device

    def spin_once(self, input):
        output = self.slam(input)
        if not output or self.slam.stop_condition():
            super().shutdown_module()
        return output

    def initialize_module(self):
        if self.name != "VioSLAM":
            raise NotImplementedError(self.name)
        from .slam import TrackingSLAM
        self.slam = TrackingSLAM(self.name, self.args, self.device)
        return super().initialize_module()


class FusionModule(MIMOPipelineModule):
    """fusion/fusion_module.py:3-45 ("nerf" only: TSDF / Sigma fusion are Open3D paths outside the hot path)."""

    def __init__(self, name, args, device="cpu"):
        super().__init__(name, args.parallel_run, args)
        self.device = device

    def spin_once(self, data_packet):
        output = self.fusion.fuse(data_packet)
        if self.fusion.stop_condition():
            super().shutdown_module()
        return output

    def initialize_module(self):
        if self.name != "nerf":
            raise NotImplementedError(f"fusion '{self.name}': only the NeRF mapper is part of this project")
        from .nerf_fusion import NerfFusion
        self.fusion = NerfFusion(self.name, self.args, self.device)
        return super().initialize_module()

    def get_input_packet(self):
        packet = super().get_input_packet(timeout=1e-10)  # never block the trainer on input (fusion_module.py:30-32)
        return packet if packet is not None else False


class DataModule(MIMOPipelineModule):
    """datasets/data_module.py (reference): pushes one dataset packet per spin.  The dataset readers
    (EuRoC / TUM / Replica / ...) are outside this project's scope; `dataset` is any sequence of packets in the
    reference's layout {"k": [k], "images": [HxWx3|4 uint8], "poses", "depths", "calibs", "t_cams", "is_last_frame"}."""

    def __init__(self, name, args, device="cpu", dataset=None):
        super().__init__(name, args.parallel_run, args)
        self.device, self.dataset, self._it = device, dataset, None

    def initialize_module(self):
        if self.dataset is None:
            raise NotImplementedError(f"dataset '{self.name}': pass a packet sequence (dataset=...)")
        self._it = iter(self.dataset)
        return super().initialize_module()

    def get_input_packet(self):
        return next(self._it, None)

    def spin_once(self, input):
        return input

    def spin(self):
        if not self.is_initialized:
            self.initialize_module()
        while not self.shutdown:
            packet = self.get_input_packet()
            if packet is None:
                super().shutdown_module()
                return False
            self.push_output_packet(packet)
            if not self.parallel_run:
                return True
        return False
