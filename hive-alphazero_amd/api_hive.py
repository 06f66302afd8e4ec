"""HiveModelAPI: mirror of the reference's woker/api_hive.py (:9-74) -- the pipe-served evaluator.

Same protocol: clients `send(planes[12,12,56])` on their end of a multiprocessing Pipe and `recv()`
`(p[1584] ndarray, float v)`.  The worker thread drains every ready pipe, stacks the planes and runs
ONE batched forward on the GPU.  Differences from the reference: it blocks on the pipes instead of
polling every millisecond, and the batch goes through alpha_net.InferenceNet (bf16 NHWC + HIP graph
for repeated batch sizes) when given one; a plain ChessNet is evaluated in fp32 NCHW like upstream.
"""
from multiprocessing import Pipe, connection
from threading import Thread

import numpy as np
import torch

from .alpha_net import InferenceNet


class HiveModelAPI:
    def __init__(self, agent_model):
        self.agent_model = agent_model
        self.pipes = []
        self._running = True

    def start(self):
        worker = Thread(target=self._predict_batch_worker, name="prediction_worker")
        worker.daemon = True
        worker.start()

    def create_pipe(self):
        me, you = Pipe()
        self.pipes.append(me)
        return you

    def stop(self):
        self._running = False

    def _forward(self, batch_hwc):
        m = self.agent_model
        if isinstance(m, InferenceNet):
            p, v = m(torch.from_numpy(batch_hwc).to(m.device))
        else:
            dev = next(m.parameters()).device
            with torch.no_grad():
                p, v = m(torch.from_numpy(np.ascontiguousarray(batch_hwc.transpose(0, 3, 1, 2))).to(dev))
        return p.detach().float().cpu().numpy(), v.detach().float().cpu().numpy().reshape(-1)

    def _predict_batch_worker(self):
        while self._running:
            try:
                ready = connection.wait(self.pipes, timeout=0.05)
            except (OSError, EOFError, ValueError):      # the clients closed their ends: stop serving
                return
            if not ready:
                continue
            data, result_pipes = [], []
            try:
                for pipe in ready:
                    while pipe.poll():
                        data.append(np.asarray(pipe.recv(), dtype=np.float32))
                        result_pipes.append(pipe)
            except (OSError, EOFError):
                return
            if not data:
                continue
            try:
                policy_ary, value_ary = self._forward(np.stack(data))
            except Exception as exc:          # keep serving: a failed batch answers with the exception
                for pipe in result_pipes:
                    pipe.send(exc)
                continue
            for pipe, p, v in zip(result_pipes, policy_ary, value_ary):
                pipe.send((p, float(v)))
