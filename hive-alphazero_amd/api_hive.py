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
        self.pipes = []                 # server ends, one per client (the reference keeps them under this name too)
        self._serving = True

    def start(self):
        Thread(target=self._serve, name="prediction_worker", daemon=True).start()

    def create_pipe(self):
        server_end, client_end = Pipe()
        self.pipes.append(server_end)
        return client_end

    def stop(self):
        self._serving = False

    def _forward(self, batch_hwc):
        m = self.agent_model
        if isinstance(m, InferenceNet):
            p, v = m(torch.from_numpy(batch_hwc).to(m.device))
        else:
            dev = next(m.parameters()).device
            with torch.no_grad():
                p, v = m(torch.from_numpy(np.ascontiguousarray(batch_hwc.transpose(0, 3, 1, 2))).to(dev))
        return p.detach().float().cpu().numpy(), v.detach().float().cpu().numpy().reshape(-1)

    @staticmethod
    def _drain(ends):
        """Every request waiting on the readable ends: ([planes float32[12,12,56]], [the end to answer on])."""
        planes, owners = [], []
        for end in ends:
            while end.poll():
                planes.append(np.asarray(end.recv(), dtype=np.float32))
                owners.append(end)
        return planes, owners

    def _serve(self):
        """Block until some client has sent planes, evaluate everything that is waiting in ONE forward, answer each
        client with (p[1584], float v) -- api_hive.py:47-74, which polls every millisecond instead."""
        while self._serving:
            try:
                readable = connection.wait(self.pipes, timeout=0.05)
                planes, owners = self._drain(readable) if readable else ([], [])
            except (OSError, EOFError, ValueError):      # the clients closed their ends: stop serving
                return
            if not planes:
                continue
            try:
                policies, values = self._forward(np.stack(planes))
            except Exception as exc:          # keep serving: a failed batch answers with the exception
                for end in owners:
                    end.send(exc)
                continue
            for end, p, v in zip(owners, policies, values):
                end.send((p, float(v)))
