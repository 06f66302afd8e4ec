"""Deterministic evaluator stand-in shared by oracle/gen_golden.py (cmd_mcts) and the MCTS tests:
(p[1584] float32, v float) as a pure function of the planes; never touches the global numpy RNG."""
import zlib

import numpy as np


def stub_predict(planes):
    h = zlib.crc32(np.ascontiguousarray(planes, dtype=np.float32).tobytes())
    rs = np.random.RandomState(h)
    p = rs.dirichlet(np.full(1584, 0.5)).astype(np.float32)
    v = float(rs.uniform(-1.0, 1.0))
    return p, v


class StubPipe:
    def send(self, x):
        self._x = x

    def recv(self):
        return stub_predict(self._x)
