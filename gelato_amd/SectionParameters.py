"""PSparams: per-phase LGR data and index arithmetic (same public interface as
lib/SectionParameters.py:30-114 of the reference)."""
import numpy as np

from .PSfunctions import differentiation_matrix_LGR, nodes_LGR


class PSparams:
    def __init__(self, num_nodes):
        self._num_nodes = [int(n) for n in num_nodes]
        self._num_sections = len(self._num_nodes)
        cache = {}
        for n in self._num_nodes:
            if n not in cache:
                cache[n] = (nodes_LGR(n), differentiation_matrix_LGR(n))
        self._tau = [cache[n][0] for n in self._num_nodes]
        self._D = [cache[n][1] for n in self._num_nodes]
        self._index_start_u = list(np.concatenate([[0], np.cumsum(self._num_nodes)[:-1]]).astype(int))
        self._N = int(sum(self._num_nodes))

    def _chk(self, i):
        if i < 0 or i >= self._num_sections:
            raise ValueError("Index out of range")

    def tau(self, i):
        self._chk(i)
        return self._tau[i]

    def D(self, i):
        self._chk(i)
        return self._D[i]

    def index_start_u(self, i):
        return self._index_start_u[i]

    def index_end_u(self, i):
        return self._index_start_u[i] + self._num_nodes[i]

    def index_start_x(self, i):
        return self._index_start_u[i] + i

    def index_end_x(self, i):
        return self.index_start_x(i) + self._num_nodes[i] + 1

    def num_u(self):
        return self._N

    def num_x(self):
        return self._N + self._num_sections

    def num_sections(self):
        return self._num_sections

    def nodes(self, i):
        self._chk(i)
        return self._num_nodes[i]

    def time_nodes(self, i, to, tf):
        t = np.zeros(self._num_nodes[i] + 1)
        t[0] = to
        t[1:] = self.tau(i) * (tf - to) / 2 + (tf + to) / 2
        return t

    def get_index(self, section):
        """-> ua, ub, xa, xb, n (start/end of the phase in u-indexing and x-indexing)."""
        ua = self._index_start_u[section]
        n = self._num_nodes[section]
        return ua, ua + n, ua + section, ua + section + n + 1, n

    def __getitem__(self, i):
        self._chk(i)
        return {"index_start": self._index_start_u[i], "nodes": self._num_nodes[i], "D": self._D[i],
                "tau": self._tau[i]}
