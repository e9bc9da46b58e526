"""PSparams -- the mesh object GELATO keeps in ``pdict["ps_params"]``: node count, LGR abscissae, differentiation matrix and
index ranges of every phase.  The method names and meanings are the boundary (lib/SectionParameters.py:30-114 is what
``Trajectory_Optimization.py:132`` builds and every ``lib/con_*.py`` reads); the object here is a view over three integer
tables and the engine's LGR generator (``gel_lgr_nodes`` / ``gel_lgr_diffmat`` through gelato_amd.PSfunctions):

    phase i:  n[i] collocation nodes,  control rows  u0[i] .. u0[i] + n[i] - 1,  state rows  x0[i] .. x0[i] + n[i]
              u0 = exclusive prefix sum of n,  x0 = u0 + i  (every phase owns one extra state row: its first, at tau = -1)
"""
import numpy as np

from . import PSfunctions


class PSparams:
    def __init__(self, num_nodes):
        n = np.asarray(list(num_nodes), dtype=np.int64)
        if n.ndim != 1 or n.size == 0 or np.any(n < 1):
            raise ValueError("num_nodes: one positive node count per phase")
        self._mesh = np.stack([n, np.cumsum(n) - n, np.cumsum(n) - n + np.arange(n.size)])   # rows: n, u0, x0
        self._lgr = {int(k): None for k in np.unique(n)}                                     # filled on first use, once per size

    # ---- LGR data (one generator call per distinct phase size) ----
    def _phase(self, i):
        i = int(i)
        if not 0 <= i < self._mesh.shape[1]:
            raise ValueError("phase %d of a mesh of %d" % (i, self._mesh.shape[1]))
        return i

    def _data(self, i):
        k = int(self._mesh[0, self._phase(i)])
        if self._lgr[k] is None:
            self._lgr[k] = (PSfunctions.nodes_LGR(k), PSfunctions.differentiation_matrix_LGR(k))
        return self._lgr[k]

    def tau(self, i):
        return self._data(i)[0]

    def D(self, i):
        return self._data(i)[1]

    def time_nodes(self, i, to, tf):
        """times of the phase's n + 1 state rows: the knot `to`, then the LGR abscissae mapped onto (to, tf]"""
        return np.concatenate([[to], self.tau(i) * (tf - to) / 2 + (tf + to) / 2])

    # ---- sizes ----
    def nodes(self, i):
        return int(self._mesh[0, self._phase(i)])

    def num_sections(self):
        return int(self._mesh.shape[1])

    def num_u(self):
        return int(self._mesh[0].sum())

    def num_x(self):
        return self.num_u() + self.num_sections()

    # ---- index ranges ----
    def index_start_u(self, i):
        return int(self._mesh[1, i])

    def index_end_u(self, i):
        return int(self._mesh[1, i] + self._mesh[0, i])

    def index_start_x(self, i):
        return int(self._mesh[2, i])

    def index_end_x(self, i):
        return int(self._mesh[2, i] + self._mesh[0, i] + 1)

    def get_index(self, section):
        """-> (ua, ub, xa, xb, n): the phase's control rows [ua, ub) and state rows [xa, xb)"""
        n, u0, x0 = (int(v) for v in self._mesh[:, section])
        return u0, u0 + n, x0, x0 + n + 1, n

    def __getitem__(self, i):
        tau, D = self._data(i)
        return {"index_start": self.index_start_u(i), "nodes": self.nodes(i), "D": D, "tau": tau}
