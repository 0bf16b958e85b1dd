"""ctypes front-end of the dense-solver oracle (oracle/sor_oracle.c, in libpic_oracle.so).

TEST INFRASTRUCTURE ONLY.  `OracleSOR` mirrors the object the reference's
matrix_webgl.makeSORIterative(spec) returns (matrix_webgl.js:35-711).
"""
import ctypes

import numpy as np

import pic_oracle


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class OracleSOR:
    def __init__(self, n_power, relaxation=None):
        self.n_power = int(n_power)
        self.omega = float(relaxation) if relaxation else 1.0          # spec.relaxation || 1.0 (:54)
        self.vec_height = 1 << self.n_power
        self.vec_length = 4 * self.vec_height ** 2
        self.mat_height = 2 * self.vec_height ** 2
        L = self.vec_length
        self.A = np.zeros(L * L, dtype=np.float32)
        self.b = np.zeros(L, dtype=np.float32)
        self.x_guess, self.x_result, self.x_stats = (np.zeros(L, dtype=np.float32) for _ in range(3))
        self._lib = pic_oracle.lib()

    def set_matrix(self, matrix):
        self.A[:] = np.asarray(matrix, dtype=np.float64).astype(np.float32).reshape(-1)
        return self

    def set_b(self, b):
        self.b[:] = np.asarray(b, dtype=np.float64).astype(np.float32)
        return self

    def init_vector(self, x):
        self.x_result[:] = np.asarray(x, dtype=np.float64).astype(np.float32)
        return self

    def build_R(self):
        R = np.zeros(4 * self.mat_height ** 2, dtype=np.float32)
        self._lib.orc_sor_build_R(_p(self.A), self.n_power, ctypes.c_double(self.omega), _p(R))
        return R

    def build_C(self):
        C = np.zeros(self.vec_length, dtype=np.float32)
        self._lib.orc_sor_build_C(_p(self.A), _p(self.b), self.n_power, ctypes.c_double(self.omega), _p(C))
        return C

    def solve(self, tolerance, substep=None, max_iterations=None):
        res = (ctypes.c_double * 3)()
        self._lib.orc_sor_solve(_p(self.A), _p(self.b), self.n_power, ctypes.c_double(self.omega), ctypes.c_double(tolerance),
                                int(substep or 0), int(max_iterations is not None), int(max_iterations or 0),
                                _p(self.x_guess), _p(self.x_result), _p(self.x_stats), res)
        # `result` is the closure array x2_arr: the last read-back of x_result, or, when no
        # iteration ran, what the debug read-back of C left in it (matrix_webgl.js:592-593)
        result = self.x_result.copy() if res[2] > 0 else self.build_C()
        return {"correlation": res[0], "diff": res[1], "iterations": int(res[2]), "result": result}
