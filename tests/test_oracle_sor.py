"""The dense-solver restatement (oracle/sor_oracle.c) against the reference's own
makeSORIterative: run by a real WebGL (tests/golden/webgl_sor.*, oracle/make_golden_webgl.py:
Chromium's ANGLE/SwiftShader executing matrix_webgl.js) and evaluated in software
(tests/golden/swgl_sor.*, oracle/make_golden.js section 9).  Every texture and every number
solve() returns is compared bit for bit, against both."""
import numpy as np
import pytest

from helpers import load_f32gz, load_json, same_bits
from sor_oracle import OracleSOR

FIXTURES = {}
for _name in ("swgl_sor", "webgl_sor"):
    _m = load_json(_name + ".json")
    FIXTURES[_name] = (_m, load_f32gz(_m["file"]))
META, BLOB = FIXTURES["swgl_sor"]
CASES = [(f, c) for f in sorted(FIXTURES) for c in sorted(FIXTURES[f][0]["cases"])]


def arr(at, blob=None):
    blob = BLOB if blob is None else blob
    return blob[at[0]: at[0] + at[1]]


def num(v):
    return float("nan") if v == "NaN" else float(v)


def same_number(a, b):
    return (np.isnan(a) and np.isnan(b)) or a == b


@pytest.mark.parametrize("fixture,name", CASES)
def test_solver_replay_is_bit_identical(fixture, name):
    meta, blob = FIXTURES[fixture]
    case = meta["cases"][name]
    arr = lambda at: blob[at[0]: at[0] + at[1]]
    L = case["vec_length"]
    eq = OracleSOR(case["n_power"], case["relaxation"])
    assert eq.vec_length == L and eq.vec_height == case["vec_height"]
    eq.set_matrix(arr(case["A"]).reshape(L, L)).set_b(arr(case["b"])).init_vector(arr(case["x0"]))
    assert same_bits(eq.x_result, arr(case["x_after_init"]))
    for call in case["calls"]:
        res = eq.solve(**call["params"])
        assert same_bits(eq.build_R(), arr(call["R"])), "iteration matrix"
        assert same_bits(eq.build_C(), arr(call["C"])), "constant vector"
        assert res["iterations"] == call["iterations"]
        assert same_bits(eq.x_result, arr(call["x_result"]))
        assert same_bits(eq.x_guess, arr(call["x_guess"]))
        if call["iterations"]:
            assert same_bits(eq.x_stats, arr(call["x_stats"]))
        assert same_bits(res["result"], arr(call["result"]))
        assert same_number(res["diff"], num(call["diff"])), (res["diff"], call["diff"])
        assert same_number(res["correlation"], num(call["correlation"])), (res["correlation"], call["correlation"])


def test_the_real_webgl_and_the_software_evaluator_agree_bit_for_bit():
    """No transcendental and no division by a varying in these shaders: the two fixtures hold the same numbers."""
    (ms, bs), (mw, bw) = FIXTURES["swgl_sor"], FIXTURES["webgl_sor"]
    assert sorted(ms["cases"]) == sorted(mw["cases"])
    assert mw["gl"]["version"].startswith("WebGL 1.0")
    for name, cs in ms["cases"].items():
        cw = mw["cases"][name]
        for key in ("A", "b", "x0", "x_after_init"):
            assert same_bits(arr(cs[key], bs), arr(cw[key], bw)), (name, key)
        for a, b in zip(cs["calls"], cw["calls"]):
            assert a["iterations"] == b["iterations"] and a["diff"] == b["diff"] and a["correlation"] == b["correlation"]
            for key in ("result", "x_result", "x_guess", "x_stats", "R", "C"):
                assert same_bits(arr(a[key], bs), arr(b[key], bw)), (name, key)


def test_fixture_covers_the_quirks():
    """No max_iterations -> nothing runs and `result` is the constant vector; the first
    statistics of a zero start are NaN; relaxation != 1 takes the u_X path."""
    c = META["cases"]
    last = c["p3_jacobi"]["calls"][-1]
    assert last["iterations"] == 0 and "max_iterations" not in last["params"]
    assert same_bits(arr(last["result"]), arr(last["C"]))
    assert c["p1_jacobi"]["calls"][0]["correlation"] == "NaN"
    assert c["p2_relaxed"]["relaxation"] == 0.8


def test_row_permutation_quirk_q14():
    """Element e of the update receives matrix row (2X + c%2) + 2vh(2Y + c//2), not row e
    (matrix_webgl.js:389-424 against :222-262): the fixed point solves the row-permuted
    system, not A x = b, once vh > 1."""
    case = META["cases"]["p1_jacobi"]
    L, vh = case["vec_length"], case["vec_height"]
    A = arr(case["A"]).reshape(L, L).astype(np.float64)
    b = arr(case["b"]).astype(np.float64)
    x = arr(case["calls"][-1]["x_result"]).astype(np.float64)
    rho = np.empty(L, dtype=int)
    for Y in range(vh):
        for X in range(vh):
            for c in range(4):
                rho[4 * (X + vh * Y) + c] = (2 * X + c % 2) + 2 * vh * (2 * Y + c // 2)
    assert not np.array_equal(rho, np.arange(L))
    # fixed point of x[e] = sum_col R[rho(e)][col] x[col] + C[e]
    D = np.diag(A)
    Rm = -A / D[:, None]
    np.fill_diagonal(Rm, 0.0)
    fixed = np.linalg.solve(np.eye(L) - Rm[rho], b / D)
    assert np.abs(x - fixed).max() < 1e-5
    assert np.abs(x - np.linalg.solve(A, b)).max() > 1e-3
