"""cvxopt_amd.solvers validates the solver options exactly like the reference drivers (coneprog.py:425-455, :502-509, :1770-1801,
:1862-1869) -- same ValueError, same text, in the same order -- BEFORE anything touches the GPU, so this runs on a CPU-only host
against the live reference (oracle/_ref)."""
import numpy as np
import pytest

from cvxopt_amd import synth

BAD = [
    {'kktreg': -1.0}, {'kktreg': 'x'}, {'maxiters': 0}, {'maxiters': 2.5}, {'maxiters': '3'}, {'abstol': 'a'}, {'reltol': None},
    {'abstol': 0.0, 'reltol': 0.0}, {'abstol': -1.0, 'reltol': -1e-3}, {'feastol': 0.0}, {'feastol': -1.0}, {'feastol': 'f'},
    {'refinement': -1}, {'refinement': 1.5}, {'refinement': 'two'},
    {'maxiters': 0, 'feastol': -1.0},                      # two bad options: the one the reference checks first wins
    {'feastol': -1.0, 'refinement': -2},
]


def _message(call):
    with pytest.raises(ValueError) as ei:
        call()
    return str(ei.value)


@pytest.mark.parametrize("bad", BAD, ids=[repr(b) for b in BAD])
def test_bad_options_raise_the_reference_error(ref_cvxopt, bad):
    from cvxopt import matrix, solvers
    import cvxopt_amd.solvers as gs
    pr = synth.dense_qp(6, 9, seed=1)
    P, q, G, h = (matrix(pr[k]) for k in ('P', 'q', 'G', 'h'))
    o = dict(bad)
    o['show_progress'] = False
    want = _message(lambda: solvers.coneqp(P, q, G, h, options=o))
    assert _message(lambda: gs.coneqp(P, q, G, h, options=o)) == want
    assert _message(lambda: gs.qp(P, q, G, h, options=o)) == want
    sp = synth.socp(5, 2, 3, seed=2, ml=2)
    c, Gq, hq = matrix(sp['c']), matrix(sp['G']), matrix(sp['h'])
    want = _message(lambda: solvers.conelp(c, Gq, hq, sp['dims'], options=o))
    assert _message(lambda: gs.conelp(c, Gq, hq, sp['dims'], options=o)) == want


def test_a_bad_option_wins_over_a_bad_kktsolver_name_and_vice_versa(ref_cvxopt):
    """order of the checks (coneprog.py:435-466 / :1783-1813 then :502-509 / :1862-1869): tolerances, kktsolver name, refinement"""
    from cvxopt import matrix, solvers
    import cvxopt_amd.solvers as gs
    pr = synth.dense_qp(6, 9, seed=1)
    P, q, G, h = (matrix(pr[k]) for k in ('P', 'q', 'G', 'h'))
    for o in ({'feastol': -1.0}, {'refinement': -1}):
        want = _message(lambda: solvers.coneqp(P, q, G, h, kktsolver='nope', options=o))
        assert _message(lambda: gs.coneqp(P, q, G, h, kktsolver='nope', options=o)) == want
    want = _message(lambda: solvers.conelp(q, G, h, kktsolver='nope', options={'refinement': -1}))
    assert _message(lambda: gs.conelp(q, G, h, kktsolver='nope', options={'refinement': -1})) == want


def test_module_level_options_are_read_like_the_reference(ref_cvxopt):
    """no options= argument: solvers.options of the cvxopt package the caller uses (coneprog.py:425, :1770)"""
    from cvxopt import matrix, solvers
    import cvxopt_amd.solvers as gs
    pr = synth.dense_qp(6, 9, seed=1)
    P, q, G, h = (matrix(pr[k]) for k in ('P', 'q', 'G', 'h'))
    old = dict(solvers.options)
    try:
        solvers.options['maxiters'] = -3
        want = _message(lambda: solvers.coneqp(P, q, G, h))
        assert _message(lambda: gs.coneqp(P, q, G, h)) == want
    finally:
        solvers.options.clear()
        solvers.options.update(old)


def test_defaults_and_use_correction_reach_the_device_loop(ref_cvxopt, monkeypatch):
    """what _options hands to the device loops: the reference's defaults (refinement 0 / 1 by cone type) and use_correction"""
    import cvxopt_amd.solvers as gs
    lpd, qd = {'l': 4, 'q': [], 's': []}, {'l': 0, 'q': [3], 's': []}
    o, kktreg, debug, ks = gs._options({'options': {}}, lpd, False, None)
    assert o['refinement'] == 0 and o['use_correction'] is True and ks == 'chol2' and kktreg is None and not debug
    o, _, _, ks = gs._options({'options': {'use_correction': False, 'refinement': 3}}, qd, False, None)
    assert o['refinement'] == 3 and o['use_correction'] is False and ks == 'chol'
    o, _, _, ks = gs._options({'options': {}}, qd, True, None)
    assert o['refinement'] == 1 and 'use_correction' not in o and ks == 'qr'
    seen = {}

    def fake_device(P, q, G, h, dims, A, b, **kw):
        seen.update(kw)
        return {'x': np.zeros(1), 'y': np.zeros(0), 's': np.zeros(1), 'z': np.zeros(1)}
    monkeypatch.setattr(gs._kkt, "coneqp_device", fake_device)
    from cvxopt import matrix
    gs.coneqp(matrix(1.0), matrix(1.0), matrix(-1.0), matrix(0.0), options={'use_correction': False, 'maxiters': 7},
              initvals={})
    assert seen['use_correction'] is False and seen['maxiters'] == 7 and seen['initvals'] == {} and seen['kktsolver'] == 'chol2'
