import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, oracle, states, fd_noise
from gelato_amd import Engine
from test_exact_fd import STATES, block_entries
G = dict(np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/g15_exact_fd.npz")))
name = sys.argv[1]
prob, x = STATES[name]()
P = oracle.Problem(prob); prob = dict(prob); prob["tau"] = [P.tau(i) for i in range(P.S)]
terms = fd_noise.velocity_noise_terms(oracle, prob, x)
E = Engine(prob, D=[P.D(i) for i in range(P.S)], tau=prob["tau"], barC20=oracle.BARC20_CPP)
vals, rc = E.eval_jacobian(x)
J = E.jac_dicts(vals)["vel"]
E2 = Engine(prob, D=[P.D(i) for i in range(P.S)], tau=prob["tau"], barC20=oracle.BARC20_CPP, flags=8)
vals2, rc = E2.eval_jacobian(x)
J2 = E2.jac_dicts(vals2)["vel"]
Jo = P.jacobian("vel", x)
for ph in G[name + "_phases"]:
    ph = int(ph); t = terms[ph]
    ex = G["%s_p%d_position" % (name, ph)]
    g1 = block_entries(J, prob, ph, "position"); g2 = block_entries(J2, prob, ph, "position"); go = block_entries(Jo, prob, ph, "position")
    unit = fd_noise.EPS * t["chain"] * abs(t["scale"])
    e1 = np.abs(g1 - ex).max(axis=(1, 2)); e2 = np.abs(g2 - ex).max(axis=(1, 2)); eo = np.abs(go - ex).max(axis=(1, 2))
    order = np.argsort(-e1 / unit)[:8]
    print("phase", ph)
    for j in order:
        k = np.unravel_index(np.argmax(np.abs(g1[j] - ex[j])), (3, 3))
        print("  node %3d lat %6.1f alt %8.0f  chain %.3g  delta-form err %.2e (%.0f eps-chain) comp %s | recompute-form err %.2e | oracle err %.2e | exact %.4g  altterm %.2e" % (
            j, np.rad2deg(t["lat"][j]), t["alt"][j], t["chain"][j], e1[j], e1[j] / unit[j], k, e2[j], eo[j], ex[j][k], 2 * t["dfdalt"][j] * t["dalt"][j] * abs(t["scale"])))
